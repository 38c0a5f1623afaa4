#!/usr/bin/env python3
"""Robustness of the large-problem path (device hierarchy set-up + deflated COCG + SELL-64 with 16-bit deltas) away from
the structured annulus numbering: the same geometry with (a) interior nodes jittered by +-0.2 h, (b) the node numbering
permuted at random (wide band: every SELL slice falls back to 32-bit columns, aggregation sees an unordered graph),
(c) reverse Cuthill-McKee after the random permutation, (d) the Morton order of helmholtz_x.fem.locality_permutation.
The eigenvalues of (b), (c), (d) must equal those of (a) (the same discrete problem) to 1e-8.  usage: robustness_unstructured.py [annulus-260k]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "helmholtz-x_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ.setdefault("HX_QUIET", "1")

import numpy as np
import scipy.sparse as sp
from scipy.sparse.csgraph import reverse_cuthill_mckee

import bench
from helmholtz_x.fem import Mesh, MeshTags
from helmholtz_x.synthetic import micca_sound_speed


def renumber(case, perm):
    """perm[old] = new node id."""
    mesh = case["mesh"]
    m2 = mesh.renumbered(perm)
    ft, sd = case["facet_tags"], case["subdomains"]
    out = dict(case)
    out.update(mesh=m2, facet_tags=MeshTags(m2, ft.dim, ft.indices, ft.values),
               subdomains=MeshTags(m2, sd.dim, sd.indices, sd.values), c=micca_sound_speed(m2))
    out.pop("h", None)
    return out


def run(label, case):
    import contextlib
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(sys.stderr):
        E, _ = bench.hot_path(case, "amg")
    dt = time.perf_counter() - t0
    inner = next(v for v in E.inner_cache.values() if hasattr(v, "stats"))
    om = [E.getEigenvalue(i) for i in range(4)]
    st = inner.stats()
    print("%-22s %6.2f s  levels %s  %s  its/solve %.1f  fallbacks %d  ops %d" % (
        label, dt, st["levels"], st["fine_format"], st["iters_per_solve"], st["cocg_fallbacks"],
        E.state.get("krylov_ops")), flush=True)
    return om


def main(wl=None):
    wl = wl or (sys.argv[1] if len(sys.argv) > 1 else "annulus-260k")
    shape = bench.WORKLOADS[wl]
    case = bench.build_case(shape)
    mesh = case["mesh"]
    x = mesh.geometry.x.copy()
    rng = np.random.default_rng(0)
    # jitter interior nodes: h ~ smallest grid spacing
    nr, nth, nz = shape
    h = min(0.07 / nr, 0.29 / nz, 2 * np.pi * 0.14 / nth)
    r = np.hypot(x[:, 0], x[:, 1])
    interior = (r > 0.14 + 1e-9) & (r < 0.21 - 1e-9) & (x[:, 2] > -0.09 + 1e-9) & (x[:, 2] < 0.2 - 1e-9)
    x[interior] += rng.uniform(-0.2 * h, 0.2 * h, size=(int(interior.sum()), 3))
    base = dict(case)
    base["mesh"] = Mesh(x, mesh.topology.cells, facets=mesh.facets, facet_cells=mesh.facet_cells)
    base["facet_tags"] = MeshTags(base["mesh"], case["facet_tags"].dim, case["facet_tags"].indices, case["facet_tags"].values)
    base["subdomains"] = MeshTags(base["mesh"], case["subdomains"].dim, case["subdomains"].indices, case["subdomains"].values)
    base["c"] = micca_sound_speed(base["mesh"])
    run("warm-up (structured)", case)
    ref = run("jittered", base)
    n = x.shape[0]
    perm = rng.permutation(n).astype(np.int64)
    shuffled = renumber(base, perm)
    got = run("jittered + random ids", shuffled)
    cells = shuffled["mesh"].topology.cells.astype(np.int64)
    i = np.repeat(cells, 4, axis=1).ravel()
    j = np.tile(cells, (1, 4)).ravel()
    G = sp.csr_matrix((np.ones(i.size, np.int8), (i, j)), shape=(n, n))
    order = reverse_cuthill_mckee(G, symmetric_mode=True)          # order[new] = old
    p2 = np.empty(n, dtype=np.int64)
    p2[order] = np.arange(n)
    rcm = renumber(shuffled, p2)
    got2 = run("... + RCM", rcm)
    from helmholtz_x.fem import locality_permutation
    got3 = run("... + Morton order", renumber(shuffled, locality_permutation(shuffled["mesh"].geometry.x)))
    worst = 0.0
    for name, g in (("random", got), ("rcm", got2), ("morton", got3)):
        d = max(abs(a - b) / abs(b) for a, b in zip(sorted(g, key=lambda z: z.real), sorted(ref, key=lambda z: z.real)))
        print("max |d omega| / |omega| vs jittered structured numbering (%s): %.2e" % (name, d))
        assert d < 1e-8
        worst = max(worst, d)
    print(json.dumps({"omega": [[w.real, w.imag] for w in ref]}))
    return worst


if __name__ == "__main__":
    main()
