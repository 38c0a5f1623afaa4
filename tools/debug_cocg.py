#!/usr/bin/env python3
"""Convergence histories of hx_cocg / hx_gmres on a synthetic annulus (HX_GMRES_TRACE=1 prints them to stderr)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "helmholtz-x_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ.setdefault("HX_QUIET", "0")
os.environ.setdefault("HX_GMRES_TRACE", "1")
import numpy as np, torch
import bench
from helmholtz_x import _hx
from helmholtz_x.acoustic_matrices import AcousticMatrices
from helmholtz_x.amg import AMGSolver
from helmholtz_x.petsc import combine_csr, to_device

wl = sys.argv[1] if len(sys.argv) > 1 else "annulus-16k"
case = bench.build_case(bench.WORKLOADS[wl])
P, mesh = case["P"], case["mesh"]
M = AcousticMatrices(mesh, case["facet_tags"], {11: {'Robin': P.R_outlet}}, case["c"])
s = P.target
K = combine_csr(M.A.shape, M.A.csr_terms + [(s * c, m) for c, m in M.B.csr_terms] + [(s ** 2 * c, m) for c, m in M.C.csr_terms])
solver = AMGSolver(K, recycle_min_rows=0, recycle=int(os.environ.get("K", "6")))
rng = np.random.default_rng(0)
for k, relax in enumerate([1.0, 1.0, 1.0, 30.0, 30.0, 1e4, 1e2, 1.0]):
    b = to_device(rng.standard_normal(K.shape[0]) + 1j * rng.standard_normal(K.shape[0]))
    if k % 2:
        bb = torch.empty_like(b)
        M.C.csr_part().spmv(b, bb)               # smooth right-hand sides, as on the path
        b = bb
    x = torch.empty_like(b)
    solver.solve(b, x, relax=relax)
    print("solve %d: its %d relres %.2e stats %s err=%s" % (k, solver.last[0], solver.last[1], solver.stats(), _hx.lib.hx_last_error()), file=sys.stderr)
