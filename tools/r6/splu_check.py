"""Device sparse LU (hx_splu_*) on K = A + sigma B + sigma^2 C of the reference's MICCA mesh (refined `level` times):
residuals, agreement with SciPy's SuperLU, factorisation / solve times.  usage: splu_check.py [max_level] [--only level] [--graph] [--general] [--no-scipy]  (--general: no symmetric mode)"""
import ctypes as C
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "helmholtz-x_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from cases import Micca
from helmholtz_x import _hx, synthetic
from helmholtz_x.io_utils import read_xdmf
from helmholtz_x.petsc import to_device
from oracle import fem


def build(levels):
    mesh, sub, ft = read_xdmf(os.path.join(ROOT, "tests", "golden", "meshes", "micca_full", "mesh"))
    out = []
    for lvl in range(levels + 1):
        if lvl:
            mesh, sub, ft = synthetic.refine_tets(mesh, cell_tags=sub, facet_tags=ft)
        x, cells = mesh.geometry.x, mesh.topology.cells
        n = x.shape[0]
        c = Micca.c_of_z(x[cells].mean(axis=1)[:, 2])
        A = fem.assemble_A(x, cells, c, c_is_cellwise=True)
        Cm = fem.assemble_C(x, cells)
        sel = ft.find(11)
        B = fem.assemble_B(x, n, [(mesh.facets[sel], fem.robin_Z(Micca.R_outlet), c, mesh.facet_cells[sel])])
        s = Micca.target
        K = (A + s * B + s * s * Cm).tocsr()
        K.sort_indices()
        out.append((K, x))
    return out


def main():
    levels = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 0
    use_scipy = "--no-scipy" not in sys.argv
    only = int(sys.argv[sys.argv.index("--only") + 1]) if "--only" in sys.argv else None
    graph_too = "--graph" in sys.argv
    for lvl, (K, x) in enumerate(build(levels)):
        if only is not None and lvl != only:
            continue
        n = K.shape[0]
        rp = np.ascontiguousarray(K.indptr, np.int32)
        ci = np.ascontiguousarray(K.indices, np.int32)
        xy = np.ascontiguousarray(x, np.float64)
        for coords in ((xy, None) if graph_too else (xy,)):
            h = C.c_void_p()
            t0 = time.perf_counter()
            _hx.check(_hx.lib.hx_splu_analyse(C.byref(h), n, _hx.nptr(rp), _hx.nptr(ci), None if coords is None else _hx.nptr(coords),
                                               0 if coords is None else 3, 0, 0 if "--general" in sys.argv else 1))
            ta = time.perf_counter() - t0
            val = to_device(np.ascontiguousarray(K.data, np.complex128))
            io, do = (C.c_int64 * 16)(), (C.c_double * 8)()
            for rep in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _hx.check(_hx.lib.hx_splu_factor(h, _hx.ptr(val), _hx.stream()))
                torch.cuda.synchronize()
                tf = time.perf_counter() - t0
                _hx.check(_hx.lib.hx_splu_info(h, io, do))
                print("level %d n %d %s: analyse %.3f s, factor wall %.4f s (events %.4f s) = %.2f TFLOP/s executed (%s mode), nnz(L) %.3e, fronts %d, levels %d, "
                      "launches f/fw/bw %d/%d/%d, tiny pivots %d, factor %.2f GB temp %.2f GB"
                      % (lvl, n, "coords" if coords is not None else "graph", ta, tf, do[2], do[4] / max(do[2], 1e-9) / 1e12, "symmetric" if do[5] else "general", io[3], io[1], io[2],
                         io[13], io[14], io[15], io[12], io[4] * 16 / 1e9, io[6] * 16 / 1e9), flush=True)
            rng = np.random.default_rng(0)
            b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
            bd = to_device(b)
            xd = torch.empty_like(bd)
            for tr in ((0,) if "--no-trans" in sys.argv else (0, 1)):
                _hx.check(_hx.lib.hx_splu_solve(h, _hx.ptr(bd), _hx.ptr(xd), tr, _hx.stream()))
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10):
                    _hx.check(_hx.lib.hx_splu_solve(h, _hx.ptr(bd), _hx.ptr(xd), tr, _hx.stream()))
                torch.cuda.synchronize()
                ts = (time.perf_counter() - t0) / 10
                xs = xd.cpu().numpy()
                M = K.T if tr else K
                print("   solve%s %.3f ms (%.2f TB/s of 2 nnz(L) 16 B), residual %.2e" %
                      (" transposed" if tr else "", ts * 1e3, 2 * io[3] * 16 / ts / 1e12, np.linalg.norm(M @ xs - b) / np.linalg.norm(b)), flush=True)
            if use_scipy and lvl == 0 and coords is not None:
                t0 = time.perf_counter()
                lu = spla.splu(K.tocsc())
                print("   scipy splu %.2f s, nnz(L) %.3e" % (time.perf_counter() - t0, lu.L.nnz))
                _hx.check(_hx.lib.hx_splu_solve(h, _hx.ptr(bd), _hx.ptr(xd), 0, _hx.stream()))
                ref = lu.solve(b)
                print("   max |x - x_splu| / max |x_splu| = %.2e" % (np.max(np.abs(xd.cpu().numpy() - ref)) / np.max(np.abs(ref))))
            _hx.check(_hx.lib.hx_splu_destroy(h))


if __name__ == "__main__":
    main()
