#!/usr/bin/env python3
"""Minimal SpMV driver for counter collection: assemble K on a synthetic annulus and launch hx_csr_spmv_z and hx_sell_spmv_z N times.
Usage: python3 tools/spmv_only.py annulus-5M 20   (run under rocprofv3 --pmc ... --kernel-trace)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "helmholtz-x_amd"), os.path.join(ROOT, "tests")]
os.environ.setdefault("HX_QUIET", "1")
import torch  # noqa: E402

import bench  # noqa: E402
from helmholtz_x.acoustic_matrices import AcousticMatrices  # noqa: E402
from helmholtz_x.petsc import combine_csr  # noqa: E402

name, reps = sys.argv[1], int(sys.argv[2])
case = bench.build_case(bench.WORKLOADS[name])
P = case["P"]
M = AcousticMatrices(case["mesh"], case["facet_tags"], {11: {'Robin': P.R_outlet}}, case["c"])
K = combine_csr(M.A.shape, M.A.csr_terms + [(P.target * c, m) for c, m in M.B.csr_terms]
                + [(P.target ** 2 * c, m) for c, m in M.C.csr_terms])
x = torch.randn(K.shape[0], dtype=torch.complex128, device=K.val.device)
y = torch.empty_like(x)
import ctypes as C  # noqa: E402
from helmholtz_x import _hx  # noqa: E402
sell = C.c_void_p()
_hx.check(_hx.lib.hx_sell_create(C.byref(sell), K.shape[0], _hx.ptr(K.rowptr), _hx.ptr(K.col), _hx.ptr(K.val), 1, 0, _hx.stream()))
one, zero = _hx.z2(1.0), _hx.z2(0.0)
for _ in range(reps):
    K.spmv(x, y)                                                              # k_spmv<8, 0, c128>
    _hx.lib.hx_sell_spmv_z(sell, _hx.ptr(x), _hx.ptr(y), one, zero, _hx.stream())   # k_sell<c128, 0>
torch.cuda.synchronize()
print("n=%d nnz=%d algorithmic_bytes=%d" % (K.shape[0], K.nnz, K.nnz * 20 + K.shape[0] * 36))
