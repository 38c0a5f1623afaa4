#!/usr/bin/env python3
"""Minimal driver for counter collection on the SELL-64 kernels: operator product (complex128), V-cycle residual and smoother
step (complex64 arithmetic on complex-half values) on the annulus operator.
Usage: sell_kernels_only.py annulus-5M 10 [numbering [jitter]]   (numbering: native | rcm | morton | random)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "helmholtz-x_amd"), os.path.join(ROOT, "tests")]
os.environ.setdefault("HX_QUIET", "1")
import torch  # noqa: E402

import bench  # noqa: E402
from helmholtz_x import _hx  # noqa: E402
from helmholtz_x.acoustic_matrices import AcousticMatrices  # noqa: E402
from helmholtz_x.petsc import combine_csr  # noqa: E402

name, reps = sys.argv[1], int(sys.argv[2])
numbering = sys.argv[3] if len(sys.argv) > 3 else "native"
jitter = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
shape = bench.WORKLOADS[name] if name in bench.WORKLOADS else tuple(int(v) for v in name.split("x"))   # or NRxNTxNZ
case = bench.build_case_real(shape[1]) if shape[0] == "real" else bench.build_case(shape, numbering, jitter)
P = case["P"]
M = AcousticMatrices(case["mesh"], case["facet_tags"], {11: {'Robin': P.R_outlet}}, case["c"])
K = combine_csr(M.A.shape, M.A.csr_terms + [(P.target * c, m) for c, m in M.B.csr_terms]
                + [(P.target ** 2 * c, m) for c, m in M.C.csr_terms])
n = K.shape[0]
h = C.c_void_p()
_hx.check(_hx.lib.hx_sell_create(C.byref(h), n, _hx.ptr(K.rowptr), _hx.ptr(K.col), _hx.ptr(K.val), 1, 2, _hx.stream()))
x = torch.randn(n, dtype=torch.complex128, device=K.val.device)
y = torch.empty_like(x)
x32, y32, b32, d32, p32 = (torch.randn(n, dtype=torch.complex64, device=K.val.device) for _ in range(5))
one, zero = _hx.z2(1.0), _hx.z2(0.0)
calls = {
    "k_sell<c128,0>   operator product": lambda: _hx.lib.hx_sell_spmv_z(h, _hx.ptr(x), _hx.ptr(y), one, zero, _hx.stream()),
    "k_sell<c64,0,h2> residual        ": lambda: _hx.lib.hx_sell_spmv_c64(h, _hx.ptr(x32), _hx.ptr(y32), -1.0, 1.0, _hx.stream()),
    "k_sell<c64,1,h2> smoother step   ": lambda: _hx.lib.hx_sell_smooth_c64(h, _hx.ptr(d32), _hx.ptr(b32), _hx.ptr(x32), _hx.ptr(y32), 0.6,
                                                                            _hx.ptr(p32), 0.2, _hx.stream()),
}

_ns, _pad, _tail, _split = C.c_int32(), C.c_int64(), C.c_int64(), C.c_int32()
_hx.lib.hx_sell_info(h, C.byref(_ns), C.byref(_pad))
_hx.lib.hx_sell_tail_info(h, C.byref(_tail), C.byref(_split))
print("n=%d nnz=%d stencil slices %d; stored %d + tail %d entries in %d of %d slices" % (
    n, K.nnz, _hx.lib.hx_sell_stencil_slices(h), _pad.value, _tail.value, _split.value, _ns.value))
for label, fn in calls.items():          # back-to-back launches between two events: average time per launch
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%s  %.1f us per launch" % (label, 1e3 * e0.elapsed_time(e1) / reps))
