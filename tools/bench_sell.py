#!/usr/bin/env python3
"""CSR vs SELL-64 SpMV on the assembled annulus operator (complex128 and complex64)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "helmholtz-x_amd"), os.path.join(ROOT, "tests")]
os.environ.setdefault("HX_QUIET", "1")
import torch, bench
from helmholtz_x import _hx
from helmholtz_x.acoustic_matrices import AcousticMatrices
for name in sys.argv[1:]:
    case = bench.build_case(bench.WORKLOADS[name])
    P = case["P"]
    M = AcousticMatrices(case["mesh"], case["facet_tags"], {11: {'Robin': P.R_outlet}}, case["c"])
    K = M.A.csr_part()
    n, nnz = K.shape[0], K.nnz
    h = C.c_void_p()
    _hx.check(_hx.lib.hx_sell_create(C.byref(h), n, _hx.ptr(K.rowptr), _hx.ptr(K.col), _hx.ptr(K.val), 1, 1, _hx.stream()))
    ns, pad = C.c_int32(), C.c_int64()
    _hx.lib.hx_sell_info(h, C.byref(ns), C.byref(pad))
    x = torch.randn(n, dtype=torch.complex128, device="cuda"); y = torch.empty_like(x)
    x32 = x.to(torch.complex64); y32 = torch.empty_like(x32); v64 = K.val.to(torch.complex64)
    one, zero = _hx.z2(1.0), _hx.z2(0.0)
    def timeit(fn, reps=30):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    t_csr = timeit(lambda: K.spmv(x, y))
    t_sell = timeit(lambda: _hx.lib.hx_sell_spmv_z(h, _hx.ptr(x), _hx.ptr(y), one, zero, _hx.stream()))
    t_csr32 = timeit(lambda: _hx.lib.hx_csr_spmv_c64(n, nnz, _hx.ptr(K.rowptr), _hx.ptr(K.col), _hx.ptr(v64), _hx.ptr(x32), _hx.ptr(y32), 1.0, 0.0, _hx.stream()))
    t_sell32 = timeit(lambda: _hx.lib.hx_sell_spmv_c64(h, _hx.ptr(x32), _hx.ptr(y32), 1.0, 0.0, _hx.stream()))
    alg = nnz * 20 + n * 36
    alg32 = nnz * 12 + n * 20
    print("%s n=%d nnz=%d padding=%.3f | c128: CSR %.4f ms %.0f GB/s, SELL %.4f ms %.0f GB/s | c64: CSR %.4f ms %.0f GB/s, SELL %.4f ms %.0f GB/s"
          % (name, n, nnz, pad.value / nnz, t_csr, alg / t_csr / 1e6, t_sell, alg / t_sell / 1e6, t_csr32, alg32 / t_csr32 / 1e6, t_sell32, alg32 / t_sell32 / 1e6), flush=True)
    _hx.lib.hx_sell_destroy(h)
