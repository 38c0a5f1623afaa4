#!/usr/bin/env python3
"""Times of the Gram-Schmidt panel kernels of the Krylov-Schur basis (hx_zpanel_dot, hx_zpanel_sub) against their bytes:
   python tools/bench_panel_ops.py [n] -- k columns of length n (complex128), k = 4 ... 40."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "helmholtz-x_amd")]
import torch  # noqa: E402

from helmholtz_x import _hx  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_005_056
dev = _hx.require_gpu()
V = torch.randn((41, n), dtype=torch.complex128, device=dev)
w = torch.randn(n, dtype=torch.complex128, device=dev)
h = torch.zeros(64, dtype=torch.complex128, device=dev)
ws = torch.empty(_hx.lib.hx_panel_dot_ws(41), dtype=torch.complex128, device=dev)
print("n = %d" % n)
for k in (4, 8, 16, 24, 32, 40):
    for name, fn in (("dot", lambda: _hx.check(_hx.lib.hx_zpanel_dot(n, k, _hx.ptr(V), V.stride(0), _hx.ptr(w), _hx.ptr(h), _hx.ptr(ws), _hx.stream()))),
                     ("sub", lambda: _hx.check(_hx.lib.hx_zpanel_sub(n, k, _hx.ptr(V), V.stride(0), _hx.ptr(h), _hx.ptr(w), _hx.stream())))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        gb = (k + (1 if name == "dot" else 2)) * n * 16 / 1e9
        print("  k = %2d  %s  %.3f ms  %.2f GB  %.2f TB/s" % (k, name, ms, gb, gb / ms))
