#!/usr/bin/env python3
"""Where the one-off AMG set-up time goes (host passes vs transfers), for one workload.  Usage: setup_breakdown.py [workload] [warm]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "helmholtz-x_amd"), os.path.join(ROOT, "tests")]
os.environ.setdefault("HX_QUIET", "1")
import torch  # noqa: E402

import bench  # noqa: E402
from helmholtz_x import amg  # noqa: E402

acc = {}


def timed(name, fn):
    def wrap(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn(*a, **k)
        torch.cuda.synchronize()
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return out
    return wrap


for f in ("_dev_filter", "_dev_aggregate", "_dev_transpose", "_dev_spgemm", "build_hierarchy_device", "_is_complex_symmetric",
          "_filter", "_aggregate", "_prolongator", "_galerkin", "_spectral_radius_dinv", "build_hierarchy"):
    setattr(amg, f, timed(f, getattr(amg, f)))
from helmholtz_x.petsc import CSR  # noqa: E402
from helmholtz_x import _hx  # noqa: E402


class _TimedLib:
    """libhx entry points of the set-up, timed (synchronised) by name"""

    def __init__(self, lib, names):
        self._lib, self._names = lib, set(names)

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        return timed("lib." + name, fn) if name in self._names else fn


_hx.lib = _TimedLib(_hx.lib, ["hx_sell_create", "hx_amg_enable_recycling", "hx_amg_set_level_c", "hx_amg_set_coarse_c",
                              "hx_amg_create", "hx_amg_destroy", "hx_sell_destroy", "hx_amg_set_level_sell"])
amg._hx = _hx
CSR.to_scipy = timed("K.to_scipy (D2H + scipy)", CSR.to_scipy)

case = bench.build_case(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "annulus-5M"])
case["mesh"].assembler()
import contextlib  # noqa: E402
with contextlib.redirect_stdout(sys.stderr):
    if len(sys.argv) > 2 and sys.argv[2] == "warm":      # first pass pays pinned-buffer / code-object one-offs
        E, _ = bench.hot_path(case)
        E.destroy()
        del E
        import gc
        gc.collect()
        acc.clear()
    E, _ = bench.hot_path(case)
inner = next(v for v in E.inner_cache.values() if hasattr(v, "stats"))
print("AMGSolver set-up total %.3f s" % inner.setup_s)
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-28s %.3f s" % (k, v))
