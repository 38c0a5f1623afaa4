import os, sys
ROOT="/root/repo"
sys.path[:0]=[ROOT, os.path.join(ROOT,"helmholtz-x_amd"), os.path.join(ROOT,"tests")]
os.environ.setdefault("HX_QUIET","1")
import bench
from helmholtz_x import _hx
case = bench.build_case(bench.WORKLOADS[sys.argv[1]])
E,_ = bench.hot_path(case)
inner = next(v for v in E.inner_cache.values() if hasattr(v,"stats"))
print("fine stencil", _hx.lib.hx_sell_stencil_slices(inner.sell), "levels", [lv["n"] for lv in inner.levels], "nnz", [lv["K"].nnz for lv in inner.levels])
for shl in inner.coarse_sells:
    import ctypes as C
    ns=C.c_int32(0); pe=C.c_int64(0)
    _hx.lib.hx_sell_info(shl, C.byref(ns), C.byref(pe))
    print("coarse sell: slices", ns.value, "padded entries", pe.value, "stencil slices", _hx.lib.hx_sell_stencil_slices(shl), "index bits", _hx.lib.hx_sell_index_bits(shl), "wide", _hx.lib.hx_sell_wide_slices(shl))
