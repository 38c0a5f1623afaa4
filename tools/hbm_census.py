#!/usr/bin/env python3
"""Where the HBM of one fixed-point call sits: device memory in use, torch allocator (allocated / reserved),
libhx pool cache, after building the case, after every call and after the solver objects are destroyed.
Usage: python tools/hbm_census.py [workload] [steps]"""
import ctypes as C
import gc
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "helmholtz-x_amd"), os.path.join(ROOT, "tests")]
os.environ.setdefault("HX_QUIET", "1")
import torch  # noqa: E402

import bench  # noqa: E402
from helmholtz_x import _hx  # noqa: E402


def census(tag):
    torch.cuda.synchronize()
    free_b, total_b = torch.cuda.mem_get_info()
    cached = C.c_int64(0)
    _hx.check(_hx.lib.hx_pool_stats(None, None, None, None, C.byref(cached), 0))
    g = 2.0 ** 30
    rec = {"at": tag, "device_in_use_gib": round((total_b - free_b) / g, 2), "torch_allocated_gib": round(torch.cuda.memory_allocated() / g, 2),
           "torch_reserved_gib": round(torch.cuda.memory_reserved() / g, 2), "libhx_pool_cached_gib": round(cached.value / g, 2)}
    rec["libhx_live_and_runtime_gib"] = round(rec["device_in_use_gib"] - rec["torch_reserved_gib"] - rec["libhx_pool_cached_gib"], 2)
    print(json.dumps(rec), flush=True)
    return rec


def live_tensors(top=14):
    """Largest device tensors still referenced, with the kind of object that holds each (leak hunting)."""
    seen, rows = set(), []
    for o in gc.get_objects():
        try:
            if torch.is_tensor(o) and o.is_cuda:
                st = o.untyped_storage()
                if st.data_ptr() in seen:
                    continue
                seen.add(st.data_ptr())
                holders = []
                for r in gc.get_referrers(o)[:6]:
                    if isinstance(r, dict):
                        owners = [type(q).__name__ for q in gc.get_referrers(r)[:3] if not isinstance(q, (list, tuple))]
                        keys = [k for k, v in r.items() if v is o][:2]
                        holders.append("dict%s of %s" % (keys, owners))
                    elif not isinstance(r, type(sys._getframe())):
                        holders.append(type(r).__name__)
                rows.append((st.nbytes(), tuple(o.shape), str(o.dtype), holders[:3]))
        except Exception:
            pass
    rows.sort(key=lambda r: -r[0])
    for nb, shape, dt, holders in rows[:top]:
        print(json.dumps({"live_tensor_mib": round(nb / 2 ** 20, 1), "shape": shape, "dtype": dt, "held_by": holders}), flush=True)
    print(json.dumps({"live_device_tensors": len(rows), "total_gib": round(sum(r[0] for r in rows) / 2 ** 30, 2)}), flush=True)



def who_holds(cls_name, depth=8):
    """Referrer chains (type names; dict keys where they help) of every live instance of a class: what keeps it alive."""
    me = sys._getframe()
    found = [q for q in gc.get_objects() if type(q).__name__ == cls_name]
    skip = {id(found), id(me)}

    def rank(r):
        if isinstance(r, dict):
            return 0
        if hasattr(r, "__dict__") and not isinstance(r, type):
            return 1
        if isinstance(r, tuple):
            return 2
        return 3 if isinstance(r, list) and len(r) < 64 else 9

    for o in found:
        chain, cur = [], o
        for _ in range(depth):
            refs = [r for r in gc.get_referrers(cur) if id(r) not in skip and not isinstance(r, type(me)) and type(r).__name__ != "cell"
                    and r is not globals() and rank(r) < 9]
            if not refs:
                break
            r = min(refs, key=rank)
            if isinstance(r, dict):
                keys = [str(k) for k, v in r.items() if v is cur][:1]
                chain.append("dict[%s]" % (keys[0] if keys else "?"))
            else:
                chain.append(type(r).__name__)
            skip.add(id(r))
            cur = r
        print(json.dumps({"instance": cls_name, "shape": getattr(o, "shape", None), "held_through": chain}), flush=True)



workload = sys.argv[1] if len(sys.argv) > 1 else "annulus-5M"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
census("process start")
case = bench.build_case(bench.WORKLOADS[workload])
case["mesh"].assembler()
census("case resident")
import contextlib  # noqa: E402
E = None
for k in range(steps):
    if E is not None:
        E.destroy()
    E = matrices = None
    gc.collect()
    with contextlib.redirect_stdout(sys.stderr):
        E, matrices = bench.hot_path(case)
    census("after call %d (solver objects alive)" % (k + 1))
print(json.dumps({"listing": "largest device tensors while the solver objects of the last call are alive"}))
live_tensors(40)
E.destroy()
census("E.destroy() alone (no garbage collection yet)")
for cls in ("DeviceBasis", "AMGSolver", "WoodburySolver"):
    who_holds(cls)
E = matrices = None
census("references dropped (no garbage collection yet)")
gc.collect()
census("solver objects destroyed")


live_tensors()


who_holds("CSR")
who_holds("AcousticMatrices")
torch.cuda.empty_cache()
census("torch cache emptied")
_hx.check(_hx.lib.hx_pool_trim())
census("libhx pool trimmed")
