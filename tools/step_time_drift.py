#!/usr/bin/env python3
"""Do consecutive fixed-point calls slow down, and is it the machine or the process?  N calls, a pause, N calls (same process).
Usage: python tools/step_time_drift.py [workload] [N] [pause_s]"""
import contextlib
import gc
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "helmholtz-x_amd"), os.path.join(ROOT, "tests")]
os.environ.setdefault("HX_QUIET", "1")
import torch  # noqa: E402

import bench  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "annulus-1M"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
pause = float(sys.argv[3]) if len(sys.argv) > 3 else 15.0
MODE = os.environ.get("DRIFT_MODE", "")          # "profile": in-situ HIP events as bench.py; "destroy": destroy + gc inside the timing


def clocks():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(out)
        c = d[sorted(d)[0]]
        return {k: v for k, v in c.items() if any(t in k.lower() for t in ("sclk", "mclk", "power", "junction", "edge"))}
    except Exception as e:      # diagnostics only
        return {"rocm-smi": repr(e)}


case = bench.build_case(bench.WORKLOADS[workload])
case["mesh"].assembler()
E = None
if "profile" in MODE:
    from helmholtz_x import _hx
    _hx.check(_hx.lib.hx_profile_sell_spmv(1, case["mesh"].num_vertices))
for leg in range(2):
    times = []
    parts = []
    for k in range(N):
        if "destroy" in MODE:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        ta = time.perf_counter()
        if E is not None:
            E.destroy()
        tb = time.perf_counter()
        E = None
        tc = time.perf_counter()
        gc.collect()
        td = time.perf_counter()
        torch.cuda.synchronize()
        te = time.perf_counter()
        parts.append([round(1e3 * (b - a), 1) for a, b in ((ta, tb), (tb, tc), (tc, td), (td, te))])
        if "destroy" not in MODE:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        with contextlib.redirect_stdout(sys.stderr):
            E, _m = bench.hot_path(case)
        torch.cuda.synchronize()
        times.append(round(1e3 * (time.perf_counter() - t0), 1))
        del _m
        if "freeze" in MODE and leg == 0 and k == 0:
            gc.collect()
            gc.freeze()              # everything alive now (modules, the case) leaves the collector's generations
    print(json.dumps({"mode": MODE, "leg": leg, "ms_each_call": times, "ms_destroy_drop_gc_sync": parts[-3:], "after": clocks()}), flush=True)
    if leg == 0:
        time.sleep(pause)
        print(json.dumps({"paused_s": pause, "idle": clocks()}), flush=True)
