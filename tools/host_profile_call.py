#!/usr/bin/env python3
"""cProfile of ONE fixed-point call (after a warm-up call): where the host time of the path goes.
Usage: python tools/host_profile_call.py [workload] [top]"""
import contextlib
import cProfile
import gc
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "helmholtz-x_amd"), os.path.join(ROOT, "tests")]
os.environ.setdefault("HX_QUIET", "1")
import torch  # noqa: E402

import bench  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "annulus-1M"
top = int(sys.argv[2]) if len(sys.argv) > 2 else 45
case = bench.build_case(bench.WORKLOADS[workload])
case["mesh"].assembler()
with contextlib.redirect_stdout(sys.stderr):
    E, m = bench.hot_path(case)
E.destroy()
E = m = None
gc.collect()
gc.freeze()
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
with contextlib.redirect_stdout(sys.stderr):
    E, m = bench.hot_path(case)
torch.cuda.synchronize()
pr.disable()
print("call under cProfile: %.1f ms" % (1e3 * (time.perf_counter() - t0)))
for key in ("tottime", "cumtime"):
    buf = io.StringIO()
    pstats.Stats(pr, stream=buf).strip_dirs().sort_stats(key).print_stats(top)
    print(buf.getvalue())
