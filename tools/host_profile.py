#!/usr/bin/env python3
"""cProfile of one hot-path call (after a warm-up call): where the HOST spends the call.  Usage: host_profile.py annulus-1M"""
import cProfile
import io
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "helmholtz-x_amd")]
os.environ.setdefault("HX_QUIET", "1")
import contextlib  # noqa: E402

import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "annulus-1M"
case = bench.build_case(bench.WORKLOADS[name])
with contextlib.redirect_stdout(sys.stderr):
    bench.hot_path(case, None)
    pr = cProfile.Profile()
    pr.enable()
    bench.hot_path(case, None)
    pr.disable()
for key in ("cumulative", "tottime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print(s.getvalue())
