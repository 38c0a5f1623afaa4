#!/usr/bin/env python3
"""cProfile of one hot-path pass (host-side view: where the wall time goes outside the kernels)."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "helmholtz-x_amd"), os.path.join(ROOT, "tests")]
os.environ.setdefault("HX_QUIET", "1")
import torch  # noqa: E402

import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "annulus-1M"
st = sys.argv[2] if len(sys.argv) > 2 else "amg"
case = bench.build_case(bench.WORKLOADS[name])
case["mesh"].assembler()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
E, M = bench.hot_path(case, st)
torch.cuda.synchronize()
pr.disable()
ps = pstats.Stats(pr).sort_stats("cumulative")
ps.print_stats(45)
