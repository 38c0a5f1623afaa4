#!/usr/bin/env python3
"""Host profile (cProfile, by internal time) of one warm fixed-point call of a bench workload.  usage: call_profile.py workload st_type"""
import cProfile
import os
import pstats
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "helmholtz-x_amd"), os.path.join(ROOT, "tests")]
os.environ.setdefault("HX_QUIET", "1")
import torch  # noqa: E402
import bench  # noqa: E402

wl, st = sys.argv[1], sys.argv[2]
shape = bench.WORKLOADS[wl]
case = bench.build_case_real(shape[1]) if shape[0] == "real" else bench.build_case(shape)
for trial in range(3):
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    E, _ = bench.hot_path(case, st_type=st, nev=4, tol=1e-3)
    torch.cuda.synchronize()
    pr.disable()
    E.destroy()
    if trial == 2:
        pstats.Stats(pr).sort_stats("tottime").print_stats(28)
