#!/usr/bin/env python3
"""A/B of run-time options of libhx INSIDE one process (hx_set_option takes effect at the next library call): the case is built
once, the configurations are visited round-robin several times, the best and the median time of a whole hot-path call are kept per
configuration -- no process-to-process or box-to-box spread in the comparison.

  python tools/ab_inprocess.py annulus-1M "amg_fold_diag=1" "amg_fold_diag=0" "amg_fold_diag=1,cocg_pq_in_product=0" [--rounds 4]
(names starting with HX_ are environment switches of the Python layer, set for the calls of their configuration only)
"""
import gc
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "helmholtz-x_amd"), os.path.join(ROOT, "tests")]
os.environ.setdefault("HX_QUIET", "1")
import torch  # noqa: E402

import bench  # noqa: E402
from helmholtz_x import _hx  # noqa: E402

argv = sys.argv[1:]
rounds = 4
if "--rounds" in argv:
    k = argv.index("--rounds")
    rounds = int(argv[k + 1])
    del argv[k:k + 2]
workload, configs = argv[0], argv[1:]
assert configs and all("=" in c for c in configs), "configurations are comma-separated name=value lists"
case = bench.build_case(bench.WORKLOADS[workload])
case["mesh"].assembler()


def call():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    E, _ = bench.hot_path(case, None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    inner = next(v for v in E.inner_cache.values() if hasattr(v, "stats"))
    its = inner.stats()["gmres_iters"]
    w = E.getEigenvalue(0)
    E.destroy()
    gc.collect()
    return dt, its, w


def apply(cfg, on):
    for kv in cfg.split(","):
        if kv:
            k, v = kv.split("=")
            if k.startswith("HX_"):                  # a switch of the Python layer: read from the environment when the solver is built
                if on:
                    os.environ[k] = v
                else:
                    os.environ.pop(k, None)
            else:
                _hx.set_option(k, v if on else None)


for _ in range(2):
    call()
gc.freeze()
times = {c: [] for c in configs}
info = {}
for r in range(rounds):
    for c in configs:
        apply(c, True)
        dt, its, w = call()
        apply(c, False)
        times[c].append(dt)
        info[c] = (its, w)
print("workload %s, %d rounds (round-robin, one process)" % (workload, rounds))
for c in configs:
    t = times[c]
    print("  %-50s best %.1f ms  median %.1f ms  inner iterations %d  omega %.6f%+.6fj" % (
        c, 1e3 * min(t), 1e3 * statistics.median(t), info[c][0], info[c][1].real, info[c][1].imag))
