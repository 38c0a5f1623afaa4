#!/usr/bin/env python3
"""Robustness sweep of the iterative inner solver: passive PEP solves of the synthetic annulus at the targets of the
reference's Modes/ runs (AnnularCombustor/Micca/fullAnnulus/active_modes.py: 1000 ... 11000 rad/s)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "helmholtz-x_amd"), os.path.join(ROOT, "tests")]
os.environ.setdefault("HX_QUIET", "1")
import torch  # noqa: E402

import bench  # noqa: E402
from helmholtz_x.acoustic_matrices import AcousticMatrices  # noqa: E402
from helmholtz_x.eigensolvers import pep_solver  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "annulus-260k"
case = bench.build_case(bench.WORKLOADS[name])
P = case["P"]
M = AcousticMatrices(case["mesh"], case["facet_tags"], {11: {'Robin': P.R_outlet}}, case["c"])
for target in (1000.0, 2000.0, 3225.12 + 481j, 5000.0, 9000.0, 10000.0, 11000.0):
    caches = ({}, {}, {})
    t0 = time.perf_counter()
    try:
        E = pep_solver(M.A, M.B, M.C, target, nev=2, _caches=caches)
        torch.cuda.synchronize()
        inner = next(v for v in caches[0].values() if hasattr(v, 'stats'))
        print("target %-18s -> omega %s  | %.2f s, %s" % (target, ["%.3f%+.3fj" % (E.getEigenvalue(i).real, E.getEigenvalue(i).imag) for i in range(2)],
              time.perf_counter() - t0, {k: v for k, v in inner.stats().items() if k in ("solves", "iters_per_solve", "setup_s", "recycled_pairs")}), flush=True)
    except Exception as exc:                                    # report, keep sweeping
        print("target %-18s FAILED: %s" % (target, exc), flush=True)
