#!/usr/bin/env python3
"""Where the set-up of the 5 M-DOF hierarchy spends its host time (cProfile of the second build).  Run from the tree's root."""
import cProfile
import os
import pstats
import sys
ROOT = os.getcwd()
sys.path[:0] = [ROOT, os.path.join(ROOT, "helmholtz-x_amd"), os.path.join(ROOT, "tests")]
os.environ.setdefault("HX_QUIET", "1")
import torch  # noqa: E402
import bench  # noqa: E402
from helmholtz_x.acoustic_matrices import AcousticMatrices  # noqa: E402
from helmholtz_x.amg import AMGSolver  # noqa: E402
from helmholtz_x.petsc import combine_csr  # noqa: E402

case = bench.build_case(bench.WORKLOADS["annulus-5M"])
P = case["P"]
M = AcousticMatrices(case["mesh"], case["facet_tags"], {11: {'Robin': P.R_outlet}}, case["c"])
K = combine_csr(M.A.shape, M.A.csr_terms + [(P.target * c, m) for c, m in M.B.csr_terms] + [(P.target ** 2 * c, m) for c, m in M.C.csr_terms])
K.symmetric = True
for trial in range(3):
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    S = AMGSolver(K)
    torch.cuda.synchronize()
    pr.disable()
    print("build %d: setup_s %.4f" % (trial, S.setup_s))
    if trial == 2:
        pstats.Stats(pr).sort_stats("tottime").print_stats(18)
    S.destroy()
