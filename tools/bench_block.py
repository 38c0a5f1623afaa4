#!/usr/bin/env python3
"""Block (4 right-hand sides) against single-vector inner solves on one operator: wall time per solve and per iteration.
Usage: bench_block.py <workload> [numbering [jitter]]      e.g.  bench_block.py micca-real-r2   |   annulus-5M rcm 0.2
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel averages (k_sell_blk / k_csr_blk / k_bdots / k_bcg_update)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "helmholtz-x_amd"), os.path.join(ROOT, "tests")]
os.environ.setdefault("HX_QUIET", "1")
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from helmholtz_x.acoustic_matrices import AcousticMatrices  # noqa: E402
from helmholtz_x.amg import AMGSolver  # noqa: E402
from helmholtz_x.petsc import combine_csr, to_device  # noqa: E402

name = sys.argv[1]
numbering = sys.argv[2] if len(sys.argv) > 2 else "native"
jitter = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
case = bench.build_case(bench.WORKLOADS[name], numbering, jitter)
P = case["P"]
M = AcousticMatrices(case["mesh"], case["facet_tags"], {11: {'Robin': P.R_outlet}}, case["c"])
K = combine_csr(M.A.shape, M.A.csr_terms + [(P.target * c, m) for c, m in M.B.csr_terms]
                + [(P.target ** 2 * c, m) for c, m in M.C.csr_terms])
n = K.shape[0]
solver = AMGSolver(K)
rng = np.random.default_rng(0)
B = to_device(rng.standard_normal((8, n)) + 1j * rng.standard_normal((8, n)))
X = torch.zeros_like(B)
for c in range(2):
    solver.solve(B[c], X[c])            # GMRES + harvest, then the first COCG solve (prepares the deflation pairs)
torch.cuda.synchronize()
t0 = time.perf_counter()
its = []
for c in range(4):
    solver.solve(B[c], X[c])
    its.append(solver.last[0])
torch.cuda.synchronize()
t_single = time.perf_counter() - t0
Xb = torch.zeros_like(B)
solver.solve_block(B[:4], Xb[:4])       # (allocates the block work vectors)
torch.cuda.synchronize()
b0 = solver.stats()["block_iters"]
t0 = time.perf_counter()
solver.solve_block(B[:4], Xb[:4])
torch.cuda.synchronize()
t_block = time.perf_counter() - t0
bi = solver.stats()["block_iters"] - b0
err = max(float(torch.linalg.norm(Xb[c] - X[c]) / torch.linalg.norm(X[c])) for c in range(4))
print("n=%d  single: 4 solves %.1f ms, iterations %s (%.3f ms per iteration)" % (n, 1e3 * t_single, its, 1e3 * t_single / sum(its)))
print("      block : 4 columns %.1f ms, %d block iterations (%.3f ms per block iteration = %.3f per column)  max rel. diff %.1e"
      % (1e3 * t_block, bi, 1e3 * t_block / bi, 1e3 * t_block / bi / 4, err))
print("      block / single time: %.2f" % (t_block / t_single))
