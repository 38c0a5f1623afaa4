#!/usr/bin/env python3
"""Experiment: what does restarting the inner COCG solve cost?  K x = b to 1e-11 in one run against iterative refinement
(outer residual, inner solve of the correction to a loose tolerance, repeated) -- the iteration-count side of a mixed-precision
inner loop.  Usage: exp_ir.py annulus-1M"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "helmholtz-x_amd"), os.path.join(ROOT, "tests")]
os.environ.setdefault("HX_QUIET", "1")
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from helmholtz_x.acoustic_matrices import AcousticMatrices  # noqa: E402
from helmholtz_x.amg import AMGSolver  # noqa: E402
from helmholtz_x.petsc import combine_csr  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "annulus-1M"
case = bench.build_case(bench.WORKLOADS[name])
P = case["P"]
M = AcousticMatrices(case["mesh"], case["facet_tags"], {11: {'Robin': P.R_outlet}}, case["c"])
K = combine_csr(M.A.shape, M.A.csr_terms + [(P.target * c, m) for c, m in M.B.csr_terms]
                + [(P.target ** 2 * c, m) for c, m in M.C.csr_terms])
n = K.shape[0]
solver = AMGSolver(K)
gen = torch.Generator(device=K.val.device)
gen.manual_seed(0)


def rhs():
    return torch.view_as_complex(torch.randn((n, 2), generator=gen, device=K.val.device, dtype=torch.float64))


x = torch.empty(n, dtype=torch.complex128, device=K.val.device)
solver.solve(rhs(), x)            # harvest
for trial in range(3):
    b = rhs()
    solver.solve(b, x)
    full = solver.last[0]
    line = ["one run: %d its (relres %.1e)" % (full, solver.last[1])]
    for inner_tol, label in ((1e-4, "1e-4"), (1e-6, "1e-6"), (3e-3, "3e-3")):
        xs = torch.zeros_like(b)
        r = b.clone()
        e = torch.empty_like(b)
        its, outers = 0, 0
        bn = float(torch.linalg.norm(b))
        while float(torch.linalg.norm(r)) > 1e-11 * bn and outers < 12:
            rel = float(torch.linalg.norm(r)) / bn
            tol = max(inner_tol, 1e-11 / rel * 0.5)
            solver.solve(r, e, relax=tol / 1e-11)
            its += solver.last[0]
            xs += e
            r = b.clone()
            K.spmv(xs, r, alpha=-1.0, beta=1.0)
            outers += 1
        line.append("IR(%s): %d its in %d outer steps (final %.1e)" % (label, its, outers, float(torch.linalg.norm(r)) / bn))
    print(" | ".join(line))
