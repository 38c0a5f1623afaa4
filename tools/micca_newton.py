#!/usr/bin/env python3
"""AnnularCombustor/Micca/fullAnnulus/active_newton.py on the reference's mesh through the drop-in API (diagnostic run:
reproduces the reference's Newton log, including its decaying relaxation).  Usage: micca_newton.py [i] [st_type]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "helmholtz-x_amd"), os.path.join(ROOT, "tests")]
os.environ.setdefault("HX_QUIET", "1")
from test_gpu_parity import _micca_full_problem  # noqa: E402
from helmholtz_x.eigensolvers import newtonSolver  # noqa: E402

i = int(sys.argv[1]) if len(sys.argv) > 1 else 0
st = sys.argv[2] if len(sys.argv) > 2 else None
matrices, D = _micca_full_problem(os.path.join(ROOT, "tests", "golden"))
t0 = time.perf_counter()
omega, p = newtonSolver(matrices, D, 3260 + 460j, i=i, nev=2, tol=1e-2, st_type=st)
print("RESULT i=%d omega=%r  %.1f s" % (i, complex(omega), time.perf_counter() - t0))
