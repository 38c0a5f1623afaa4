#!/usr/bin/env python3
"""BASELINE config 2: 2-D manufactured box, (nx, ny) = (640, 160) -> 103,201 DOF, Robin top wall, PEP nev=2,
checked against the analytic dispersion relation (manufacturedHelmholtz.py:36-53).  Prints timings."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "helmholtz-x_amd"), os.path.join(ROOT, "tests")]
os.environ.setdefault("HX_QUIET", "1")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cases import dispersion_root  # noqa: E402
from helmholtz_x.acoustic_matrices import AcousticMatrices  # noqa: E402
from helmholtz_x.dolfinx_utils import RectangleSetup  # noqa: E402
from helmholtz_x.eigensolvers import pep_solver  # noqa: E402
from helmholtz_x.parameters_utils import c_uniform  # noqa: E402

nx, ny = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 160)
t0 = time.perf_counter()
mesh, subdomains, facet_tags = RectangleSetup(nx, ny, 0.4, 0.1)
c = c_uniform(mesh, 450.0)
print("mesh %d DOF, %.2f s" % (mesh.num_vertices, time.perf_counter() - t0))
omega = 600 * 2 * np.pi
for Z in np.linspace(-10, 10, 10)[:5]:
    t1 = time.perf_counter()
    M = AcousticMatrices(mesh, facet_tags, {4: {'Robin': (Z - 1) / (Z + 1)}}, c)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    caches = ({}, {}, {})
    E = pep_solver(M.A, M.B, M.C, omega, nev=2, st_type=os.environ.get("HX_ST"), _caches=caches)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    omega = E.getEigenvalue(0)
    k = dispersion_root(omega / 450.0, Z)
    inner = next(v for v in caches[0].values() if hasattr(v, 'stats'))
    print("Z=%6.2f f=%.6f%+.6fj analytic=%.6f%+.6fj rel=%.1e | assemble %.3f s, solve %.3f s, %s, krylov %s" % (
        Z, (omega / 2 / np.pi).real, (omega / 2 / np.pi).imag, (450 * k / 2 / np.pi).real, (450 * k / 2 / np.pi).imag,
        abs(omega - 450 * k) / abs(omega), t2 - t1, t3 - t2, inner.stats(), {a: b for a, b in caches[2].items() if not a.startswith(('start', '_'))}))
