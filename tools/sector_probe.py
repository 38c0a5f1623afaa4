#!/usr/bin/env python3
"""Diagnostic: GMRES iterations per inner solve on a 1/16 sector WITHOUT Bloch reduction (Neumann cut planes), to separate
the effect of cell anisotropy from the Bloch phase rows.  Usage: sector_probe.py nr ntheta nz"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "helmholtz-x_amd"), os.path.join(ROOT, "tests")]
os.environ.setdefault("HX_QUIET", "1")
from cases import Micca  # noqa: E402
from helmholtz_x.acoustic_matrices import AcousticMatrices  # noqa: E402
from helmholtz_x.eigensolvers import eps_solver  # noqa: E402
from helmholtz_x.synthetic import annulus, micca_sound_speed  # noqa: E402

shape = tuple(int(a) for a in sys.argv[1:4])
half = np.pi / 16
mesh, sub, ft = annulus(*shape, n_flames=1, sector=(-half, half))
M = AcousticMatrices(mesh, ft, {11: 'Neumann', 12: 'Neumann', 13: 'Neumann'}, micca_sound_speed(mesh))
caches = ({}, {}, {})
E = eps_solver(M.A, M.C, Micca.target, nev=2, st_type="amg", _caches=caches)
inner = next(v for v in caches[0].values() if hasattr(v, "stats"))
print(shape, mesh.num_vertices, "DOF:", inner.stats(), [np.sqrt(E.getEigenvalue(i)) for i in range(2)])
