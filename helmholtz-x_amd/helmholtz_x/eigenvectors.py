"""Eigenvector extraction / normalisation (reference: helmholtz_x/eigenvectors.py:11-64).

Only ``normalize_eigenvector`` is on the path: phase fixed on entry 0, then scaled so that
the BILINEAR integral int p p dx = 1 (not the sesquilinear norm).  The integral is p^T C p with the P1 mass matrix,
evaluated on the device (hx_csr_spmv_z + hx_zpanel_dot)."""
import numpy as np
import torch

from .fem import Function, FunctionSpace
from .petsc import panel_dot
from .petsc4py_utils import FixSign, matrix_vector
from .slepc import EPS, PEP


def normalize_eigenvector(mesh, obj, i, absolute=False, degree=1, which='right', BlochRemapper=None, matrices=None,
                          print_eigs=True):
    A = obj.getOperators()[0]
    vr, vi = A.createVecs()
    if isinstance(obj, EPS):
        omega = np.sqrt(obj.getEigenvalue(i))
        if which == 'right':
            obj.getEigenvector(i, vr, vi)
        elif which == 'left':
            obj.getLeftEigenvector(i, vr, vi)
    elif isinstance(obj, PEP):
        omega = obj.getEigenpair(i, vr, vi)
    else:
        raise TypeError("expected an EPS or PEP solver object")
    if BlochRemapper:
        vr = matrix_vector(BlochRemapper, vr)
    V = matrices.V if matrices else FunctionSpace(mesh, "CG", degree)
    FixSign(vr)
    C = mesh.assembler().stiffness_mass(None, want_A=False)[1].csr_part()
    Cp = torch.empty_like(vr.t)
    C.spmv(vr.t, Cp)
    meas = np.sqrt(panel_dot(torch.conj_physical(vr.t).reshape(1, -1), 1, Cp)[0])     # sqrt(p^T C p)
    temp = vr.array / meas
    if absolute:
        temp = np.abs(temp)
        temp = temp / temp.max()
    p = Function(V, "p")
    p.x.array[:] = temp
    if print_eigs:
        print(f"Eigenvalue-> {omega:.6f} | Eigenfrequency-> {omega/(2*np.pi):.6f}\n ")
    return omega, p
