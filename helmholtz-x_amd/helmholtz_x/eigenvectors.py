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
    shard = getattr(obj, "shard", None)
    if BlochRemapper and shard is not None and getattr(mesh, "shard", None) is not None:
        # row-sharded Bloch run: the solver lives on the REDUCED numbering (obj.shard); the remapper gives this rank's owned
        # entries of the full vector from its owned reduced entries (both partners of a master/slave pair are owned by one
        # rank), and the normalisation below runs on the full distributed mesh
        from .petsc import Vec
        full = mesh.shard
        x_red, x_own = Vec(vr.t[:shard.n_own].clone()), Vec.zeros(full.n_own)
        BlochRemapper.mult(x_red, x_own)
        vr = Vec.zeros(full.n_ext)
        vr.t[:full.n_own].copy_(x_own.t)
        shard = full
    elif BlochRemapper:
        vr = matrix_vector(BlochRemapper, vr)
    V = matrices.V if matrices else FunctionSpace(mesh, "CG", degree)
    if shard is not None:
        return omega, _normalize_distributed(mesh, shard, V, vr, omega, absolute, print_eigs)
    FixSign(vr)
    C = V.assembler().stiffness_mass(None, want_A=False)[1].csr_part()
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


def _normalize_distributed(mesh, shard, V, vr, omega, absolute, print_eigs):
    """The same conventions on a row-sharded eigenvector (vr: this rank's owned entries, room for the ghosts): the phase is
    fixed on GLOBAL entry 0 (its owner contributes it to a one-number all-reduce, as the reference's FixSign does with a
    broadcast, petsc4py_utils.py:100-111), the bilinear integral int p p dx is the sum of the ranks' p_own^T (C p)_own with the
    owned rows of the mass matrix (one halo of p, one all-reduce; eigenvectors.py:47,54 are allreduces too).  Returns a
    Function on the rank's local mesh (owned + ghost entries)."""
    n_own = shard.n_own
    first = torch.zeros(1, dtype=torch.complex128, device=vr.t.device)
    if shard.r0 == 0:
        first[0] = vr.t[0]
    shard.all_reduce_(first)
    x0 = complex(first[0].item())
    vr.scale(1.0 / (x0 / abs(x0)))
    shard.halo(vr.t)
    C = V.assembler().stiffness_mass(None, want_A=False)[1].csr_part()          # owned rows over [owned | ghost]
    Cp = torch.empty(n_own, dtype=torch.complex128, device=vr.t.device)
    C.spmv(vr.t, Cp)
    acc = torch.zeros(1, dtype=torch.complex128, device=vr.t.device)
    acc[0] = torch.dot(vr.t[:n_own], Cp)                                        # bilinear: no conjugate
    shard.all_reduce_(acc)
    meas = np.sqrt(complex(acc[0].item()))
    temp = vr.array / meas
    if absolute:
        temp = np.abs(temp)
        mx = torch.tensor([float(temp[:n_own].max())], dtype=torch.float64, device=vr.t.device)
        shard.all_reduce_(mx, op="max")
        temp = temp / float(mx.item())
    p = Function(V, "p")
    p.x.array[:] = temp
    if print_eigs and shard.rank == 0:
        print(f"Eigenvalue-> {omega:.6f} | Eigenfrequency-> {omega/(2*np.pi):.6f}\n ")
    return p


def normalize_adjoint(omega_dir, p_dir, p_adj, matrices, D=None):
    """Scale the adjoint mode so that  <p_adj, dL/domega p_dir> = 1  (reference: eigenvectors.py:125-177; the
    normalisation the shape-derivative consumer expects).  dL/domega = [B] + 2 omega C [- D'(omega)], applied lazily on
    the device (the flame derivative stays factored); the bracket is the reference's ``vector_matrix_vector``, i.e.
    PETSc's conjugating dot."""
    from .petsc4py_utils import multiply, vector_matrix_vector
    from .solver_utils import info
    info("- Normalizing the adjoint eigenvector to calculate shape derivatives..")
    B = matrices.B
    p_dir_vec, p_adj_vec = p_dir.x.petsc_vec, p_adj.x.petsc_vec
    dL_domega = matrices.C * (2 * omega_dir)
    if B:
        dL_domega = B + dL_domega
    if D:
        dL_domega = dL_domega - D.get_derivative(omega_dir)
    meas = vector_matrix_vector(p_adj_vec, dL_domega, p_dir_vec)
    p_adj_vec = multiply(p_adj_vec, 1 / meas)
    p_adj1 = p_adj
    p_adj1.name = "p_adj"
    p_adj1.x.array[:] = p_adj_vec.getArray()          # (Function.x.petsc_vec is a device COPY here, not a view)
    integral = vector_matrix_vector(p_adj1.x.petsc_vec, dL_domega, p_dir_vec)
    print("! Normalization Check: ", integral)
    return p_adj1
