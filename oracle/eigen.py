"""Oracle: flame transfer functions, shift-invert eigen-solves and the fixed-point iteration.

TEST INFRASTRUCTURE ONLY.  Restates helmholtz_x/flame_transfer_function.py:5-42,
helmholtz_x/eigensolvers.py:41-67 (EPS), :69-120 (PEP), :122-276 (fixed-point iteration) and
helmholtz_x/bloch_operator.py:12-68,104-111.  SLEPc's Krylov-Schur / TOAR with ST=sinvert are
replaced by ARPACK (implicitly restarted Arnoldi) applied to the explicit shift-and-invert
operator built from a SuperLU factorisation; both converge the same eigenpairs to rounding
(the reference runs with tol=1e-15).  The flame matrix D is materialised as an explicit sparse
matrix and A-D(omega_k) is re-factorised every iteration, exactly as the reference does.
"""
from math import factorial

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


# ----------------------------------------------------------------------------- FTFs
class NTau:
    """flame_transfer_function.py:5-14."""

    def __init__(self, n, tau):
        self.n, self.tau = n, tau

    def __call__(self, omega):
        return self.n * np.exp(1j * omega * self.tau)

    def derivative(self, omega):
        return self.n * (1j * self.tau) * np.exp(1j * omega * self.tau)


class StateSpace:
    """flame_transfer_function.py:16-42."""

    def __init__(self, S1, s2, s3, s4):
        self.A, self.b, self.c, self.d = S1, s2, s3, s4
        self.Id = np.eye(*S1.shape)

    def _eval(self, omega, k):
        omega = np.conj(omega)
        Mat = (-1j) ** k * factorial(k) * np.linalg.matrix_power(1j * omega * self.Id - self.A, -(k + 1))
        return np.dot(np.dot(self.c, Mat), self.b)

    def __call__(self, omega):
        H = self._eval(omega, 0) + self.d
        return np.conj(H[0][0])

    def derivative(self, omega):
        return np.conj(self._eval(omega, 1)[0][0])


# ----------------------------------------------------------------------------- eigen-solves
class OracleE:
    """Minimal result object: eigenvalues sorted by |lambda - sigma| (TARGET_MAGNITUDE)."""

    def __init__(self, kind, eigs, vecs, sigma):
        order = np.argsort(np.abs(eigs - sigma), kind='stable')
        self.kind = kind
        self.eigs = np.asarray(eigs)[order]
        self.vecs = np.asarray(vecs)[:, order]
        self.sigma = sigma

    def getEigenvalue(self, i):
        return self.eigs[i]

    def getConverged(self):
        return len(self.eigs)


def _ncv(nev, n):
    return min(max(2 * nev, nev + 15), n - 1)


def eps_solve(A, C, target, nev):
    """eigensolvers.py:41-67: A x = lambda (-C) x, lambda = omega^2, sinvert at target^2."""
    n = A.shape[0]
    sigma = complex(target) ** 2
    lu = spla.splu((A + sigma * C).tocsc())
    mC = (-C).tocsr()
    OP = spla.LinearOperator((n, n), matvec=lambda v: lu.solve(mC @ v), dtype=complex)
    k = min(nev, n - 2)
    theta, vecs = spla.eigs(OP, k=k, which='LM', ncv=_ncv(k, n), tol=0, v0=np.ones(n, dtype=complex),
                            maxiter=10 * n)
    return OracleE('eps', sigma + 1.0 / theta, vecs, sigma)


def pep_solve(A, B, C, target, nev):
    """eigensolvers.py:69-120: (A + lambda B + lambda^2 C) x = 0, sinvert at target.

    First companion linearisation z=[x; lambda x]; with K = A + sigma B + sigma^2 C:
      y1 = -K^{-1}[(B + sigma C) x1 + C x2],  y2 = x1 + sigma y1,   lambda = sigma + 1/theta."""
    n = A.shape[0]
    sigma = complex(target)
    Bm = B if B is not None else sp.csr_matrix((n, n), dtype=complex)
    lu = spla.splu((A + sigma * Bm + sigma ** 2 * C).tocsc())
    BsC = (Bm + sigma * C).tocsr()
    Cr = C.tocsr()

    def op(z):
        x1, x2 = z[:n], z[n:]
        y1 = -lu.solve(BsC @ x1 + Cr @ x2)
        return np.concatenate([y1, x1 + sigma * y1])

    OP = spla.LinearOperator((2 * n, 2 * n), matvec=op, dtype=complex)
    theta, vecs = spla.eigs(OP, k=nev, which='LM', ncv=_ncv(nev, 2 * n), tol=0,
                            v0=np.ones(2 * n, dtype=complex), maxiter=20 * n)
    return OracleE('pep', sigma + 1.0 / theta, vecs[:n], sigma)


class OracleFlame:
    """Explicit D_ij (flame_matrices.py:96-115)."""

    def __init__(self, D_ij, FTF, D_ij_adj=None):
        self.D_ij, self.D_ij_adj, self.FTF = D_ij, D_ij_adj, FTF

    def matrix(self, omega, problem_type='direct'):
        if problem_type == 'direct':
            return self.D_ij * self.FTF(omega)
        return self.D_ij_adj * np.conj(self.FTF(np.conj(omega)))


def fixed_point_iteration(A, B, C, D, target, nev=2, i=0, tol=1e-8, maxiter=50,
                          problem_type='direct', B_adj=None, trace=None):
    """eigensolvers.py:122-276.  Returns (E, omega_history)."""
    if problem_type == 'adjoint':
        B = B_adj
    use_pep = B is not None and B.nnz > 0
    omega = np.zeros(maxiter, dtype=complex)
    f = np.zeros(maxiter, dtype=complex)
    alpha = np.zeros(maxiter, dtype=complex)
    if use_pep:
        E = pep_solve(A, B, C, target, nev)
        omega[0] = E.getEigenvalue(i)
    else:
        E = eps_solve(A, C, target, nev)
        omega[0] = np.sqrt(E.getEigenvalue(i))
    alpha[0] = 0.5
    domega = 2 * tol
    k = -1
    hist = [omega[0]]
    while abs(domega) > tol:
        k += 1
        D_Mat = A - D.matrix(omega[k], problem_type)
        if use_pep:
            E = pep_solve(D_Mat, B, C, target, nev)
            f[k] = E.getEigenvalue(i)
        else:
            E = eps_solve(D_Mat, C, target, nev)
            f[k] = np.sqrt(E.getEigenvalue(i))
        if k != 0:
            alpha[k] = 1 / (1 - ((f[k] - f[k - 1]) / (omega[k] - omega[k - 1])))
        omega[k + 1] = alpha[k] * f[k] + (1 - alpha[k]) * omega[k]
        domega = omega[k + 1] - omega[k]
        hist.append(omega[k + 1])
        if trace:
            trace(k + 1, omega[k + 1], abs(domega))
    return E, np.array(hist)


# ----------------------------------------------------------------------------- Bloch
def bloch_maps(n, dofs_master, dofs_slave, N, b=1.0):
    """bloch_operator.py:12-68: BN (n x (n-nb)) and NB ((n-nb) x n); master/slave paired by order."""
    f_b = np.exp(b * 1j * 2 * np.pi / N)
    dofs_master = np.asarray(dofs_master)
    dofs_slave = np.asarray(dofs_slave)
    assert len(dofs_master) == len(dofs_slave)
    direct = np.setdiff1d(np.arange(n), dofs_master)
    ii = np.concatenate([direct, dofs_master])
    jj = np.concatenate([direct, dofs_slave])
    BN = sp.csr_matrix((np.ones(len(ii), dtype=complex), (ii, jj)), shape=(n, n)).tolil()
    NB = sp.csr_matrix((np.ones(len(ii), dtype=complex), (jj, ii)), shape=(n, n)).tolil()
    for m, s in zip(dofs_master, dofs_slave):
        BN[m, s] = f_b
        NB[s, m] = 1 / f_b
    keep = np.setdiff1d(np.arange(n), np.unique(dofs_master))
    return BN.tocsr()[:, keep], NB.tocsr()[keep, :]


def blochify(M, BN, NB):
    """bloch_operator.py:104-111."""
    return (NB @ M.tocsr() @ BN).tocsr()


def newton_solver(A, B, C, flame, dflame, init, i=0, tol=1e-3, maxiter=100):
    """eigensolvers.py:278-348 restated with dense LAPACK (small cases only).  ``flame(omega)`` / ``dflame(omega)``
    return the explicit D(omega) and dD/domega.  Reproduces the reference's quirks: relaxation *= 0.8 per iteration
    and ``vector_matrix_vector`` = PETSc ``y.dot(A x)`` = (A x)^H y, i.e. the conjugate of y^H A x (petsc4py_utils.py:57-87)."""
    import scipy.linalg as sla
    omega = [complex(init)]
    relaxation = 1.0
    domega = 2 * tol
    k = 0
    Cd = C.toarray()
    while abs(domega) > tol and k < maxiter:
        w = omega[-1]
        Bk = B if B is not None else sp.csr_matrix(A.shape, dtype=complex)
        L = (A + w * Bk + w ** 2 * C - flame(w)).toarray()
        dL = (Bk + 2 * w * C - dflame(w)).toarray()
        lam, vl, vr = sla.eig(L, Cd, left=True, right=True)
        order = np.argsort(np.abs(lam))
        j = order[i]
        x, y = vr[:, j], vl[:, j]
        num = np.vdot(dL @ x, y)          # (dL x)^H y
        den = np.vdot(Cd @ x, y)
        domega = -relaxation * lam[j] / (num / den)
        relaxation *= 0.8
        omega.append(w + domega)
        k += 1
    return np.array(omega)


def normalize_eigenvector(vec, C):
    """eigenvectors.py:11-64 + petsc4py_utils.py:100-111: phase fixed so that entry 0 is real positive, then scaled so
    that the BILINEAR integral  int p p dx = p^T C p = 1  (C = mass matrix; principal complex square root).  The
    reference's entry 0 is DOLFINx's dof 0, so results agree with its files up to the sign +-1 that p^T C p = 1 leaves."""
    v = np.asarray(vec, dtype=complex)
    v = v / (v[0] / abs(v[0]))
    meas = np.sqrt(v @ (C @ v))
    return v / meas


# ----------------------------------------------------------------------------- factored flame term (large supports)
class FactoredFlame:
    """D(omega) = FTF(omega) * a b^T kept as two vectors.  Used where the explicit block a (x) b of OracleFlame would not
    fit a sparse factorisation (P2 spaces, large 3-D Gaussian supports): mathematically the same operator, applied
    through the Sherman-Morrison identity on top of ONE factorisation of the passive shifted operator."""

    def __init__(self, a, b, FTF):
        self.a, self.b, self.FTF = np.asarray(a, dtype=complex), np.asarray(b, dtype=complex), FTF


def pep_solve_factored(A, B, C, target, nev, flame=None, omega=None, lu=None):
    """pep_solve for (A - FTF(omega) a b^T) + lambda B + lambda^2 C with the rank-1 term handled by Sherman-Morrison:
    (K - f a b^T)^-1 r = y + w * f (b.y) / (1 - f (b.w)),  y = K^-1 r,  w = K^-1 a."""
    n = A.shape[0]
    sigma = complex(target)
    Bm = B if B is not None else sp.csr_matrix((n, n), dtype=complex)
    if lu is None:
        lu = spla.splu((A + sigma * Bm + sigma ** 2 * C).tocsc())
    BsC = (Bm + sigma * C).tocsr()
    Cr = C.tocsr()
    if flame is not None and omega is not None:
        f = flame.FTF(omega)
        w = lu.solve(flame.a)
        denom = 1.0 - f * (flame.b @ w)

        def ksolve(r):
            y = lu.solve(r)
            return y + w * (f * (flame.b @ y) / denom)
    else:
        ksolve = lu.solve

    def op(z):
        x1, x2 = z[:n], z[n:]
        y1 = -ksolve(BsC @ x1 + Cr @ x2)
        return np.concatenate([y1, x1 + sigma * y1])

    OP = spla.LinearOperator((2 * n, 2 * n), matvec=op, dtype=complex)
    theta, vecs = spla.eigs(OP, k=nev, which='LM', ncv=_ncv(nev, 2 * n), tol=0,
                            v0=np.ones(2 * n, dtype=complex), maxiter=20 * n)
    return OracleE('pep', sigma + 1.0 / theta, vecs[:n], sigma), lu


def fixed_point_iteration_factored(A, B, C, flame, target, nev=2, i=0, tol=1e-8, maxiter=50):
    """eigensolvers.py:197-259 with the flame term factored (see FactoredFlame); same update rule."""
    omega = np.zeros(maxiter, dtype=complex)
    f = np.zeros(maxiter, dtype=complex)
    alpha = np.zeros(maxiter, dtype=complex)
    E, lu = pep_solve_factored(A, B, C, target, nev)
    omega[0] = E.getEigenvalue(i)
    alpha[0] = 0.5
    domega = 2 * tol
    k = -1
    hist = [omega[0]]
    while abs(domega) > tol:
        k += 1
        E, _ = pep_solve_factored(A, B, C, target, nev, flame, omega[k], lu)
        f[k] = E.getEigenvalue(i)
        if k != 0:
            alpha[k] = 1 / (1 - ((f[k] - f[k - 1]) / (omega[k] - omega[k - 1])))
        omega[k + 1] = alpha[k] * f[k] + (1 - alpha[k]) * omega[k]
        domega = omega[k + 1] - omega[k]
        hist.append(omega[k + 1])
    return E, np.array(hist)
