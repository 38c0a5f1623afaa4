"""Krylov-Schur (thick-restart Arnoldi) driver for the shift-and-invert operator.

Replaces the part of SLEPc the reference reaches through ``E.solve()`` (eigensolvers.py:62,113):
EPS ``krylovschur`` / PEP ``toar`` with ST ``sinvert`` and ``TARGET_MAGNITUDE``.  The n-length work
(operator application, classical Gram-Schmidt with one re-orthogonalisation, basis rotation at restart)
is delegated to a *basis* object -- on the product path ``DeviceBasis`` (libhx kernels, see slepc.py);
the small projected problem (<= ncv x ncv) is dense host LAPACK, as in SLEPc.

Basis interface (columns are Krylov vectors):
    m                      number of columns available (ncv + 1)
    set_start(j)           fill column j with the start vector and normalise it
    expand(j) -> (h, beta) w = OP v_j ; CGS2 against columns 0..j ; h = V^H w (length j+1), beta = ||w||;
                           stores w/beta in column j+1
    rotate(Q, k)           columns[0:k] = columns[0:m_active] @ Q  (Q is m_active x k), and move the
                           residual column m_active to column k
"""
import numpy as np
import scipy.linalg as sla


class KrylovSchurResult:
    def __init__(self, theta, nconv, its, resid):
        self.theta, self.nconv, self.its, self.resid = theta, nconv, its, resid


def default_ncv(nev):
    """SLEPc default: ncv = max(2*nev, nev+15)."""
    return max(2 * nev, nev + 15)


def krylov_schur(basis, nev, ncv, tol=1e-15, max_it=100, keep_fraction=0.5, relaxed_tol=1e-12):
    """Largest-magnitude eigenvalues theta of OP (theta = 1/(lambda - sigma): closest to the target).

    Returns KrylovSchurResult with Ritz values sorted by |theta| descending; the first ``nconv`` have
    residual estimate <= tol*|theta|.  After return, basis columns 0..nconv-1 hold the converged Ritz
    VECTORS (not just Schur vectors)."""
    m = ncv
    assert basis.m >= m + 1
    S = np.zeros((m + 1, m), dtype=complex)      # OP V_m = V_{m+1} S
    basis.set_start(0)
    k = 0                                         # columns 0..k-1 are kept Schur vectors
    its = 0
    nconv = 0
    prev_lead = np.inf
    while True:
        its += 1
        for j in range(k, m):
            h, beta = basis.expand(j)
            S[:j + 1, j] = h
            S[j + 1, j] = beta
        Sm = S[:m, :m]
        b = S[m, :m].copy()                       # residual row: OP V_m = V_m S_m + v_{m+1} b
        theta, Y = sla.eig(Sm)
        order = np.argsort(-np.abs(theta), kind="stable")
        theta, Y = theta[order], Y[:, order]
        Y = Y / np.linalg.norm(Y, axis=0, keepdims=True)
        resid = np.abs(b @ Y)
        conv = resid <= tol * np.abs(theta)
        nconv = 0
        while nconv < m and conv[nconv]:
            nconv += 1
        if nconv >= nev or its >= max_it:
            break
        # stagnation guard: the estimate cannot always reach tol = 1e-15 in double precision; once the wanted
        # pairs sit below relaxed_tol and stopped improving, accept them (reported through nconv).
        lead = float(np.max(resid[:nev] / np.abs(theta[:nev])))
        if its >= 2 and lead <= relaxed_tol and lead >= 0.5 * prev_lead:
            nconv = int(np.sum(np.cumprod(resid <= relaxed_tol * np.abs(theta))))
            break
        prev_lead = lead
        # ---- thick restart: keep the leading k Ritz directions as an orthonormal (Schur) basis
        k = min(max(nconv + 1, nev + int(keep_fraction * (m - nev))), m - 1)
        Q, R = np.linalg.qr(Y[:, :k])
        T = R @ np.diag(theta[:k]) @ np.linalg.inv(R)          # upper triangular: Q^H S_m Q
        basis.rotate(Q, k)
        S[:, :] = 0.0
        S[:k, :k] = T
        S[k, :k] = b @ Q
    # ---- Ritz vectors of the converged (or best available) pairs into the leading columns: Y has unit columns and
    # V_m is orthonormal, so V_m Y[:, i] are unit-norm Ritz vectors.
    nout = max(nconv, min(nev, m))
    basis.rotate(Y[:, :nout], nout)
    return KrylovSchurResult(theta[:nout], nconv, its, resid[:nout])
