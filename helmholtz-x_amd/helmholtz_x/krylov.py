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
import contextlib

import numpy as np
import scipy.linalg as sla

try:                                   # the projected problems are tiny: keep the BLAS pool from spinning on them
    from threadpoolctl import ThreadpoolController
    _blas = ThreadpoolController()

    def _small_dense():
        return _blas.limit(limits=1, user_api="blas")
except Exception:                      # pragma: no cover - threadpoolctl is optional
    def _small_dense():
        return contextlib.nullcontext()


class KrylovSchurResult:
    def __init__(self, theta, nconv, its, resid):
        self.theta, self.nconv, self.its, self.resid = theta, nconv, its, resid


def default_ncv(nev):
    """SLEPc default: ncv = max(2*nev, nev+15)."""
    return max(2 * nev, nev + 15)


def _tell(basis, theta, resid, nev):
    """Let the basis relax the accuracy of its operator applications (inexact Krylov): the largest relative residual
    estimate among the wanted pairs bounds how much of the remaining Krylov directions can still enter them."""
    hook = getattr(basis, "set_inner_accuracy", None)
    if hook is not None and len(theta):
        k = min(nev, len(theta))
        hook(float(np.max(resid[:k] / np.maximum(np.abs(theta[:k]), 1e-300))), nev)


def _ritz(S, mact, tol):
    """Ritz pairs of the leading mact x mact block, sorted by |theta| descending, with residual estimates."""
    with _small_dense():
        theta, Y = sla.eig(S[:mact, :mact])
    order = np.argsort(-np.abs(theta), kind="stable")
    theta, Y = theta[order], Y[:, order]
    Y = Y / np.linalg.norm(Y, axis=0, keepdims=True)
    resid = np.abs(S[mact, :mact] @ Y)
    nconv = int(np.sum(np.cumprod(resid <= tol * np.abs(theta))))
    return theta, Y, resid, nconv


def krylov_schur(basis, nev, ncv, tol=1e-15, max_it=100, keep_fraction=0.5, relaxed_tol=1e-12, check_every=1,
                 n_extra=0):
    """Largest-magnitude eigenvalues theta of OP (theta = 1/(lambda - sigma): closest to the target).

    Returns KrylovSchurResult with Ritz values sorted by |theta| descending; the first ``nconv`` have
    residual estimate <= tol*|theta|.  After return, basis columns 0..nout-1 hold unit-norm Ritz VECTORS.
    Convergence is also tested inside a cycle every ``check_every`` expansions (the projected problem is tiny; every
    step by default: measured 85 instead of 88 operator applications per 5 M-DOF fixed-point call),
    so a warm-started solve stops as soon as the wanted pairs are there instead of filling all ncv columns.
    ``n_extra`` further (not necessarily converged) Ritz vectors are also left in the basis: good start directions
    for a following solve of a nearby problem that wants more pairs."""
    import os
    check_every = int(os.environ.get("HX_KRYLOV_CHECK_EVERY", check_every))      # experiment hook
    trace = bool(os.environ.get("HX_KRYLOV_TRACE"))                              # diagnostics: residual history
    m = ncv
    assert basis.m >= m + 1
    S = np.zeros((m + 1, m), dtype=complex)      # OP V_j = V_{j+1} S[:j+1, :j]
    basis.set_start(0)
    k = 0                                         # columns 0..k-1 are kept Schur vectors
    its = 0
    prev_lead = np.inf
    while True:
        its += 1
        mact = m
        for j in range(k, m):
            h, beta = basis.expand(j)
            S[:j + 1, j] = h
            S[j + 1, j] = beta
            cols = j + 1
            if cols < m and cols >= k + nev + 1 and (cols - k) % check_every == 0:
                th_c, _, res_c, nc_c = _ritz(S, cols, tol)
                if trace:
                    print("[krylov_schur] nev %d cols %2d  rel.resid %s" % (
                        nev, cols, " ".join("%.1e" % r for r in (res_c / np.abs(th_c))[:nev + 2])), flush=True)
                _tell(basis, th_c, res_c, nev)
                if nc_c >= nev:
                    mact = cols
                    break
        theta, Y, resid, nconv = _ritz(S, mact, tol)
        _tell(basis, theta, resid, nev)
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
        b = S[m, :m].copy()
        basis.rotate(Q, k)
        S[:, :] = 0.0
        S[:k, :k] = T
        S[k, :k] = b @ Q
    # ---- Ritz vectors of the converged (or best available) pairs into the leading columns: Y has unit columns and
    # V is orthonormal, so V Y[:, i] are unit-norm Ritz vectors.
    nout = min(max(nconv, min(nev, mact)) + n_extra, mact)
    basis.rotate(Y[:, :nout], nout)
    return KrylovSchurResult(theta[:nout], nconv, its, resid[:nout])
