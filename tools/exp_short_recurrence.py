#!/usr/bin/env python3
"""Experiment (round 2): can the inner solve of K use a SHORT-RECURRENCE Krylov method instead of flexible GMRES?

K = A + sigma B + sigma^2 C is complex SYMMETRIC (K^T = K), and the V-cycle with equal pre / post Chebyshev smoothing and
R = P^T is a complex symmetric operator too (up to complex64 rounding), so COCG / COCR apply.  They need no Gram-Schmidt
panel: per iteration 1 V-cycle + 1 operator product + a handful of vector passes, against CGS2 on ~30 panel rows.
This script measures iteration counts (same V-cycle handle, same tolerance) of
  gmres        hx_gmres as shipped (first solve = no recycled pairs, later solves = 6 recycled pairs)
  cocg / cocr  plain
  cocg-defl    deflated with k harmonic Ritz vectors of a 40-step Arnoldi run (the same vectors hx_gmres recycles)
Python/torch vector ops: timing is NOT representative, only the counts are.

usage: python tools/exp_short_recurrence.py [annulus-1M] [rtol]
"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "helmholtz-x_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ.setdefault("HX_QUIET", "1")

import numpy as np
import torch

import bench
from helmholtz_x import _hx
from helmholtz_x.acoustic_matrices import AcousticMatrices
from helmholtz_x.amg import AMGSolver, harmonic_ritz_recycle
from helmholtz_x.petsc import combine_csr


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "annulus-1M"
    rtol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-11
    case = bench.build_case(bench.WORKLOADS[wl])
    P, mesh = case["P"], case["mesh"]
    mesh.assembler().rebuild_maps()
    mats = AcousticMatrices(mesh, case["facet_tags"], {11: {'Robin': P.R_outlet}}, case["c"])
    A, B, Cm = mats.A, mats.B, mats.C
    sig = P.target
    K = combine_csr(A.shape, A.csr_terms + [(sig * c, m) for c, m in B.csr_terms]
                    + [(sig ** 2 * c, m) for c, m in Cm.csr_terms])
    n = K.shape[0]
    solver = AMGSolver(K)
    print("n = %d, levels %s" % (n, [lv["n"] for lv in solver.levels]), flush=True)
    dev = K.val.device
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    rhs = []
    for _ in range(3):
        v = torch.view_as_complex(torch.randn((n, 2), generator=gen, device=dev, dtype=torch.float64))
        b = torch.empty_like(v)
        Cm.csr_part().spmv(v, b)                 # the right-hand sides of the path are C x (smooth data)
        rhs.append(b / torch.linalg.vector_norm(b))
    one, zero = _hx.z2(1.0), _hx.z2(0.0)

    def Kx(x):
        y = torch.empty_like(x)
        _hx.check(_hx.lib.hx_sell_spmv_z(solver.sell, _hx.ptr(x), _hx.ptr(y), one, zero, _hx.stream()))
        return y

    def Mx(x):
        y = torch.empty_like(x)
        _hx.check(_hx.lib.hx_amg_vcycle(solver.handle, _hx.ptr(x), _hx.ptr(y), _hx.stream()))
        return y

    def bdot(x, y):                               # bilinear (unconjugated) product
        return torch.dot(x, y)

    def nrm(x):
        return float(torch.linalg.vector_norm(x))

    # ---- symmetry check of the preconditioner:  y^T M x  vs  x^T M y
    x1, x2 = rhs[0], rhs[1]
    s12, s21 = complex(bdot(x2, Mx(x1))), complex(bdot(x1, Mx(x2)))
    print("M symmetry: |y^T M x - x^T M y| / |.| = %.2e" % (abs(s12 - s21) / abs(s12)), flush=True)

    # ---- GMRES as shipped
    for i, b in enumerate(rhs):
        x = torch.empty_like(b)
        solver.solve(b, x)
        tr = nrm(b - Kx(x)) / nrm(b)
        print("gmres   rhs %d: %3d its  (recycled pairs at entry: %d)  true relres %.2e"
              % (i, solver.last[0], solver.recycled if i else 0, tr), flush=True)

    # ---- deflation space: harmonic Ritz vectors of a 40-step (flexible) Arnoldi run, as the harvest of hx_gmres does
    def arnoldi_harvest(b, m=40, k=6):
        V = torch.empty((m + 1, n), dtype=torch.complex128, device=dev)
        Z = torch.empty((m, n), dtype=torch.complex128, device=dev)
        H = np.zeros((m + 1, m), dtype=np.complex128)
        V[0] = b / nrm(b)
        for j in range(m):
            Z[j] = Mx(V[j])
            w = Kx(Z[j])
            for _ in range(2):
                h = torch.mv(V[:j + 1].conj(), w)
                w = w - torch.mv(V[:j + 1].T, h)
                H[:j + 1, j] += h.cpu().numpy()
            H[j + 1, j] = nrm(w)
            V[j + 1] = w / H[j + 1, j]
        Uc, Cc, theta = harmonic_ritz_recycle(H, k)
        U = (torch.from_numpy(Uc).to(dev).T.contiguous() @ Z)            # k x n : K U = C
        Cv = (torch.from_numpy(Cc).to(dev).T.contiguous() @ V)
        return U, Cv, theta

    def cocg(b, defl=None, maxit=200, name="cocg"):
        """Preconditioned COCG; ``defl`` = (W, KW, Einv): search directions kept K-conjugate (bilinear) to span(W)."""
        x = torch.zeros_like(b)
        r = b.clone()
        if defl is not None:
            W, KW, Einv = defl
            x = W.T @ (Einv @ (W @ r))             # Galerkin start: W^T r = 0 afterwards
            r = b - Kx(x)
        bn = nrm(b)
        hist = []

        def project(z):
            if defl is None:
                return z
            return z - W.T @ (Einv @ (KW @ z))     # (KW)^T-bilinear: rows of KW dotted without conjugation

        z = Mx(r)
        p = project(z)
        rho = bdot(r, z)
        for it in range(1, maxit + 1):
            q = Kx(p)
            alpha = rho / bdot(p, q)
            x = x + alpha * p
            r = r - alpha * q
            rel = nrm(r) / bn
            hist.append(rel)
            if rel <= rtol:
                break
            z = Mx(r)
            rho_new = bdot(r, z)
            beta = rho_new / rho
            rho = rho_new
            p = project(z) + beta * p
        tr = nrm(b - Kx(x)) / bn
        print("%-9s: %3d its  recurrence relres %.2e  TRUE relres %.2e   max/min of history jumps: peak %.2e"
              % (name, it, hist[-1], tr, max(hist)), flush=True)
        return it, hist

    def cocr(b, maxit=200):
        x = torch.zeros_like(b)
        r = b.clone()
        bn = nrm(b)
        rt = Mx(r)
        t = Kx(rt)
        p, q = rt.clone(), t.clone()
        u = Mx(q)
        rho = bdot(rt, t)
        hist = []
        for it in range(1, maxit + 1):
            alpha = rho / bdot(q, u)
            x = x + alpha * p
            r = r - alpha * q
            rt = rt - alpha * u
            rel = nrm(r) / bn
            hist.append(rel)
            if rel <= rtol:
                break
            t = Kx(rt)
            rho_new = bdot(rt, t)
            beta = rho_new / rho
            rho = rho_new
            p = rt + beta * p
            q = t + beta * q
            u = Mx(q)
        tr = nrm(b - Kx(x)) / bn
        print("cocr     : %3d its  recurrence relres %.2e  TRUE relres %.2e  peak %.2e" % (it, hist[-1], tr, max(hist)),
              flush=True)
        return it, hist

    for i, b in enumerate(rhs[:2]):
        print("--- rhs %d" % i)
        _, h1 = cocg(b)
        print("   cocg history: " + " ".join("%.1e" % v for v in h1), flush=True)
        _, h2 = cocr(b)
        print("   cocr history: " + " ".join("%.1e" % v for v in h2), flush=True)

    def cocg_additive(b, W, Einv, maxit=200, name="cocg-add", mult=False, KW=None):
        """COCG with the two-level preconditioner  z = M r + W E^-1 W^T r  (additive coarse correction on span(W); a
        symmetric operator for any W, so rounding W to complex64 cannot destabilise the recurrence).
        mult: z = Q r + M (r - K Q r)  (multiplicative, unsymmetric in general)."""
        x = torch.zeros_like(b)
        r = b.clone()
        bn = nrm(b)

        def prec(r):
            qr = W.T @ (Einv @ (W @ r))
            if mult:
                return qr + Mx(r - KW.T @ (Einv @ (W @ r)))
            return Mx(r) + qr

        z = prec(r)
        p = z.clone()
        rho = bdot(r, z)
        hist = []
        for it in range(1, maxit + 1):
            q = Kx(p)
            alpha = rho / bdot(p, q)
            x = x + alpha * p
            r = r - alpha * q
            rel = nrm(r) / bn
            hist.append(rel)
            if rel <= rtol:
                break
            z = prec(r)
            rho_new = bdot(r, z)
            p = z + (rho_new / rho) * p
            rho = rho_new
        tr = nrm(b - Kx(x)) / bn
        print("%-12s: %3d its  recurrence relres %.2e  TRUE relres %.2e  peak %.2e" % (name, it, hist[-1], tr, max(hist)),
              flush=True)
        return it

    for k in (6, 12, 20):
        U, Cv, theta = arnoldi_harvest(rhs[2], m=40, k=k)
        print("harmonic Ritz |theta| (k=%d): %s" % (k, np.array2string(theta, precision=3)))
        W = U                                         # k x n
        KW = torch.stack([Kx(W[j]) for j in range(k)])
        E = (W @ KW.T)                                # W^T K W (bilinear), k x k
        Einv = torch.linalg.inv(E)
        W32 = W.to(torch.complex64).to(torch.complex128)
        KW32 = KW.to(torch.complex64).to(torch.complex128)
        for i, b in enumerate(rhs[:2]):
            cocg(b, defl=(W, KW, Einv), name="cocg-d%d" % k)
            cocg(b, defl=(W32, KW32, Einv), name="cocg-d%d-c64" % k, maxit=60)
            cocg_additive(b, W, Einv, name="cocg-add%d" % k)
            cocg_additive(b, W32, torch.linalg.inv(W32 @ torch.stack([Kx(W32[j]) for j in range(k)]).T), name="cocg-add%d-c64" % k)
            cocg_additive(b, W, Einv, name="cocg-mult%d" % k, mult=True, KW=KW)


if __name__ == "__main__":
    t0 = time.time()
    main()
    print("done in %.0f s" % (time.time() - t0))
