"""Smoothed-aggregation AMG hierarchy + device GMRES: the iterative "inner solve" of shift-invert Arnoldi.

Role on the path: K^-1 for K = A + sigma B + sigma^2 C when a sparse direct factorisation does not fit (3-D, from
~1e5 DOF up to the 5 M-DOF headline case); the reference delegates this to PETSc PCLU -> MUMPS (eigensolvers.py:
49-50, 102-103).  The thermoacoustic problems here are LOW-frequency Helmholtz problems with a complex shift
(the domain spans 1-2 wavelengths, sigma has a 10-30 % imaginary part), for which a multigrid V-cycle built on the
Laplacian-like part of K is an effective preconditioner and every coarse grid still resolves the wave.

Split of work
  host (once per K, integer / graph work + sparse products): strength-of-connection filter, greedy aggregation
      (hx_host_aggregate, C++), tentative + Jacobi-smoothed prolongator, Galerkin products P^T K P, dense inverse
      of the coarsest operator;
  device (every solve): hx_gmres -> hx_amg_vcycle -> fused Jacobi sweeps / residual / P, R products / CGS2 panels.
"""
import ctypes as C
import os

import numpy as np
import scipy.sparse as sp
import torch

from . import _hx
from .petsc import to_device
from .solver_utils import info

PARAMS = dict(theta=0.04, max_levels=8, coarse_size=int(os.environ.get("HX_AMG_COARSE_SIZE", "1000")), pre=2, post=2, jacobi_omega=None, rtol=1e-11, restart=50,
              max_it=200, precision="mixed", smoother="chebyshev", sell_min_rows=int(os.environ.get("HX_SELL_MIN_ROWS", "150000")), sell_max_row=int(os.environ.get("HX_SELL_MAX_ROW", "40")), vcycle_values="fp16", graph_max_rows=0, recycle=8,
              recycle_min_rows=0, recycle_theta=0.45, krylov="auto", cocg_min_rows=0, setup="device",
              replicate_max_rows=int(os.environ.get("HX_AMG_REPLICATE_MAX_ROWS", "200000")))


MAX_DENSE_COARSE = 6000          # the coarsest operator is inverted densely on the host


def _aggregate(S):
    """Greedy aggregation of the (symmetric) strength graph S (CSR; diagonal entries ignored, |value| = strength)."""
    n = S.shape[0]
    rp = np.ascontiguousarray(S.indptr, dtype=np.int32)
    ci = np.ascontiguousarray(S.indices, dtype=np.int32)
    w = np.ascontiguousarray(S.data, dtype=np.float64)
    agg = np.empty(n, dtype=np.int32)
    na = C.c_int32(0)
    _hx.check(_hx.lib.hx_host_aggregate(n, _hx.nptr(rp), _hx.nptr(ci), _hx.nptr(w), _hx.nptr(agg), C.byref(na)))
    return agg, na.value


def _filter(K, sign, theta):
    """Filtered Laplacian Lf (strong couplings + lumped diagonal) of L = sign * Re(K) as scipy CSR -- one native pass
    (hx_host_sa_filter) that reads the real parts of K's complex values in place."""
    n = K.shape[0]
    rp = np.ascontiguousarray(K.indptr, dtype=np.int32)
    ci = np.ascontiguousarray(K.indices, dtype=np.int32)
    kv = np.ascontiguousarray(K.data, dtype=np.complex128)
    orp = np.empty(n + 1, dtype=np.int32)
    args = (n, _hx.nptr(rp), _hx.nptr(ci), _hx.nptr(kv), 2, float(sign), float(theta))
    _hx.check(_hx.lib.hx_host_sa_filter(*args, _hx.nptr(orp), None, None))
    oci = np.empty(orp[-1], dtype=np.int32)
    ov = np.empty(orp[-1], dtype=np.float64)
    _hx.check(_hx.lib.hx_host_sa_filter(*args, _hx.nptr(orp), _hx.nptr(oci), _hx.nptr(ov)))
    return sp.csr_matrix((ov, oci, orp), shape=K.shape)


def _prolongator(Lf, agg, na, omega):
    """P = (I - omega Df^-1 Lf) P0 as scipy CSR with sorted columns (hx_host_sa_prolongator)."""
    n = Lf.shape[0]
    rp = np.ascontiguousarray(Lf.indptr, dtype=np.int32)
    ci = np.ascontiguousarray(Lf.indices, dtype=np.int32)
    lf = np.ascontiguousarray(Lf.data, dtype=np.float64)
    isc = 1.0 / np.sqrt(np.bincount(agg, minlength=na).astype(np.float64))
    orp = np.empty(n + 1, dtype=np.int32)
    args = (n, _hx.nptr(rp), _hx.nptr(ci), _hx.nptr(lf), _hx.nptr(agg), _hx.nptr(isc), float(omega))
    _hx.check(_hx.lib.hx_host_sa_prolongator(*args, _hx.nptr(orp), None, None))
    oci = np.empty(orp[-1], dtype=np.int32)
    ov = np.empty(orp[-1], dtype=np.float64)
    _hx.check(_hx.lib.hx_host_sa_prolongator(*args, _hx.nptr(orp), _hx.nptr(oci), _hx.nptr(ov)))
    return sp.csr_matrix((ov, oci, orp), shape=(n, na))


def _galerkin(R, K, P, n_threads=None):
    """Kc = R K P (scipy CSR in, scipy CSR out) by the fused multi-threaded native pass hx_host_galerkin_rap."""
    import os
    n_threads = n_threads or min(16, os.cpu_count() or 1)
    nc = R.shape[0]
    arrs = [np.ascontiguousarray(R.indptr, np.int32), np.ascontiguousarray(R.indices, np.int32),
            np.ascontiguousarray(R.data, np.float64),
            np.ascontiguousarray(K.indptr, np.int32), np.ascontiguousarray(K.indices, np.int32),
            np.ascontiguousarray(K.data, np.complex128),
            np.ascontiguousarray(P.indptr, np.int32), np.ascontiguousarray(P.indices, np.int32),
            np.ascontiguousarray(P.data, np.float64)]
    ptrs = [_hx.nptr(a) for a in arrs]
    orp = np.empty(nc + 1, dtype=np.int32)
    _hx.check(_hx.lib.hx_host_galerkin_rap(nc, *ptrs, _hx.nptr(orp), None, None, n_threads))
    oci = np.empty(orp[-1], dtype=np.int32)
    ov = np.empty(orp[-1], dtype=np.complex128)
    _hx.check(_hx.lib.hx_host_galerkin_rap(nc, *ptrs, _hx.nptr(orp), _hx.nptr(oci), _hx.nptr(ov), n_threads))
    return sp.csr_matrix((ov, oci, orp), shape=(nc, P.shape[1]))


def _diagonal(K):
    """Diagonal of a scipy CSR (complex128) by the native row-parallel pass."""
    n = K.shape[0]
    rp = np.ascontiguousarray(K.indptr, dtype=np.int32)
    ci = np.ascontiguousarray(K.indices, dtype=np.int32)
    kv = np.ascontiguousarray(K.data, dtype=np.complex128)
    out = np.empty(n, dtype=np.complex128)
    _hx.check(_hx.lib.hx_host_csr_diagonal(n, _hx.nptr(rp), _hx.nptr(ci), _hx.nptr(kv), _hx.nptr(out)))
    return out


def _spectral_radius_dinv(Lf, iters=None, seed=0):
    """Power-iteration estimate of rho(Df^-1 Lf) (similar to an SPD matrix: the norm ratio converges from below; the
    smoothing weights derived from it carry a 4/3 safety factor).  Native row-parallel pass (hx_host_power_rho)."""
    n = Lf.shape[0]
    if iters is None:
        iters = int(os.environ.get("HX_AMG_RHO_ITERS", "10"))
    v = np.random.default_rng(seed).standard_normal(n)
    rp = np.ascontiguousarray(Lf.indptr, dtype=np.int32)
    ci = np.ascontiguousarray(Lf.indices, dtype=np.int32)
    lf = np.ascontiguousarray(Lf.data, dtype=np.float64)
    rho = C.c_double(0.0)
    _hx.check(_hx.lib.hx_host_power_rho(n, _hx.nptr(rp), _hx.nptr(ci), _hx.nptr(lf), _hx.nptr(v), int(iters),
                                        C.byref(rho)))
    return rho.value


def _is_complex_symmetric(K, tol=1e-11):
    """x^T K y == y^T K x for two random device vectors (bilinear form, no conjugation): K^T = K up to rounding."""
    n = K.shape[0]
    gen = torch.Generator(device=K.val.device)
    gen.manual_seed(20260101)
    x, y = (torch.view_as_complex(torch.randn((n, 2), generator=gen, device=K.val.device, dtype=torch.float64))
            for _ in range(2))
    kx, ky = torch.empty_like(x), torch.empty_like(x)
    K.spmv(x, kx)
    K.spmv(y, ky)
    a, b = complex(torch.dot(y, kx)), complex(torch.dot(x, ky))
    scale = float(torch.linalg.vector_norm(kx) * torch.linalg.vector_norm(y))
    return abs(a - b) <= tol * max(scale, 1e-300)


def build_hierarchy(K, theta=PARAMS["theta"], max_levels=PARAMS["max_levels"], coarse_size=PARAMS["coarse_size"]):
    """K: scipy CSR (complex).  Returns a list of dicts {K, n, dinv, P, R, omega, n_coarse} (host scipy objects),
    finest first.  Graph passes (filter, aggregation, prolongator) are native host code in libhx; the Galerkin
    triple products are SciPy SpGEMMs."""
    levels = []
    Kl = sp.csr_matrix(K, dtype=np.complex128)
    th = theta
    while True:
        n = Kl.shape[0]
        diag = _diagonal(Kl)
        level = dict(K=Kl, n=n)
        if n <= coarse_size or len(levels) == max_levels - 1:
            if n > MAX_DENSE_COARSE:
                raise ArithmeticError("AMG hierarchy still has %d unknowns after %d levels" % (n, len(levels) + 1))
            levels.append(level)
            break
        # Laplacian-like real part with positive diagonal drives strength / prolongator smoothing
        sgn = -1.0 if 2 * np.count_nonzero(diag.real < 0) > n else 1.0
        Lf = _filter(Kl, sgn, th)
        agg, na = _aggregate(Lf)
        for _ in range(3):                         # too few couplings count as strong: relax the threshold and retry
            if na < 0.8 * n:
                break
            th *= 0.25
            Lf = _filter(Kl, sgn, th)
            agg, na = _aggregate(Lf)
        if na >= 0.8 * n:                          # coarsening stalled for good: this level becomes the coarsest one
            if n > MAX_DENSE_COARSE:
                raise ArithmeticError("AMG coarsening stalled at %d unknowns (strength threshold %.3g): the coarsest "
                                      "level would need a dense inverse of that size" % (n, th))
            levels.append(level)
            break
        rho = _spectral_radius_dinv(Lf)
        P = _prolongator(Lf, agg, na, 4.0 / (3.0 * rho))
        R = P.T.tocsr()
        R.sort_indices()
        # smoother: damped Jacobi on K itself; rho(D^-1 L) differs from rho(Df^-1 Lf) only by the lumped weak
        # couplings (a few per cent), so the estimate is reused with a margin instead of a second power iteration
        level.update(P=P, R=R, dinv=1.0 / diag, omega=4.0 / (3.0 * 1.05 * rho), rho=1.05 * rho, n_coarse=na)
        levels.append(level)
        Kl = _galerkin(R, Kl, P)
        th *= 0.5
    return levels


class DevCSR:
    """Device CSR of the hierarchy: int32 rowptr / col tensors + values (complex128 for operators, float64 for the
    transfer operators)."""

    def __init__(self, shape, rowptr, col, val):
        self.shape = (int(shape[0]), int(shape[1]))
        self.rowptr, self.col, self.val = rowptr, col, val
        self.nnz = int(col.numel())

    def to_scipy(self):
        return sp.csr_matrix((self.val.cpu().numpy(), self.col.cpu().numpy(), self.rowptr.cpu().numpy()), shape=self.shape)

    def rows(self, r0, r1):
        """Rows [r0, r1) as a new device CSR (row-sharded level 0)."""
        a, b = int(self.rowptr[r0]), int(self.rowptr[r1])
        return DevCSR((r1 - r0, self.shape[1]), (self.rowptr[r0:r1 + 1] - a).contiguous(), self.col[a:b].contiguous(),
                      self.val[a:b].contiguous())


def _dev_filter(K, sign, theta, sd):
    """Lf on the device (hx_sa_filter_*): DevCSR with float64 values."""
    n, dev = K.shape[0], K.val.device
    rp = torch.empty(n + 1, dtype=torch.int32, device=dev)
    nnz = C.c_int64(0)
    args = (n, _hx.ptr(K.rowptr), _hx.ptr(K.col), _hx.ptr(K.val), float(sign), float(theta), _hx.ptr(sd))
    _hx.check(_hx.lib.hx_sa_filter_count(*args, _hx.ptr(rp), C.byref(nnz), _hx.stream()))
    ci = torch.empty(nnz.value, dtype=torch.int32, device=dev)
    v = torch.empty(nnz.value, dtype=torch.float64, device=dev)
    _hx.check(_hx.lib.hx_sa_filter_fill(*args, _hx.ptr(rp), _hx.ptr(ci), _hx.ptr(v), _hx.stream()))
    return DevCSR(K.shape, rp, ci, v)


_PINNED = {}
AGG_TIMES = []        # (rows, graph entries, s device -> pinned host, s host pass) of the last aggregation passes (diagnostics)


def _to_host_pinned(t, key):
    """Device tensor -> numpy view of a page-locked staging buffer kept per ``key`` (grown on demand): the strength graph
    of a 5 M-row operator is 0.35 GB, which a pageable copy moves at a third of the PCIe rate."""
    buf = _PINNED.get(key)
    if buf is None or buf.numel() < t.numel() or buf.dtype != t.dtype:
        buf = torch.empty(max(t.numel(), 1), dtype=t.dtype, pin_memory=True)
        _PINNED[key] = buf
    view = buf[:t.numel()]
    view.copy_(t, non_blocking=True)
    return view


def _dev_aggregate(Lf):
    """Greedy aggregation of the strength graph: the one sequential pass of the set-up, run on the host on the
    (pattern, |weight| as float32) of Lf -- the filtered graph is a fraction of K and K itself stays in HBM."""
    import time
    n = Lf.shape[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rp = _to_host_pinned(Lf.rowptr, "agg_rp")
    ci = _to_host_pinned(Lf.col, "agg_ci")
    w = _to_host_pinned(Lf.val.to(torch.float32), "agg_w")
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    agg = np.empty(n, dtype=np.int32)
    na = C.c_int32(0)
    _hx.check(_hx.lib.hx_host_aggregate_f32(n, C.c_void_p(rp.data_ptr()), C.c_void_p(ci.data_ptr()),
                                            C.c_void_p(w.data_ptr()), _hx.nptr(agg), C.byref(na)))
    AGG_TIMES.append((n, int(Lf.nnz), round(t1 - t0, 4), round(time.perf_counter() - t1, 4)))
    del AGG_TIMES[:-8]
    return agg, na.value


def _dev_transpose(A, n_cols):
    dev = A.val.device
    rp = torch.empty(n_cols + 1, dtype=torch.int32, device=dev)
    ci = torch.empty(A.nnz, dtype=torch.int32, device=dev)
    v = torch.empty(A.nnz, dtype=torch.float64, device=dev)
    _hx.check(_hx.lib.hx_csr_transpose_d(A.shape[0], n_cols, A.nnz, _hx.ptr(A.rowptr), _hx.ptr(A.col), _hx.ptr(A.val),
                                         _hx.ptr(rp), _hx.ptr(ci), _hx.ptr(v), _hx.stream()))
    return DevCSR((n_cols, A.shape[0]), rp, ci, v)


def _dev_spgemm(A, B):
    """C = A B on the device (hx_spgemm_*); operands real (float64) or complex128, result complex128, sorted columns."""
    dev = A.val.device
    rp = torch.empty(A.shape[0] + 1, dtype=torch.int32, device=dev)
    nnz, plan = C.c_int64(0), C.c_void_p()
    _hx.check(_hx.lib.hx_spgemm_symbolic(C.byref(plan), A.shape[0], _hx.ptr(A.rowptr), _hx.ptr(A.col), B.shape[1],
                                         _hx.ptr(B.rowptr), _hx.ptr(B.col), _hx.ptr(rp), C.byref(nnz), _hx.stream()))
    try:
        ci = torch.empty(nnz.value, dtype=torch.int32, device=dev)
        v = torch.empty(nnz.value, dtype=torch.complex128, device=dev)
        _hx.check(_hx.lib.hx_spgemm_numeric(plan, _hx.ptr(A.rowptr), _hx.ptr(A.col), _hx.ptr(A.val),
                                            int(A.val.is_complex()), _hx.ptr(B.rowptr), _hx.ptr(B.col), _hx.ptr(B.val),
                                            int(B.val.is_complex()), _hx.ptr(ci), _hx.ptr(v), _hx.stream()))
    finally:
        _hx.lib.hx_spgemm_destroy(plan)
    return DevCSR((A.shape[0], B.shape[1]), rp, ci, v)


def build_hierarchy_device(K, theta=PARAMS["theta"], max_levels=PARAMS["max_levels"], coarse_size=PARAMS["coarse_size"],
                           level_offset=0, graph=None):
    """Same smoothed-aggregation hierarchy as ``build_hierarchy``, built in HBM: K (petsc.CSR / DevCSR on the device) is
    never copied to the host.  Device passes (amg_setup.hip): diagonal, strength filter, power iteration, smoothed
    prolongator, transpose, Galerkin product as R (K P).  Host: the greedy aggregation of the filtered graph and the
    dense inverse of the coarsest operator (a few hundred unknowns).  Returns level dicts whose K / P / R are DevCSR.

    ``graph`` (CSR on K's pattern, e.g. the stiffness part A of K = A + sigma B + sigma^2 C): strength of connection, aggregation
    and prolongator smoothing are taken from IT on every level (G_{l+1} = R G_l P is carried down beside K_{l+1} = R K_l P) -- the
    transfer operators are then those of the Laplacian whatever the shift does to the signs and sizes of K's entries (at high
    frequency sigma^2 C outweighs A on the coarse levels and the "Laplacian-like part of K" is no Laplacian any more)."""
    levels = []
    Kl = DevCSR(K.shape, K.rowptr, K.col, K.val)
    Gl = None if graph is None else DevCSR(graph.shape, graph.rowptr, graph.col, graph.val)
    dev = K.val.device
    # (level_offset > 0: continuation of a row-sharded set-up below its last sharded level -- theta was halved per level)
    th = float(os.environ.get("HX_AMG_THETA", theta)) if level_offset == 0 else float(theta)
    p_damp = float(os.environ.get("HX_AMG_PDAMP", 4.0 / 3.0))
    rho_iters = int(os.environ.get("HX_AMG_RHO_ITERS", "10"))
    max_levels = max_levels - level_offset
    while True:
        n = Kl.shape[0]
        diag = torch.empty(n, dtype=torch.complex128, device=dev)
        _hx.check(_hx.lib.hx_csr_diagonal_z(n, _hx.ptr(Kl.rowptr), _hx.ptr(Kl.col), _hx.ptr(Kl.val), _hx.ptr(diag),
                                            _hx.stream()))
        level = dict(K=Kl, n=n)
        if n <= coarse_size or len(levels) == max_levels - 1:
            if n > MAX_DENSE_COARSE:
                raise ArithmeticError("AMG hierarchy still has %d unknowns after %d levels" % (n, len(levels) + 1))
            levels.append(level)
            break
        Sl = Kl if Gl is None else Gl              # the operator whose real part plays the Laplacian
        if Gl is None:
            sgn = -1.0 if 2 * int((diag.real < 0).sum()) > n else 1.0
        else:
            gd = torch.empty(n, dtype=torch.complex128, device=dev)
            _hx.check(_hx.lib.hx_csr_diagonal_z(n, _hx.ptr(Gl.rowptr), _hx.ptr(Gl.col), _hx.ptr(Gl.val), _hx.ptr(gd), _hx.stream()))
            sgn = -1.0 if 2 * int((gd.real < 0).sum()) > n else 1.0
            del gd
        sd = torch.empty(n, dtype=torch.float64, device=dev)
        Lf = _dev_filter(Sl, sgn, th, sd)
        agg, na = _dev_aggregate(Lf)
        for _ in range(3):                         # too few couplings count as strong: relax the threshold and retry
            if na < 0.8 * n:
                break
            th *= 0.25
            Lf = _dev_filter(Sl, sgn, th, sd)
            agg, na = _dev_aggregate(Lf)
        if na >= 0.8 * n:                          # coarsening stalled for good: this level becomes the coarsest one
            if n > MAX_DENSE_COARSE:
                raise ArithmeticError("AMG coarsening stalled at %d unknowns (strength threshold %.3g): the coarsest "
                                      "level would need a dense inverse of that size" % (n, th))
            levels.append(level)
            break
        gen = torch.Generator(device=dev)
        gen.manual_seed(len(levels) + level_offset)
        v = torch.randn(n, generator=gen, device=dev, dtype=torch.float64)
        w = torch.empty_like(v)
        rho = C.c_double(0.0)
        _hx.check(_hx.lib.hx_sa_power_rho(n, _hx.ptr(Lf.rowptr), _hx.ptr(Lf.col), _hx.ptr(Lf.val), _hx.ptr(v), _hx.ptr(w),
                                          rho_iters, C.byref(rho), _hx.stream()))
        rho = rho.value
        agg_d = to_device(agg)
        isc = to_device(1.0 / np.sqrt(np.bincount(agg, minlength=na).astype(np.float64)))
        prp = torch.empty(n + 1, dtype=torch.int32, device=dev)
        pnnz = C.c_int64(0)
        _hx.check(_hx.lib.hx_sa_prolongator_count(n, _hx.ptr(Lf.rowptr), _hx.ptr(Lf.col), _hx.ptr(agg_d), _hx.ptr(prp),
                                                  C.byref(pnnz), _hx.stream()))
        pci = torch.empty(pnnz.value, dtype=torch.int32, device=dev)
        pv = torch.empty(pnnz.value, dtype=torch.float64, device=dev)
        _hx.check(_hx.lib.hx_sa_prolongator_fill(n, _hx.ptr(Lf.rowptr), _hx.ptr(Lf.col), _hx.ptr(Lf.val), _hx.ptr(agg_d),
                                                 _hx.ptr(isc), p_damp / rho, _hx.ptr(prp), _hx.ptr(pci), _hx.ptr(pv),
                                                 _hx.stream()))
        P = DevCSR((n, na), prp, pci, pv)
        R = _dev_transpose(P, na)
        level.update(P=P, R=R, dinv=1.0 / diag, omega=4.0 / (3.0 * 1.05 * rho), rho=1.05 * rho, n_coarse=na)
        levels.append(level)
        del Lf, sd, v, w
        Kl = _dev_spgemm(R, _dev_spgemm(Kl, P))
        if Gl is not None:
            Gl = _dev_spgemm(R, _dev_spgemm(Gl, P))
        th *= 0.5
    return levels



# ----------------------------------------------------------------------------- row-sharded set-up
def _host_sum(shard, *vals):
    """Sum of host integers over the ranks (set-up decisions must be identical everywhere)."""
    t = torch.tensor([float(v) for v in vals], dtype=torch.float64, device=_hx.require_gpu())
    shard.all_reduce_(t)
    return [int(round(x)) for x in t.cpu().tolist()]


def _all_gather_var(shard, t):
    """Concatenation over the ranks (rank order) of 1-D device tensors of different lengths: set-up only."""
    import torch.distributed as dist
    sizes = [None] * shard.world
    dist.all_gather_object(sizes, int(t.numel()), group=shard.group)
    cap = max(max(sizes), 1)
    v = torch.view_as_real(t) if t.is_complex() else t
    pad = torch.zeros((cap,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
    pad[:t.numel()] = v
    stage = shard.backend == "gloo" and pad.is_cuda
    src = pad.cpu() if stage else pad
    outs = [torch.empty_like(src) for _ in range(shard.world)]
    dist.all_gather(outs, src, group=shard.group)
    cat = torch.cat([o[:m] for o, m in zip(outs, sizes)]).to(t.device)
    return torch.view_as_complex(cat.contiguous()) if t.is_complex() else cat


def _is_complex_symmetric_sharded(K, shard, tol=1e-11):
    """x^T K y == y^T K x on the row-sharded operator (K: owned rows over [owned | ghost]); collective."""
    n_own = K.shape[0]
    gen = torch.Generator(device=K.val.device)
    gen.manual_seed(20260101)
    xg, yg = (torch.view_as_complex(torch.randn((shard.n_global, 2), generator=gen, device=K.val.device, dtype=torch.float64))
              for _ in range(2))
    xe, ye = shard.ext_vector(), shard.ext_vector()
    xe[:n_own], ye[:n_own] = xg[shard.r0:shard.r1], yg[shard.r0:shard.r1]
    shard.halo(xe)
    shard.halo(ye)
    kx, ky = torch.empty(n_own, dtype=torch.complex128, device=K.val.device), torch.empty(n_own, dtype=torch.complex128, device=K.val.device)
    K.spmv(xe, kx)
    K.spmv(ye, ky)
    t = torch.stack([torch.dot(ye[:n_own], kx), torch.dot(xe[:n_own], ky),
                     (torch.linalg.vector_norm(kx) ** 2).to(torch.complex128),
                     (torch.linalg.vector_norm(ye[:n_own]) ** 2).to(torch.complex128)])
    shard.all_reduce_(t)
    a, b, nk, ny = (complex(z) for z in t.cpu().tolist())
    return abs(a - b) <= tol * max(float(np.sqrt(nk.real * ny.real)), 1e-300)


def _rows_concat(a, b):
    """CSR rows of ``a`` followed by the rows of ``b`` (same column space)."""
    rp = torch.cat([a.rowptr, (b.rowptr[1:].to(torch.int64) + a.nnz).to(torch.int32)])
    return DevCSR((a.shape[0] + b.shape[0], a.shape[1]), rp, torch.cat([a.col, b.col]), torch.cat([a.val, b.val]))


def _select_columns(A, c0, c1):
    """Entries of A with column in [c0, c1), columns shifted to start at 0 (device integer work)."""
    keep = (A.col >= c0) & (A.col < c1)
    n = A.shape[0]
    lens = A.rowptr[1:] - A.rowptr[:-1]
    row_of = torch.repeat_interleave(torch.arange(n, device=A.col.device, dtype=torch.int64), lens.to(torch.int64))
    cnt = torch.zeros(n, dtype=torch.int64, device=A.col.device)
    cnt.index_add_(0, row_of[keep], torch.ones(int(keep.sum()), dtype=torch.int64, device=A.col.device))
    rp = torch.zeros(n + 1, dtype=torch.int32, device=A.col.device)
    rp[1:] = torch.cumsum(cnt, 0).to(torch.int32)
    return DevCSR((n, c1 - c0), rp, (A.col[keep] - c0).contiguous(), A.val[keep].contiguous())


def _localize_columns(A, c0, c1, ghost_sorted, n_own):
    """Global column ids -> [owned | ghost] numbering: owned [c0, c1) -> col - c0, others -> n_own + position in the
    ascending ghost list (device int64 tensor).  Returns (operator, number of columns found OUTSIDE owned + ghost set): the
    caller sums that count over the ranks before it raises, so that all of them raise together (a rank that raised alone
    would leave the others blocked in their next exchange)."""
    col = A.col.long()
    own = (col >= c0) & (col < c1)
    if ghost_sorted.numel():
        pos = torch.searchsorted(ghost_sorted, col).clamp(max=int(ghost_sorted.numel()) - 1)
        n_bad = int((~(own | (ghost_sorted[pos] == col))).sum())
    else:
        pos = torch.zeros_like(col)
        n_bad = int((~own).sum())
    lcol = torch.where(own, col - c0, n_own + pos).to(torch.int32).contiguous()
    return DevCSR((A.shape[0], n_own + int(ghost_sorted.numel())), A.rowptr, lcol, A.val), n_bad


# row-sharded levels: prolongators with the ghost vertices' rows (one halo fewer per level and cycle); HX_AMG_GHOST_P=0: owned rows only
PARAMS_GHOST_P = os.environ.get("HX_AMG_GHOST_P", "1") != "0"
# default of HX_AMG_DEEP_HALO: two ghost layers per sharded level, 6 instead of 9 halos per inner iteration at two sharded levels
# (measured on the one-GPU rehearsals: same iteration counts and eigenvalues, profiles/r05_deep_halo_rehearsals.txt); "0": classic
PARAMS_DEEP_HALO = "1"


def build_hierarchy_sharded(K, shard, theta=PARAMS["theta"], max_levels=PARAMS["max_levels"],
                            coarse_size=PARAMS["coarse_size"], replicate_max_rows=PARAMS["replicate_max_rows"]):
    """Smoothed-aggregation hierarchy of the ROW-SHARDED operator: K holds this rank's rows over [owned | ghost] columns
    (level-0 plan of ``shard``).  What PCSetUp does for the reference's distributed matrix (MUMPS / SuperLU_DIST analysis +
    factorisation on N ranks, eigensolvers.py:49-50,102-103) -- here every rank builds its SLAB of every large level:

      * strength filter row-local (the ghost entries of sqrt|l_jj| exchanged once);
      * UNCOUPLED aggregation: the greedy pass runs on the owned sub-graph, aggregates never cross a rank boundary; coarse
        ids are numbered rank after rank, so every level is again a contiguous row partition;
      * rho(Df^-1 Lf) by a distributed power iteration (halo of the iterate per step, norms summed once);
      * smoothed prolongator rows P[own, :] (aggregate ids and sizes of ghost vertices exchanged);
      * Galerkin operator: K P from owned rows after a sparse-row halo of P, then R (K P) with R = the transpose of the
        [owned | ghost] rows of P restricted to MY aggregates, after a sparse-row halo of K P: my rows of the coarse operator,
        no reduction of partial sums and bitwise independent of timing;
      * levels with more than ``replicate_max_rows`` global rows stay sharded (their own exchange plan); the first level
        below that is all-gathered (rank blocks concatenate) and the remaining levels are built replicated from it by
        ``build_hierarchy_device``.

    Returns the level dicts: sharded ones carry K / P / R / dinv in local numbering plus ``sharded=True, n (owned rows),
    n_ext, n_global``; then the replicated tail as ``build_hierarchy_device`` returns it."""
    dev = K.val.device
    levels = []
    th = float(os.environ.get("HX_AMG_THETA", theta))
    p_damp = float(os.environ.get("HX_AMG_PDAMP", 4.0 / 3.0))
    rho_iters = int(os.environ.get("HX_AMG_RHO_ITERS", "10"))
    Kl = DevCSR(K.shape, K.rowptr, K.col, K.val)
    l = 0
    ranges = list(shard.level_ranges[0])
    n_glob = ranges[-1][1]
    # deep-halo mode (round 5): operator rows of the first ghost layer + a second layer of ghost entries per sharded level, so that
    # a cycle needs 3 exchanges on a level instead of 5 (vcycle_deep, csrc/amg.hip).  Needs the ghost-row prolongators.
    deep = PARAMS_GHOST_P and os.environ.get("HX_AMG_DEEP_HALO", PARAMS_DEEP_HALO) != "0"
    ghost_l = shard.ghost.long() if shard.ghost is not None else torch.zeros(0, dtype=torch.int64, device=dev)
    while True:
        n_own, n_ext = Kl.shape
        assert (n_own, n_ext) == shard.level_n[l], "operator does not match the level's exchange plan"
        diag = torch.empty(n_own, dtype=torch.complex128, device=dev)
        _hx.check(_hx.lib.hx_csr_diagonal_z(n_own, _hx.ptr(Kl.rowptr), _hx.ptr(Kl.col), _hx.ptr(Kl.val), _hx.ptr(diag),
                                            _hx.stream()))
        level = dict(K=Kl, n=n_own, n_ext=n_ext, n_global=n_glob, sharded=True)
        (n_neg,) = _host_sum(shard, int((diag.real < 0).sum()))
        sgn = -1.0 if 2 * n_neg > n_glob else 1.0
        # strength filter with exchanged sqrt|l_jj| of the ghost columns
        sd = torch.zeros(n_ext, dtype=torch.float64, device=dev)

        def filtered(theta_l):
            _hx.check(_hx.lib.hx_sa_strength_diag(n_own, _hx.ptr(Kl.rowptr), _hx.ptr(Kl.col), _hx.ptr(Kl.val), sgn,
                                                  _hx.ptr(sd), _hx.stream()))
            shard.halo(sd, l)
            return _dev_filter(Kl, sgn, theta_l, sd)

        Lf = filtered(th)
        agg, na = _dev_aggregate(Lf)
        for _ in range(3):
            (na_glob,) = _host_sum(shard, na)
            if na_glob < 0.8 * n_glob:
                break
            th *= 0.25
            Lf = filtered(th)
            agg, na = _dev_aggregate(Lf)
        # the decision is COLLECTIVE: a rank that raised alone would leave the others blocked in the next exchange
        na_glob, n_empty = _host_sum(shard, na, int(na == 0))
        if na_glob >= 0.8 * n_glob or n_empty > 0:
            raise ArithmeticError("row-sharded AMG: coarsening stalled at %d unknowns (strength threshold %.3g; %d rank(s) "
                                  "without aggregates)" % (n_glob, th, int(n_empty)))
        # coarse numbering: rank after rank
        counts = [None] * shard.world
        import torch.distributed as dist
        dist.all_gather_object(counts, int(na), group=shard.group)
        offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        c_ranges = [(int(offs[p]), int(offs[p + 1])) for p in range(shard.world)]
        c0, c1 = c_ranges[shard.rank]
        # aggregate id (global) and 1/sqrt(size) per vertex, ghosts by halo
        agg_ext = torch.zeros(n_ext, dtype=torch.int32, device=dev)
        agg_ext[:n_own] = to_device(agg) + int(c0)
        shard.halo(agg_ext, l)
        isc_own = 1.0 / np.sqrt(np.bincount(agg, minlength=na).astype(np.float64))
        iscv = torch.zeros(n_ext, dtype=torch.float64, device=dev)
        iscv[:n_own] = to_device(isc_own[agg])
        shard.halo(iscv, l)
        g_agg = agg_ext[n_own:].long()
        ghost_aggs = torch.unique(g_agg)                                  # ascending global ids, all off-rank
        na_ext = na + int(ghost_aggs.numel())
        agg_loc = torch.empty(n_ext, dtype=torch.int32, device=dev)
        agg_loc[:n_own] = agg_ext[:n_own] - int(c0)
        isc = torch.zeros(na_ext, dtype=torch.float64, device=dev)
        isc[:na] = to_device(isc_own)
        if ghost_aggs.numel():
            gpos = torch.searchsorted(ghost_aggs, g_agg)
            agg_loc[n_own:] = (na + gpos).to(torch.int32)
            isc[na + gpos] = iscv[n_own:]
        # rho(Df^-1 Lf): distributed power iteration
        gen = torch.Generator(device=dev)
        gen.manual_seed(l)
        vg = torch.randn(n_glob, generator=gen, device=dev, dtype=torch.float64)     # same global start vector on every rank
        r0_l = ranges[shard.rank][0]
        v = torch.zeros(n_ext, dtype=torch.float64, device=dev)
        v[:n_own] = vg[r0_l:r0_l + n_own]
        del vg
        w = torch.zeros(n_ext, dtype=torch.float64, device=dev)
        norms = torch.zeros(rho_iters, dtype=torch.float64, device=dev)
        scratch = torch.empty(1024, dtype=torch.float64, device=dev)
        for it in range(rho_iters):
            shard.halo(v, l)
            _hx.check(_hx.lib.hx_sa_power_step(n_own, _hx.ptr(Lf.rowptr), _hx.ptr(Lf.col), _hx.ptr(Lf.val), _hx.ptr(v),
                                               _hx.ptr(w), C.c_void_p(norms.data_ptr() + 8 * it), _hx.ptr(scratch),
                                               _hx.stream()))
            v, w = w, v
        shard.all_reduce_(norms)
        nh = norms.cpu().numpy()
        rho = float(np.sqrt(nh[-1] / nh[-2]))
        # smoothed prolongator rows of the owned vertices over [own | ghost] aggregates
        prp = torch.empty(n_own + 1, dtype=torch.int32, device=dev)
        pnnz = C.c_int64(0)
        _hx.check(_hx.lib.hx_sa_prolongator_count(n_own, _hx.ptr(Lf.rowptr), _hx.ptr(Lf.col), _hx.ptr(agg_loc), _hx.ptr(prp),
                                                  C.byref(pnnz), _hx.stream()))
        pci = torch.empty(pnnz.value, dtype=torch.int32, device=dev)
        pv = torch.empty(pnnz.value, dtype=torch.float64, device=dev)
        _hx.check(_hx.lib.hx_sa_prolongator_fill(n_own, _hx.ptr(Lf.rowptr), _hx.ptr(Lf.col), _hx.ptr(Lf.val), _hx.ptr(agg_loc),
                                                 _hx.ptr(isc), p_damp / rho, _hx.ptr(prp), _hx.ptr(pci), _hx.ptr(pv),
                                                 _hx.stream()))
        # columns -> GLOBAL coarse ids for the products (ascending local ids = own aggregates, then ghosts ascending)
        loc2glob = torch.cat([torch.arange(c0, c1, device=dev, dtype=torch.int64), ghost_aggs])
        P_glob = DevCSR((n_own, na_glob), prp, loc2glob[pci.long()].to(torch.int32).contiguous(), pv)
        plan = shard.plans[l]
        P_ext = _rows_concat(P_glob, DevCSR((n_ext - n_own, na_glob), *plan.exchange_rows(prp, P_glob.col, pv)))
        KP = _dev_spgemm(Kl, P_ext)                                        # owned rows, global coarse columns
        KP_ext = _rows_concat(KP, DevCSR((n_ext - n_own, na_glob), *plan.exchange_rows(KP.rowptr, KP.col, KP.val)))
        R_loc = _dev_transpose(_select_columns(P_ext, c0, c1), na)         # my aggregates x [own | ghost] fine vertices
        Kc = _dev_spgemm(R_loc, KP_ext)                                    # my coarse rows, global coarse columns
        # The level's prolongator keeps the owners' rows of my ghost vertices behind my own (hx_amg_set_prolongator_ghost_rows):
        # the prolongation then corrects the ghost block of the iterate too and the first post-smoothing step needs no exchange
        P_lvl = P_ext if PARAMS_GHOST_P else P_glob
        dinv_l = 1.0 / diag
        if deep:
            # ---- ghost layers beyond the first.  Layer 1 = the level's ghost set; its OPERATOR ROWS come from their owners
            # (sparse-row halo, columns as global ids); layer 2 = their columns that are neither owned nor in layer 1; with three
            # layers (default) the rows of layer 2 are fetched as well and layer 3 = their new columns.  Vectors: [owned | g1 | g2 | g3].
            n_layers = int(os.environ.get("HX_AMG_DEEP_LAYERS", "3"))
            r0_l = ranges[shard.rank][0]
            n1 = n_ext - n_own
            loc2glob_f = torch.cat([torch.arange(r0_l, r0_l + n_own, device=dev, dtype=torch.int64), ghost_l])
            Kcol_g = loc2glob_f[Kl.col.long()].to(torch.int32).contiguous()
            known = [ghost_l]                                                  # ascending id lists of the layers so far
            rows = [DevCSR((n1, n_glob), *plan.exchange_rows(Kl.rowptr, Kcol_g, Kl.val))]
            extra_plans = []

            def new_columns(M):
                kcol = M.col.long()
                seen = (kcol >= r0_l) & (kcol < r0_l + n_own)
                for ids in known:
                    if ids.numel():
                        pos = torch.searchsorted(ids, kcol).clamp(max=int(ids.numel()) - 1)
                        seen = seen | (ids[pos] == kcol)
                return torch.unique(kcol[~seen])

            for layer in range(2, n_layers + 1):
                g = new_columns(rows[-1])                                       # ascending global ids of this layer
                pl = shard.layer_plan(l, g.cpu().numpy())
                extra_plans.append(pl)
                known.append(g)
                if layer < n_layers:                                            # its operator rows feed the next layer
                    rows.append(DevCSR((int(g.numel()), n_glob), *pl.exchange_rows(Kl.rowptr, Kcol_g, Kl.val)))
            dplan = shard.set_level_deep_plan(l, extra_plans)
            sizes = [int(ids.numel()) for ids in known]
            n_ext2 = n_own + sum(sizes)

            def localized(M):
                kcol = M.col.long()
                lcol = torch.where((kcol >= r0_l) & (kcol < r0_l + n_own), kcol - r0_l, torch.full_like(kcol, -1))
                base = n_own
                for ids in known:
                    if ids.numel():
                        pos = torch.searchsorted(ids, kcol).clamp(max=int(ids.numel()) - 1)
                        lcol = torch.where((lcol < 0) & (ids[pos] == kcol), base + pos, lcol)
                    base += int(ids.numel())
                return lcol

            Kg = rows[0]
            for M in rows[1:]:
                Kg = _rows_concat(Kg, M)
            lcol = localized(Kg)
            (n_bad_deep,) = _host_sum(shard, int((lcol < 0).sum()))
            if n_bad_deep:
                raise ValueError("row-sharded AMG: %d column(s) of the ghost-layer rows of level %d are in no layer" % (n_bad_deep, l))
            level["K1"] = DevCSR((Kg.shape[0], n_ext2), Kg.rowptr, lcol.to(torch.int32).contiguous(), Kg.val)
            level["n_g1"], level["n_gs"], level["n_ext2"], level["deep_layers"] = n1, int(Kg.shape[0]), n_ext2, n_layers
            level["nnz_g1"] = int(Kg.rowptr[n1].item())
            dinv_ext = torch.zeros(n_ext2, dtype=torch.complex128, device=dev)
            dinv_ext[:n_own] = dinv_l
            shard.halo(dinv_ext, shard.DEEP + l)
            dinv_l = dinv_ext
            # prolongator rows of layer 2 (behind those of layer 1): the correction then reaches x on owned + g1 + g2 (layer 3, if
            # present, only ever holds right-hand-side entries and the first smoothing step: empty prolongator rows)
            n2 = sizes[1]
            P_lvl = _rows_concat(P_ext, DevCSR((n2, na_glob), *extra_plans[0].exchange_rows(prp, P_glob.col, pv)))
            if n_ext2 > n_own + n1 + n2:
                n3 = n_ext2 - (n_own + n1 + n2)
                P_lvl = _rows_concat(P_lvl, DevCSR((n3, na_glob), torch.zeros(n3 + 1, dtype=torch.int32, device=dev),
                                                   torch.zeros(0, dtype=torch.int32, device=dev),
                                                   torch.zeros(0, dtype=P_ext.val.dtype, device=dev)))
        del KP, KP_ext, P_ext, Lf, sd, v, w
        level.update(dinv=dinv_l, omega=4.0 / (3.0 * 1.05 * rho), rho=1.05 * rho, n_coarse=na, n_coarse_global=na_glob,
                     R=R_loc)
        levels.append(level)
        th *= 0.5
        next_sharded = na_glob > replicate_max_rows and len(levels) < max_levels - 2 and min(counts) > 0
        if next_sharded:
            # level l + 1 stays sharded: its ghost set = the off-rank columns of my coarse rows and of my prolongator rows
            cols = torch.cat([Kc.col.long(), P_lvl.col.long()])
            ghost_c = torch.unique(cols[(cols < c0) | (cols >= c1)])
            shard.set_level_plan(l + 1, c_ranges, ghost_c.cpu().numpy())
            level["P"], bad_p = _localize_columns(P_lvl, c0, c1, ghost_c, na)
            level["p_ghost_rows"] = PARAMS_GHOST_P
            level["coarse_sharded"] = True
            Kl, bad_k = _localize_columns(Kc, c0, c1, ghost_c, na)
            (n_bad,) = _host_sum(shard, bad_p + bad_k)                     # (collective: every rank raises or none)
            if n_bad:
                raise ValueError("row-sharded AMG: %d column(s) of level %d lie outside the ghost set of their rank" % (n_bad, l + 1))
            ranges, n_glob, l, ghost_l = c_ranges, na_glob, l + 1, ghost_c
            continue
        # level l + 1 is replicated: P keeps global columns, the restriction is the transpose of MY prolongator rows (the
        # ranks' partial right-hand sides are summed by one all-reduce), the coarse operator is all-gathered
        level["P"] = P_lvl
        level["p_ghost_rows"] = PARAMS_GHOST_P
        level["R"] = _dev_transpose(P_glob, na_glob)
        level["coarse_sharded"] = False
        lens = _all_gather_var(shard, (Kc.rowptr[1:] - Kc.rowptr[:-1]).contiguous())
        rp_rep = torch.zeros(na_glob + 1, dtype=torch.int32, device=dev)
        rp_rep[1:] = torch.cumsum(lens.to(torch.int64), 0).to(torch.int32)
        K_rep = DevCSR((na_glob, na_glob), rp_rep, _all_gather_var(shard, Kc.col), _all_gather_var(shard, Kc.val))
        level["n_coarse"] = na_glob
        tail = build_hierarchy_device(K_rep, th, max_levels, coarse_size, level_offset=len(levels))
        return levels + tail


def harmonic_ritz_recycle(Hb, k):
    """Coefficients of the GCRO recycle pair from one (flexible) GMRES cycle  K Z_m = V_{m+1} Hbar_m  (Hb: (m+1) x m raw
    Hessenberg): the k harmonic Ritz vectors g of smallest |theta|,
        (H_m + |h_{m+1,m}|^2 H_m^-H e_m e_m^T) g = theta g,
    give  U = Z_m (G R^-1),  C = V_{m+1} Q  with the thin QR  Hbar_m G = Q R,  so that  K U = C  and  C^H C = I.
    Returns (Ucoef (m x k), Ccoef ((m+1) x k), |theta| of the chosen vectors)."""
    Hb = np.asarray(Hb, dtype=np.complex128)
    m = Hb.shape[1]
    Hm, h = Hb[:m, :m], Hb[m, m - 1]
    em = np.zeros(m, dtype=np.complex128)
    em[-1] = 1.0
    f = np.linalg.solve(Hm.conj().T, em)
    theta, G = np.linalg.eig(Hm + abs(h) ** 2 * np.outer(f, em))
    pick = np.argsort(np.abs(theta), kind="stable")[:k]
    Gk = G[:, pick]
    Q, R = np.linalg.qr(Hb @ Gk)
    return Gk @ np.linalg.inv(R), Q, np.abs(theta[pick])


class AMGSolver:
    """K^-1 b by V-cycle-preconditioned GMRES on the device (hx_gmres)."""

    name = "amg"

    def __init__(self, K_csr, hierarchy=None, shard=None, shifted=None, graph=None, label=None, **overrides):
        """``shard`` (sharded.ShardContext): K_csr holds this rank's rows over [owned | ghost] columns (distributed assembly;
        a global operator is cut to them first), the hierarchy is built and kept row-sharded on its large levels and the
        solve is collective (hx_cocg_sharded / hx_gmres_sharded).

        Escalation ladder (``LADDER``): a solve that stalls does not end the call -- the solver moves to its next rung, stays there
        for the rest of its life and repeats the solve; ``stats()['ladder']`` logs every step with its reason.  The rungs beyond the
        first need ``shifted(beta)`` -> CSR of K + i beta sigma^2 C (the complex-shifted operator the hierarchy is then built on:
        a damped Helmholtz problem every coarse grid can represent; the Krylov method still solves K x = b to the same tolerance)
        and ``graph()`` -> CSR of the Laplacian-like part A on K's pattern (strength of connection and prolongator smoothing
        are then taken from A at every level, whatever the frequency does to the sign of K's entries).  Both are supplied by
        slepc.EPS / PEP; without them the ladder ends after its second rung.  ``label``: text naming the operator in errors."""
        import os
        self.params = dict(PARAMS, **overrides)
        for key in ("pre", "post", "restart", "max_it"):    # experiment hooks (tools/): HX_AMG_PRE / _POST / _RESTART / _MAX_IT
            if os.environ.get("HX_AMG_" + key.upper()):
                self.params[key] = int(os.environ["HX_AMG_" + key.upper()])
        self.shard = shard
        self._K_in, self._hierarchy_in = K_csr, hierarchy
        self._shifted, self._graph, self.label = shifted, graph, label
        self.cocg_solves = self.cocg_fallbacks = self.cocg_iters = self.attainable_stops = 0
        self.backward_stable = False
        self.total_iters = 0
        self.n_solves = 0
        self.solve_s = 0.0
        self.setup_s = 0.0
        self.last = (0, 0.0)
        self.max_iters_in_a_solve = 0
        self.ladder = []                              # [{rung, after_solve, reason}]: the steps taken so far
        self.direct = None                            # the last rung: slepc.LUSolver once every iterative rung has stalled
        self.direct_refused = None
        self.rung = int(os.environ.get("HX_AMG_RUNG", "0"))      # (experiment hook: start on a later rung)
        if not 0 <= self.rung < len(self.LADDER):
            raise ValueError("HX_AMG_RUNG=%d: the escalation ladder has the rungs %s"
                             % (self.rung, ", ".join("%d (%s)" % (i, r["name"]) for i, r in enumerate(self.LADDER))))
        self._apply_rung()
        self._build()

    # ---- escalation ladder: what changes from rung to rung (everything else stays as PARAMS has it)
    #   default          V(2,2) on the hierarchy of K, FGMRES(50) for the first solve, deflated COCG with the <= 8 harmonic Ritz vectors
    #                    harvested from it, at most 200 iterations per solve
    #   wide             the same hierarchy; FGMRES(100), up to 32 recycled vectors (the harvest keeps those with |theta| < 0.45: the
    #                    number of near-kernel modes of the shifted operator grows like target^3), three times the iteration budget.
    #                    Entered when a solve stalls -- or straight after a first solve that needed more than one GMRES cycle, before
    #                    anything fails: measured at 1 M DOF (profiles/r05_envelope_*): 17 500 rad/s 105 -> 61 iterations per solve,
    #                    20 000 rad/s 448 -> 69, 22 000 rad/s no convergence -> 79, 25 000 rad/s no convergence -> 216
    #   shifted          wide + hierarchy of K + i 0.5 sigma^2 C with transfers from the Laplacian part, Krylov method on K itself:
    #                    the last resort.  Measured (EXPERIMENTS.md, round 5): where "wide" converges the shifted hierarchy is SLOWER
    #                    (1 M DOF, 22 000 rad/s: 393 against 144 iterations per solve; beta = 1 with three smoothing steps: 697) and
    #                    at 25 000 rad/s neither converges at 1 M DOF -- what limits the iterative route is the NUMBER of near-kernel
    #                    modes (the deflation panel holds 32), not an unstable smoother or coarse grid; it rescued one case (refined
    #                    real mesh, 25 000 rad/s, 566 iterations per solve) and costs nothing unless it is reached.
    LADDER = (dict(name="default"),
              dict(name="wide", max_it_factor=3, restart=100, recycle=32),
              dict(name="shifted", max_it_factor=4, restart=100, recycle=32, beta=0.5, graph=True))
    REBUILD_KEYS = ("beta", "graph", "sweeps")            # a rung that changes one of these builds another hierarchy
    RESIZE_KEYS = ("restart", "recycle")                  # ... one of these only re-sizes the Krylov workspace and the deflation panel

    def _apply_rung(self):
        import os
        r = self.LADDER[self.rung]
        self.beta = float(os.environ.get("HX_AMG_SHIFT_BETA", r.get("beta", 0.0)))
        self.use_graph = bool(int(os.environ.get("HX_AMG_GRAPH_FROM_A", int(bool(r.get("graph"))))))
        self._max_it = int(self.params["max_it"] * r.get("max_it_factor", 1))
        self._sweeps = r.get("sweeps")
        if not hasattr(self, "_base"):
            self._base = dict(restart=self.params["restart"], recycle=self.params["recycle"])
        for key in ("restart", "recycle"):
            if not os.environ.get("HX_AMG_" + key.upper()):
                self.params[key] = max(int(r.get(key, 0)), int(self._base[key])) if self._base[key] > 0 else 0
        if self.beta and self._shifted is None:
            raise ArithmeticError("rung %r needs the complex-shifted operator, which this caller did not supply" % r["name"])

    def _escalate(self, reason):
        """Next rung (rebuilding the hierarchy where the rung asks for another one).  Returns False at the end of the ladder.
        Row-sharded: the decision rests on all-reduced residual norms, so every rank takes the same step."""
        nxt = self.rung + 1
        if nxt >= len(self.LADDER) or (self.LADDER[nxt].get("beta") and self._shifted is None):
            return False
        rebuild = any(self.LADDER[nxt].get(k) != self.LADDER[self.rung].get(k) for k in self.REBUILD_KEYS)
        resize = any(self.LADDER[nxt].get(k) != self.LADDER[self.rung].get(k) for k in self.RESIZE_KEYS)
        self.rung = nxt
        self.ladder.append(dict(rung=self.LADDER[nxt]["name"], after_solve=self.n_solves, reason=reason))
        info("! inner solver: %s -- moving to rung %r of the escalation ladder" % (reason, self.LADDER[nxt]["name"]))
        self._apply_rung()
        if rebuild:
            self._teardown()
            self._build()
        elif resize:
            self._size_krylov()                   # aggregation, Galerkin products, exchange plans and SELL copies stay as they are
        return True

    def _size_krylov(self):
        """Deflation panel (inside the handle) and Krylov workspace for the rung's restart length / number of recycled pairs;
        drops the harvested pairs -- the next solve harvests again, from the larger space."""
        dev = self.tmp.device
        levels = self.levels
        n_glob = self.n if self.shard is None else self.shard.n_global
        self.recycle_k = int(os.environ.get("HX_AMG_RECYCLE", self.params["recycle"])) if len(levels) > 1 else 0
        if n_glob < self.params["recycle_min_rows"] and "HX_AMG_RECYCLE" not in os.environ:
            self.recycle_k = 0
        self.recycled = 0
        if self.recycle_k > 0:
            _hx.check(_hx.lib.hx_amg_enable_recycling(self.handle, self.n, self.recycle_k, self.params["restart"]))
        # Krylov workspace: with the recycling workspace in the handle the Arnoldi vectors live there (51 x n entries less here)
        gmres_ws = (_hx.lib.hx_gmres_ws_recycling if self.recycle_k > 0 else _hx.lib.hx_gmres_ws)(self.n, self.params["restart"])
        self.ws = None
        self.ws = torch.empty(max(gmres_ws, _hx.lib.hx_cocg_ws(self.n)), dtype=torch.complex128, device=dev)
        if self.shard is not None:
            self._hbuf = torch.zeros(max(self.params["restart"] + 2 + 32, 80), dtype=torch.complex128, device=dev)

    def _build(self):
        import os
        import time
        t_setup = time.perf_counter()
        K_csr, hierarchy, shard = self._K_in, self._hierarchy_in, self.shard
        if self._sweeps:
            self.params["pre"] = self.params["post"] = int(self._sweeps)
        dev = K_csr.val.device
        # the operator the HIERARCHY is built on: K itself, or its complex-shifted companion (same pattern)
        Kp = K_csr if not self.beta else self._shifted(self.beta)
        G = self._graph() if (self.use_graph and self._graph is not None) else None
        host_setup = os.environ.get("HX_AMG_SETUP", self.params["setup"]) == "host"
        # transfers from the Laplacian part exist in the single-GPU device set-up only: say so where the rung asks for them
        self.graph_ignored = bool(self.use_graph and (G is None or shard is not None or hierarchy is not None or host_setup))
        if self.graph_ignored:
            info("! inner solver: rung %r asks for transfer operators from the stiffness part; %s builds them from the operator itself"
                 % (self.LADDER[self.rung]["name"], "the row-sharded set-up" if shard is not None else
                    ("the host set-up" if host_setup else "this caller (no stiffness part supplied)")))
        if shard is not None:
            # row-sharded: K_csr is this rank's rows (distributed assembly) or the global operator (cut to the rank's rows
            # here); either way the hierarchy is BUILT sharded -- every rank its slab of every large level
            K_csr = shard.localize(K_csr)
            Kp = K_csr if Kp is self._K_in else shard.localize(Kp)
            levels = hierarchy if hierarchy is not None else build_hierarchy_sharded(
                Kp, shard, self.params["theta"], self.params["max_levels"], self.params["coarse_size"],
                self.params["replicate_max_rows"])
            n_rows = shard.n_global
        elif hierarchy is not None:
            levels = hierarchy
        elif os.environ.get("HX_AMG_SETUP", self.params["setup"]) == "host":      # round-1 host passes (comparison / tests)
            levels = build_hierarchy(Kp.to_scipy(), self.params["theta"], self.params["max_levels"],
                                     self.params["coarse_size"])
        else:
            levels = build_hierarchy_device(Kp, self.params["theta"], self.params["max_levels"],
                                            self.params["coarse_size"], graph=G)
        self.levels = levels
        self.n_sharded = sum(1 for lv in levels if lv.get("sharded"))
        # Krylov method of the inner solve: complex symmetric K (every shifted operator of the path except the
        # Bloch-reduced ones) + symmetric V-cycle (pre == post) -> deflated COCG (short recurrences, hx_cocg);
        # otherwise flexible GMRES.  "auto" probes the symmetry of the bilinear form with two random vectors.
        # (Rounds 1-2 kept GMRES below 1e5 rows, where the solve is launch-latency bound; since the reductions of hx_cocg are
        # summed by their consumers and its residual norm is read one iteration late, COCG wins at every size measured --
        # 16 k rows: 0.21 -> 0.14 s per fixed-point call, 34 k: 0.26 -> 0.17, 66 k: 0.37 -> 0.27, the reference's 34,787-DOF
        # MICCA mesh: 0.50 -> 0.40 s, tools/r3_small.sh -- and cocg_min_rows is 0.)
        kry = os.environ.get("HX_AMG_KRYLOV", self.params["krylov"])
        if shard is None:
            n_rows = K_csr.shape[0]
        self.use_cocg = len(levels) > 1 and self.params["pre"] == self.params["post"] and (
            kry == "cocg" or (kry == "auto" and n_rows >= self.params["cocg_min_rows"]
                              and (_is_complex_symmetric(K_csr) if shard is None
                                   else _is_complex_symmetric_sharded(K_csr, shard))))
        if shard is not None and len(levels) < 2:
            raise ValueError("row-sharded AMG needs at least one coarse level (problem too small to shard)")
        self.n = K_csr.shape[0]
        self.K = K_csr
        omega = self.params["jacobi_omega"] or min(lv["omega"] for lv in levels if "omega" in lv) if len(levels) > 1 else 0.6
        h = C.c_void_p()
        _hx.check(_hx.lib.hx_amg_create(C.byref(h), len(levels), self.params["pre"], self.params["post"], float(omega)))
        self.handle = h
        _hx.check(_hx.lib.hx_amg_set_real_transfers(h, 1))     # P, R: real values, 8 bytes per entry with the column
        self._keep = []                               # device arrays referenced by the handle
        mixed = self.params["precision"] == "mixed"   # complex64 V-cycle inside complex128 flexible GMRES
        cdt = torch.complex64 if mixed else torch.complex128
        set_level = _hx.lib.hx_amg_set_level_c if mixed else _hx.lib.hx_amg_set_level
        set_coarse = _hx.lib.hx_amg_set_coarse_c if mixed else _hx.lib.hx_amg_set_coarse

        def dev_csr(M):
            """(rowptr, col, val) device tensors of a level operator (DevCSR born on the device, or scipy CSR from the
            host set-up) in the hierarchy's working precision."""
            if isinstance(M, DevCSR):
                if M.val.is_complex():
                    val = M.val.to(cdt)
                else:               # transfer operators are real and stay real (hx_amg_set_real_transfers)
                    val = M.val.to(torch.float32 if mixed else torch.float64)
                t = (M.rowptr, M.col, val)
                self._keep.append(t)
                return t, M.nnz
            M = sp.csr_matrix(M)
            M.sort_indices()
            if np.isrealobj(M.data):
                # transfer operators are real: ship the real parts in the working precision (4x / 2x less H2D traffic
                # than complex128) and widen to complex on the device
                val = to_device(M.data.astype(np.float32 if mixed else np.float64))
            else:
                val = to_device(M.data.astype(np.complex64 if mixed else np.complex128))
            t = (to_device(M.indptr.astype(np.int32, copy=False)), to_device(M.indices.astype(np.int32, copy=False)), val)
            self._keep.append(t)
            return t, int(M.nnz)

        def values_c128(M):
            return M.val if isinstance(M, DevCSR) else to_device(sp.csr_matrix(M).data.astype(np.complex128))

        # SELL-64 companion of the fine operator (complex128 for K z in GMRES, working precision for the V-cycle)
        self.sell = self.sell_pre = None
        self.coarse_sells = []
        # (threshold: CSR with several lanes per row wins below ~1e5 rows -- 66 k DOF 0.20 s per call against 0.24 s with SELL-64,
        # 260 k DOF 0.39 against 0.35; row-sharded levels are judged by their GLOBAL size and keep SELL-64, whose slice lists the
        # halo overlap needs)
        sell_rows = lambda lv: (lv.get("n_global", lv["n"]) if lv.get("sharded") else lv["n"])
        if sell_rows(levels[0]) >= self.params["sell_min_rows"] and len(levels) > 1:
            sh = C.c_void_p()
            # complex128 copy for K z in GMRES; V-cycle copy in the working precision -- stored as scaled complex-half
            # when the cycle runs in complex64 (8 instead of 12 bytes per entry; same GMRES iteration counts measured)
            half = mixed and os.environ.get("HX_AMG_VALUES", self.params["vcycle_values"]) == "fp16"
            self.fine_values = "fp16" if half else ("c64" if mixed else "c128")
            if Kp is K_csr:
                _hx.check(_hx.lib.hx_sell_create(C.byref(sh), self.n, _hx.ptr(K_csr.rowptr), _hx.ptr(K_csr.col),
                                                 _hx.ptr(K_csr.val), 1, 2 if half else int(mixed), _hx.stream()))
                self.sell = sh
                _hx.check(_hx.lib.hx_amg_set_level_sell(h, 0, sh))
            else:
                # hierarchy of the complex-shifted companion: the Krylov method multiplies by K (complex128 copy only), the
                # V-cycle smooths with K + i beta sigma^2 C (working-precision copy only) -- two SELL-64 objects, same pattern
                _hx.check(_hx.lib.hx_sell_create(C.byref(sh), self.n, _hx.ptr(K_csr.rowptr), _hx.ptr(K_csr.col),
                                                 _hx.ptr(K_csr.val), 1, 0, _hx.stream()))
                self.sell = sh
                sp_ = C.c_void_p()
                _hx.check(_hx.lib.hx_sell_create(C.byref(sp_), self.n, _hx.ptr(Kp.rowptr), _hx.ptr(Kp.col), _hx.ptr(Kp.val),
                                                 int(not mixed), (2 if half else int(mixed)), _hx.stream()))
                self.sell_pre = sp_
                _hx.check(_hx.lib.hx_amg_set_level_sell(h, 0, sp_))
        # coarse SELL levels take the same storage (5 M DOF: iterations unchanged, 8.38 -> 8.27 s)
        half_coarse = mixed and os.environ.get("HX_AMG_VALUES", self.params["vcycle_values"]) == "fp16"
        for l, lv in enumerate(levels):
            if l == 0:
                k0 = Kp.val.to(cdt) if mixed else Kp.val
                self._keep.append(k0)
                (krp, kci, kva), knnz = (Kp.rowptr, Kp.col, k0), Kp.nnz
            else:
                (krp, kci, kva), knnz = dev_csr(lv["K"])
                # (one lane per row only pays for short rows: measured 33 us per SELL smoothing step on 88 k rows of
                # ~60 entries, a latency-bound walk, so long-row levels stay on the multi-lane CSR kernels)
                if "P" in lv and sell_rows(lv) >= self.params["sell_min_rows"] and knnz <= self.params["sell_max_row"] * lv["n"]:
                    # coarse operators with enough rows also get a SELL-64 companion (values in the working precision)
                    tmp = values_c128(lv["K"])
                    shl = C.c_void_p()
                    _hx.check(_hx.lib.hx_sell_create(C.byref(shl), lv["n"], _hx.ptr(krp), _hx.ptr(kci), _hx.ptr(tmp),
                                                     int(not mixed), (2 if half_coarse else int(mixed)), _hx.stream()))
                    del tmp
                    self.coarse_sells.append(shl)
                    _hx.check(_hx.lib.hx_amg_set_level_sell(h, l, shl))
            if "P" in lv:
                Pl, Rl, dl, nl = lv["P"], lv["R"], lv["dinv"], lv["n"]     # (sharded levels: already this rank's slabs)
                (prp, pci, pva), pnnz = dev_csr(Pl)
                (rrp, rci, rva), rnnz = dev_csr(Rl)
                dinv = (dl if isinstance(dl, torch.Tensor) else to_device(np.asarray(dl, dtype=np.complex128))).to(cdt)
                self._keep.append(dinv)
                _hx.check(set_level(h, l, nl, knnz, _hx.ptr(krp), _hx.ptr(kci), _hx.ptr(kva), _hx.ptr(dinv),
                                    lv["n_coarse"], pnnz, _hx.ptr(prp), _hx.ptr(pci), _hx.ptr(pva), rnnz,
                                    _hx.ptr(rrp), _hx.ptr(rci), _hx.ptr(rva)))
                if self.params["smoother"] == "chebyshev":
                    _hx.check(_hx.lib.hx_amg_set_smoother(h, l, float(lv["rho"])))
            else:
                _hx.check(set_level(h, l, lv["n"], knnz, _hx.ptr(krp), _hx.ptr(kci), _hx.ptr(kva), None, 0, 0, None,
                                    None, None, 0, None, None, None))
                if isinstance(lv["K"], DevCSR):
                    # dense inverse of the coarsest operator in HBM (hx_zdense_inverse: Gauss-Jordan with partial pivoting on
                    # the device; rocSOLVER through torch.linalg.inv in round 3, np.linalg.inv on the host before), handed to
                    # the V-cycle as a dense CSR
                    Kc, nc = lv["K"], lv["n"]
                    dense = torch.zeros((nc, nc), dtype=torch.complex128, device=dev)
                    lens = (Kc.rowptr[1:] - Kc.rowptr[:-1]).long()
                    rows = torch.repeat_interleave(torch.arange(nc, device=dev), lens)
                    dense[rows, Kc.col.long()] = Kc.val
                    _hx.check(_hx.lib.hx_zdense_inverse(nc, _hx.ptr(dense), _hx.stream()))
                    inv = dense
                    crp = torch.arange(0, nc * nc + 1, nc, dtype=torch.int32, device=dev)
                    cci = torch.arange(nc, dtype=torch.int32, device=dev).repeat(nc)
                    cva = inv.reshape(-1).to(cdt).contiguous()
                    self._keep.append((crp, cci, cva))
                    cnnz = nc * nc
                else:
                    Kh = lv["K"].to_scipy() if isinstance(lv["K"], DevCSR) else lv["K"]
                    (crp, cci, cva), cnnz = dev_csr(sp.csr_matrix(np.linalg.inv(Kh.toarray())))
                _hx.check(set_coarse(h, lv["n"], cnnz, _hx.ptr(crp), _hx.ptr(cci), _hx.ptr(cva)))
        if shard is not None:
            self._cb = shard.callbacks()             # (halo_fn, allreduce_fn, ctx): the library's own RCCL calls, or gloo trampolines
            # (deep-halo levels: vectors carry two ghost layers, n_ext2 = owned + layer 1 + layer 2)
            n_ext = np.ascontiguousarray([lv.get("n_ext2", lv["n_ext"]) for lv in levels[:self.n_sharded]], dtype=np.int32)
            _hx.check(_hx.lib.hx_amg_set_sharded(h, self.n_sharded, _hx.nptr(n_ext), self._cb[0], self._cb[1], self._cb[2]))
            n_pg = 0
            while n_pg < self.n_sharded and levels[n_pg].get("p_ghost_rows"):
                n_pg += 1
            if n_pg:
                _hx.check(_hx.lib.hx_amg_set_prolongator_ghost_rows(h, n_pg))
            self.deep_levels = 0
            for l, lv in enumerate(levels[:n_pg]):
                if "K1" in lv:                      # operator rows of the ghost layers: the level runs the deep-halo cycle
                    K1 = lv["K1"]
                    v1 = K1.val.to(cdt).contiguous()
                    self._keep.append((K1.rowptr, K1.col, v1))
                    _hx.check(_hx.lib.hx_amg_set_level_deep(h, l, lv["deep_layers"], lv["n_g1"], lv["nnz_g1"], lv["n_gs"], K1.nnz,
                                                            _hx.ptr(K1.rowptr), _hx.ptr(K1.col), _hx.ptr(v1)))
                    self.deep_levels += 1
            if shard.comm is not None and os.environ.get("HX_HALO_OVERLAP", "1") != "0":
                # library-issued exchanges: split every halo around the interior slices of its product (exchange on the
                # compute stream like every other operation of the communicator, interior slices on its side stream)
                self._ov = (C.cast(_hx.lib.hx_comm_halo_begin, _hx.HALO_BEGIN_FN), C.cast(_hx.lib.hx_comm_halo_end, _hx.HALO_END_FN))
                _hx.check(_hx.lib.hx_amg_set_halo_overlap(h, self._ov[0], self._ov[1]))
            self._xe = shard.ext_vector(torch.complex128)
            self._hbuf = torch.zeros(max(self.params["restart"] + 2 + 32, 80), dtype=torch.complex128, device=dev)
        # hipGraph replay of the V-cycle: off by default (graph_max_rows = 0; slower than the folded eager cycle at every size, EXPERIMENTS.md round 5)
        gmax = int(os.environ.get("HX_AMG_GRAPH_MAX_ROWS", self.params["graph_max_rows"]))
        if mixed and shard is None and len(levels) > 1 and self.n <= gmax:
            _hx.check(_hx.lib.hx_amg_enable_graph(h, 1))
        self.tmp = torch.empty(self.n, dtype=torch.complex128, device=dev)
        # subspace recycling (all solves of a fixed-point call share K): the harvest keeps the modes the spectrum asks for (3 on
        # the small synthetic annuli, 6 on the reference's MICCA mesh, 6-8 from 1 M DOF on).  Rounds 1-2 enabled it from 1e5 rows;
        # with COCG a deflation vector costs bytes, not launches, and small problems are launch-bound: 34,787-DOF MICCA mesh
        # 0.45 -> 0.35 s per call (25.1 -> 17.3 iterations per solve), 66 k DOF 0.27 -> 0.25, 16 k / 34 k unchanged
        # (tools/r3_small.sh): recycle_min_rows is 0.
        self._size_krylov()
        self.krylov_tol_floor = 10.0 * self.params["rtol"]
        torch.cuda.synchronize()
        self.setup_s += time.perf_counter() - t_setup
        self.agg_times = list(AGG_TIMES) if hierarchy is None else []
        del AGG_TIMES[:]
        self.operator_complexity = round(sum(lv["K"].nnz for lv in levels) / levels[0]["K"].nnz, 3)
        self.level_nnz = [int(lv["K"].nnz) for lv in levels]
        info("- AMG hierarchy: " + " -> ".join("%d" % lv["n"] for lv in levels)
             + "  (operator complexity %.2f)" % self.operator_complexity)

    supports_x0 = True

    def set_rtol(self, rtol):
        """Relative tolerance of the inner solves (default 1e-11).  The eigen-solver lowers it when the caller asks the fixed
        point for more digits than 1e-11 solves leave in the eigenvalue (``fixed_point_iteration(tol=1e-8)`` at |omega| ~ 10^3 is a
        relative 6e-12: the reference's direct solve delivers it; see eigensolvers._inner_rtol)."""
        self.params["rtol"] = float(rtol)
        self.krylov_tol_floor = 10.0 * self.params["rtol"]

    def set_max_it(self, max_it):
        """Iteration limit of one inner solve on the first rung (-st_ksp_max_it); later rungs multiply it."""
        self.params["max_it"] = int(max_it)
        self._apply_rung()

    def solve(self, b, x, relax=1.0, x0=False):
        """``relax`` >= 1 multiplies the relative tolerance (inexact-Krylov relaxation decided by the eigen-solver).
        ``x0``: x holds an initial guess (a related solution); used by the COCG route only, and dropped by the solver when
        it does not bring the residual below a tenth of |b| (hx_amg_next_solve_uses_x).
        A solve that ends above 100x its tolerance (and not at a rounding-level backward error) moves the solver up its
        escalation ladder and is repeated there; only the last rung raises -- with the operator's label in the message."""
        while True:
            if self.direct is not None:
                self.n_solves += 1
                return self.direct.solve(b, x)
            it, rr = self._solve_once(b, x, relax, x0)
            rtol = self.params["rtol"] * max(float(relax), 1.0)
            if not (rr > 100 * rtol and not getattr(self, "backward_stable", False)):
                return
            reason = "inner solve %d stalled at relative residual %.2e after %d iterations" % (self.n_solves, rr, it)
            x0 = False
            if not self._escalate(reason) and not self._go_direct(reason):
                raise ArithmeticError("%s on the last rung (%r) of the escalation ladder%s; steps taken: %s%s"
                                      % (reason, self.LADDER[self.rung]["name"],
                                         " -- operator: %s" % self.label if self.label else "",
                                         [st["rung"] for st in self.ladder] or "none",
                                         "; direct factorisation: %s" % self.direct_refused if self.direct_refused else ""))

    def _go_direct(self, reason):
        """Beyond the last iterative rung: factorise K on the device (slepc.LUSolver, csrc/splu.hip) when the factor fits in HBM --
        what PCLU / MUMPS is to the reference: an answer at any target.  Row-sharded runs replicate the factor (slepc.ReplicatedLUSolver)."""
        if os.environ.get("HX_AMG_DIRECT_RUNG", "1") == "0":
            self.direct_refused = "switched off"
            return False
        from .slepc import LUSolver, ReplicatedLUSolver
        self._teardown()                              # the hierarchy's memory goes to the factor
        _hx.check(_hx.lib.hx_pool_trim())
        try:                                          # (row-sharded: every rank holds the factor of the GLOBAL operator; collective)
            self.direct = LUSolver(self._K_in) if self.shard is None else ReplicatedLUSolver(self._K_in, self.shard)
        except MemoryError as e:
            self.direct_refused = str(e)
            self._build()
            return False
        self.tmp = self.direct.tmp                    # (callers size their work vectors after the inner solver's)
        self.ladder.append(dict(rung="direct", after_solve=self.n_solves, reason=reason))
        info("! inner solver: %s -- moving to the direct factorisation (nnz(L) = %.3g, %.2f s)"
             % (reason, self.direct.info()[0][3], self.direct.setup_s))
        return True

    def _solve_once(self, b, x, relax, x0):
        import time
        t0 = time.perf_counter()
        it, rr = C.c_int(0), C.c_double(0.0)
        rtol = self.params["rtol"] * max(float(relax), 1.0)
        # deflated COCG once the deflation pairs exist (or when none are wanted); the first solve of a hierarchy runs
        # GMRES, whose Hessenberg matrix yields the harmonic Ritz vectors both methods recycle
        done = False
        if self.use_cocg and (self.recycle_k == 0 or self.recycled):
            if x0:
                _hx.check(_hx.lib.hx_amg_next_solve_uses_x(self.handle))
                self.warm_starts = getattr(self, "warm_starts", 0) + 1
            done = self._solve_cocg(b, x, rtol, it, rr)
        if not done:
            self._solve_gmres(b, x, rtol, it, rr)
            be = C.c_double(-1.0)
            if rr.value > 10 * rtol:
                _hx.check(_hx.lib.hx_cocg_backward_error(self.handle, C.byref(be)))
            self.backward_stable = 0.0 <= be.value <= 10 * 2.220446049250313e-16
            self.attainable_stops += int(self.backward_stable)
        if self.recycle_k > 0 and not self.recycled:
            if (self.rung == 0 and not done and it.value > self.params["restart"] and rr.value <= 100 * rtol
                    and os.environ.get("HX_AMG_EARLY_WIDE", "1") != "0"):
                # the harvest solve needed more than one GMRES cycle: the operator has more slow modes than the default Krylov
                # space and deflation panel hold.  Move to the "wide" rung NOW (the solution just computed stands; the next solve
                # harvests from a 100-dimensional space) instead of paying 100+ iterations per solve until something stalls.
                self.solve_s += time.perf_counter() - t0
                self.total_iters += it.value
                self.n_solves += 1
                self.last = (it.value, rr.value)
                self._escalate("first solve needed %d iterations (more than one FGMRES(%d) cycle)" % (it.value, self.params["restart"]))
                return it.value, rr.value
            self._harvest()
        self.solve_s += time.perf_counter() - t0        # hx_gmres synchronises internally
        self.total_iters += it.value
        self.n_solves += 1
        self.last = (it.value, rr.value)
        self.max_iters_in_a_solve = max(self.max_iters_in_a_solve, it.value)
        if os.environ.get("HX_AMG_LOG_ITS"):
            print("[amg] solve %d: %d its, relres %.2e" % (self.n_solves, it.value, rr.value), file=__import__("sys").stderr)
        return it.value, rr.value

    # ---- blocks of right-hand sides (csrc/cocg_block.hip)
    def block_ready(self):
        """The batched route: single GPU, SELL-64 fine level with complex128 values, complex64 V-cycle, COCG in use and its
        deflation pairs prepared by an earlier solve (or none wanted)."""
        return (self.direct is None and self.shard is None and self.sell is not None and self.use_cocg and self.params["precision"] == "mixed"
                and os.environ.get("HX_BLOCK_SOLVES", "1") != "0"
                and (self.recycle_k == 0 or (self.recycled and self.cocg_solves > 0)))

    def solve_block(self, B, X, relax=1.0):
        """K X[c] = B[c] for the rows of the row-major panels B, X (nb x n): groups of hx_block_width() columns run side by side
        (hx_cocg_block); a column the batched recurrence leaves above the acceptance band is finished by ``solve`` from where it
        stands.  Falls back to one ``solve`` per row when the batched route is not available."""
        import time
        nb = B.shape[0]
        if not self.block_ready():
            for c in range(nb):
                self.solve(B[c], X[c], relax=relax)
            return
        t0 = time.perf_counter()
        w = _hx.lib.hx_block_width()
        rtol = self.params["rtol"] * max(float(relax), 1.0)
        if getattr(self, "_blk_ws", None) is None:
            self._blk_ws = _hx.device_empty(int(_hx.lib.hx_cocg_block_ws(self.n)), torch.complex128, B.device)
            self._blk_b = _hx.device_empty(self.n * w, torch.complex128, B.device)
            self._blk_x = _hx.device_empty(self.n * w, torch.complex128, B.device)
        its, rrs = (C.c_int * w)(), (C.c_double * w)()
        redo = []
        for c0 in range(0, nb, w):
            m = min(w, nb - c0)
            _hx.check(_hx.lib.hx_zblock_pack(self.n, m, _hx.ptr(B[c0]), B.stride(0), _hx.ptr(self._blk_b), _hx.stream()))
            rc = _hx.lib.hx_cocg_block(self.handle, self.n, self.sell, m, _hx.ptr(self._blk_b), _hx.ptr(self._blk_x), rtol,
                                       self._max_it, _hx.ptr(self._blk_ws), its, rrs, _hx.stream())
            if rc == _hx.ERR_BREAKDOWN:                  # a recurrence broke down: the whole group goes the single-vector way
                redo += [(c, False) for c in range(c0, c0 + m)]
                continue
            _hx.check(rc)
            _hx.check(_hx.lib.hx_zblock_unpack(self.n, m, _hx.ptr(self._blk_x), _hx.ptr(X[c0]), X.stride(0), _hx.stream()))
            self.block_solves = getattr(self, "block_solves", 0) + 1
            self.block_columns = getattr(self, "block_columns", 0) + m
            self.block_iters = getattr(self, "block_iters", 0) + max(its[c] for c in range(m))
            for c in range(m):
                self.total_iters += its[c]
                self.n_solves += 1
                if not (rrs[c] <= 10 * rtol):
                    redo.append((c0 + c, True))
        self.solve_s += time.perf_counter() - t0
        for c, warm in redo:
            self.solve(B[c], X[c], relax=relax, x0=warm)

    def _solve_gmres(self, b, x, rtol, it, rr):
        K = self.K
        if self.shard is not None:
            if self.shard.comm is not None:          # the library's communicator takes over: nothing of torch's may be in flight
                from .distributed import assert_no_outstanding_work
                assert_no_outstanding_work("hx_gmres_sharded")
            rc = _hx.lib.hx_gmres_sharded(self.handle, self.n, K.nnz, _hx.ptr(K.rowptr), _hx.ptr(K.col), _hx.ptr(K.val),
                                          self.sell, _hx.ptr(b), _hx.ptr(x), rtol, self._max_it,
                                          self.params["restart"], _hx.ptr(self.ws), _hx.ptr(self._xe),
                                          _hx.ptr(self._hbuf), C.byref(it), C.byref(rr), _hx.stream())
            if rc != 0 and self.shard.callback_error():
                raise RuntimeError("exchange callback failed inside hx_gmres_sharded:\n" + self.shard.callback_error())
            _hx.check(rc)
        else:
            _hx.check(_hx.lib.hx_gmres(self.handle, self.n, K.nnz, _hx.ptr(K.rowptr), _hx.ptr(K.col), _hx.ptr(K.val),
                                       self.sell, _hx.ptr(b), _hx.ptr(x), rtol, self._max_it,
                                       self.params["restart"], _hx.ptr(self.ws), C.byref(it), C.byref(rr), _hx.stream()))

    def _solve_cocg(self, b, x, rtol, it, rr):
        """True when hx_cocg delivered the solution; False = breakdown / stall of the bilinear recurrence (the caller
        repeats the solve with GMRES; after three such solves COCG is switched off for this operator)."""
        K = self.K
        if self.shard is not None:
            native = self.shard.comm is not None
            if native:                      # exchanges issued inside this solve (library-issued path: counted by the communicator)
                from .distributed import assert_no_outstanding_work
                assert_no_outstanding_work("hx_cocg_sharded")
                ne0, na0 = C.c_int64(0), C.c_int64(0)
                _hx.check(_hx.lib.hx_comm_counters(self.shard.comm, C.byref(ne0), C.byref(na0), 0))
            rc = _hx.lib.hx_cocg_sharded(self.handle, self.n, K.nnz, _hx.ptr(K.rowptr), _hx.ptr(K.col), _hx.ptr(K.val),
                                         self.sell, _hx.ptr(b), _hx.ptr(x), rtol, self._max_it,
                                         _hx.ptr(self.ws), _hx.ptr(self._xe),
                                         _hx.ptr(self._hbuf), C.byref(it), C.byref(rr), _hx.stream())
            if native:
                ne1, na1 = C.c_int64(0), C.c_int64(0)
                _hx.check(_hx.lib.hx_comm_counters(self.shard.comm, C.byref(ne1), C.byref(na1), 0))
                self.cocg_halos = getattr(self, "cocg_halos", 0) + ne1.value - ne0.value
                self.cocg_allreduces = getattr(self, "cocg_allreduces", 0) + na1.value - na0.value
            if rc != 0 and self.shard.callback_error():
                raise RuntimeError("exchange callback failed inside hx_cocg_sharded:\n" + self.shard.callback_error())
        else:
            rc = _hx.lib.hx_cocg(self.handle, self.n, K.nnz, _hx.ptr(K.rowptr), _hx.ptr(K.col), _hx.ptr(K.val),
                                 self.sell, _hx.ptr(b), _hx.ptr(x), rtol, self._max_it, _hx.ptr(self.ws),
                                 C.byref(it), C.byref(rr), _hx.stream())
        be = C.c_double(-1.0)
        if rc == 0 and rr.value > 10 * rtol:
            _hx.check(_hx.lib.hx_cocg_backward_error(self.handle, C.byref(be)))
        self.backward_stable = 0.0 <= be.value <= 10 * 2.220446049250313e-16
        # accepted: the true residual within 10x the tolerance (the band hx_gmres uses too), or an iterate whose backward
        # error is at rounding level; anything else is repeated with GMRES
        if rc == 0 and (rr.value <= 10 * rtol or self.backward_stable):
            self.cocg_solves += 1
            self.cocg_iters += it.value
            self.attainable_stops += int(self.backward_stable)
            return True
        if rc not in (0, _hx.ERR_BREAKDOWN):
            _hx.check(rc)
        self.cocg_fallbacks += 1
        self.total_iters += it.value                 # the abandoned iterations were paid for
        info("! hx_cocg gave up after %d iterations at relative residual %.2e (%s): repeating the solve with GMRES"
             % (it.value, rr.value, _hx.lib.hx_last_error().decode(errors="replace") if rc else "iteration limit"))
        if self.cocg_fallbacks >= 3:
            self.use_cocg = False
        return False

    def _harvest(self):
        """Install the k harmonic Ritz vectors of smallest |theta| of the solve that just finished as recycled pairs
        (U, C = K U): with K Z_m = V_{m+1} Hbar_m,  U = Z_m G R^-1, C = V_{m+1} Q  where Hbar_m G = Q R.
        The small dense algebra runs on the host from all-reduced numbers, i.e. identically on every rank."""
        m = C.c_int(0)
        _hx.check(_hx.lib.hx_gmres_hessenberg(self.handle, C.byref(m), None))
        m = m.value
        k = min(self.recycle_k, m - 2)              # a short first solve yields fewer pairs
        if k < min(4, self.recycle_k):
            return                                  # restarted / too short: try again after the next solve
        Hb = np.zeros((m + 1) * m, dtype=np.complex128)
        _hx.check(_hx.lib.hx_gmres_hessenberg(self.handle, C.byref(C.c_int(0)), _hx.nptr(Hb)))
        Hb = Hb.reshape(m, m + 1).T                 # column-major -> (m+1, m)
        Uc, Cc, theta = harmonic_ritz_recycle(Hb, k)
        # keep the modes that actually slow the preconditioned iteration down: |theta| (eigenvalue estimates of M K, 1 =
        # perfectly preconditioned) below the threshold.  A deflated vector costs two panel columns per iteration (1.7 % of
        # an iteration at 5 M DOF), and beyond the first few modes the estimates form a continuum that moves the
        # convergence rate little: measured |theta| at 5 M DOF 0.08 0.18 0.18 0.35 0.37 0.41 | 0.46 0.49 ..., at 1 M DOF
        # 0.14 0.25 0.27 0.42 | 0.46 0.49 ...; the fastest fixed counts were 6 and 4-5 -- what the threshold picks.
        tau = float(os.environ.get("HX_AMG_RECYCLE_THETA", self.params["recycle_theta"]))
        k_eff = max(min(k, int(np.sum(theta < tau))), min(3, k))
        if k_eff < k:
            k = k_eff
            Uc, Cc, theta = harmonic_ritz_recycle(Hb, k)
        Uc, Cc = np.asfortranarray(Uc), np.asfortranarray(Cc)
        _hx.check(_hx.lib.hx_amg_set_recycle(self.handle, k, _hx.nptr(Uc), _hx.nptr(Cc), _hx.ptr(self.ws), _hx.stream()))
        self.recycled = k
        self.harmonic_ritz = theta

    def stats(self):
        if self.direct is not None:               # the hierarchy is gone: what the direct rung reports, under the same keys
            out = dict(rung="direct", ladder=list(self.ladder), solves=self.n_solves, setup_s=round(self.setup_s, 3),
                       iters_per_solve=round(self.total_iters / max(self.n_solves, 1), 1), direct=self.direct.stats())
            return out
        return dict(rung=self.LADDER[self.rung]["name"], ladder=list(self.ladder), shift_beta=self.beta,
                    transfers_from="A" if (self.use_graph and self._graph is not None and not getattr(self, "graph_ignored", False)) else "K",
                    graph_transfers_ignored=bool(getattr(self, "graph_ignored", False)),
                    recycled_pairs=self.recycled, vcycle_graph=bool(_hx.lib.hx_amg_graph_active(self.handle)),
                    levels=[lv.get("n_global", lv["n"]) for lv in self.levels],
                    operator_complexity=getattr(self, "operator_complexity", None), level_nnz=getattr(self, "level_nnz", None),
                    sharded_levels=[[lv["n"], lv["n_ext"] - lv["n"]] for lv in self.levels if lv.get("sharded")],
                    deep_halo_levels=[[lv["n_g1"], lv["n_ext2"] - lv["n_ext"]] for lv in self.levels if lv.get("sharded") and "n_ext2" in lv],
                    deep_halo_layers=max([lv.get("deep_layers", 0) for lv in self.levels if lv.get("sharded")] or [0]),
                    fine_format=("sell64/%s/i%d" % (self.fine_values, _hx.lib.hx_sell_index_bits(self.sell)))
                    if self.sell else "csr",
                    stencil_slices=_hx.lib.hx_sell_stencil_slices(self.sell) if self.sell else 0, vcycle_precision="c64" if self.params["precision"] == "mixed" else "c128",
                    setup_s=round(self.setup_s, 3),
                    setup_host_aggregation=[dict(rows=a, entries=b, copy_s=c, host_pass_s=d) for a, b, c, d in self.agg_times],
                    solves=self.n_solves,
                    krylov="cocg" if self.use_cocg else "fgmres", cocg_solves=self.cocg_solves,
                    cocg_iters=self.cocg_iters, cocg_fallbacks=self.cocg_fallbacks,
                    attainable_stops=self.attainable_stops,
                    gmres_iters=self.total_iters, solve_s=round(self.solve_s, 3),
                    iters_per_solve=round(self.total_iters / max(self.n_solves, 1), 1),
                    max_iters_in_a_solve=self.max_iters_in_a_solve,
                    harmonic_ritz=[round(float(t), 3) for t in getattr(self, "harmonic_ritz", [])],
                    warm_starts=getattr(self, "warm_starts", 0),
                    cocg_halos=getattr(self, "cocg_halos", 0), cocg_allreduces=getattr(self, "cocg_allreduces", 0),
                    block_solves=getattr(self, "block_solves", 0), block_columns=getattr(self, "block_columns", 0),
                    block_iters=getattr(self, "block_iters", 0))

    def _teardown(self):
        """Handles and device arrays of the current hierarchy (destroy(); also between two rungs of the ladder)."""
        if getattr(self, "handle", None):
            _hx.lib.hx_amg_destroy(self.handle)
            self.handle = None
        for name in ("sell", "sell_pre"):
            if getattr(self, name, None):
                _hx.lib.hx_sell_destroy(getattr(self, name))
                setattr(self, name, None)
        for shl in getattr(self, "coarse_sells", []):
            _hx.lib.hx_sell_destroy(shl)
        self.coarse_sells = []
        # the device arrays the handles pointed into: released with them, not at the next garbage collection
        for name in ("ws", "tmp", "_blk_ws", "_blk_b", "_blk_x", "_xe", "_hbuf", "K", "_keep"):
            if hasattr(self, name):
                setattr(self, name, None)
        if getattr(self, "levels", None):
            self.levels = [{k: v for k, v in lv.items() if isinstance(v, (int, float, bool, str))} for lv in self.levels]

    def destroy(self):
        self._teardown()
        if getattr(self, "direct", None) is not None:
            self.direct.destroy()
            self.direct = None
        self._K_in = self._hierarchy_in = self._shifted = self._graph = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
