"""Row-sharded fixed-point solve: one process per GPU, ``torch.distributed`` over RCCL.

What the reference gets from PETSc/SLEPc under ``mpirun -np N`` (a row-distributed Mat, KSP and BV; eigensolvers.py
runs unchanged on N ranks) is provided here by a ``ShardContext`` handed to ``fixed_point_iteration(..., shard=ctx)``:

  * every rank owns a contiguous row range [r0, r1) of the fine operators (theta-slabs for the annulus numbering);
  * fine-level products  y_own = K[r0:r1, :] x  use ONE halo exchange (point-to-point with the ranks owning the ghost
    columns -- two neighbours for slabs) followed by the local CSR / SELL-64 kernel;
  * inner products (CGS2 panels of the Krylov-Schur basis and of GMRES, Woodbury sparse dots) are local fused
    kernels + one small all-reduce;
  * AMG preconditioner: level 0 (the one that carries ~85 % of the rows) is row-sharded, the coarse levels are
    REPLICATED on every rank -- the restriction sums the ranks' partial coarse right-hand sides with one all-reduce and
    the whole coarse part of the V-cycle then runs without any exchange.  The library drives the iteration
    (hx_gmres_sharded / hx_amg_vcycle in C++) and calls back into this module only for the two exchange steps.

Set-up is replicated: every rank assembles the (cheap, device) global operators and builds the same hierarchy on its
host, then keeps only its rows.  That part does not scale with N; DESIGN.md section 7 has the numbers.
"""
import traceback

import torch
import torch.distributed as dist

from . import _hx
from .distributed import HaloPlan, partition_rows
from .petsc import CSR


class ShardContext:
    """Row partition + communicator.  ``align`` cuts between mesh planes (annulus: (nr+1)*(nz+1) rows per plane)."""

    def __init__(self, n_global, group=None, align=1, ranges=None):
        if not dist.is_initialized():
            raise RuntimeError("ShardContext needs an initialised torch.distributed process group")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.n_global = int(n_global)
        self.ranges = ranges or partition_rows(self.n_global, self.world, align)
        self.r0, self.r1 = self.ranges[self.rank]
        self.n_own = self.r1 - self.r0
        if min(b - a for a, b in self.ranges) <= 0:
            raise ValueError("every rank must own at least one row")
        self.backend = dist.get_backend(group)
        self.plan = None            # HaloPlan of the fine sparsity (set by the first localize())
        self.ghost = None           # ascending global ids of the ghost columns (device, int64)
        self.n_ext = self.n_own
        self.n_allreduce = 0

    # ---- collectives
    def all_reduce_(self, t):
        """In-place SUM over ranks of a (complex or real) device tensor; identical bits on every rank."""
        self.n_allreduce += 1
        v = torch.view_as_real(t) if t.is_complex() else t
        if self.backend == "gloo" and v.is_cuda:
            h = v.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            v.copy_(h)
        else:
            dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def halo(self, x_ext):
        self.plan.exchange(x_ext, self.n_own)

    def reset_counters(self):
        self.n_allreduce = 0
        if self.plan is not None:
            self.plan.n_exchanges = 0

    def counters(self):
        return {"halo_exchanges": getattr(self.plan, "n_exchanges", 0) if self.plan is not None else 0,
                "all_reduces": self.n_allreduce}

    # ---- operators
    def set_pattern(self, K):
        """Fix the ghost set from the widest fine pattern (K = A + sigma B + sigma^2 C contains all the others)."""
        col = K.col[int(K.rowptr[self.r0]):int(K.rowptr[self.r1])].long()
        ghost = torch.unique(col[(col < self.r0) | (col >= self.r1)])          # sorted ascending
        self.ghost = ghost
        self.plan = HaloPlan(self.ranges, self.rank, ghost.cpu().numpy(), self.group)
        self.n_ext = self.n_own + int(ghost.numel())

    def localize(self, M):
        """Rows [r0, r1) of the global device CSR ``M`` over [owned | ghost] columns (integer work on the device)."""
        if self.plan is None:
            self.set_pattern(M)
        a, b = int(M.rowptr[self.r0]), int(M.rowptr[self.r1])
        rp = (M.rowptr[self.r0:self.r1 + 1] - a).to(torch.int32).contiguous()
        col = M.col[a:b].long()
        own = (col >= self.r0) & (col < self.r1)
        pos = torch.searchsorted(self.ghost, col)
        pos_c = pos.clamp(max=max(int(self.ghost.numel()) - 1, 0))
        if self.ghost.numel():
            if not bool(torch.all(own | (self.ghost[pos_c] == col))):
                raise ValueError("operator has columns outside the ghost set fixed by set_pattern()")
        elif not bool(torch.all(own)):
            raise ValueError("operator has off-rank columns but the ghost set is empty")
        lcol = torch.where(own, col - self.r0, self.n_own + pos_c).to(torch.int32).contiguous()
        out = CSR((self.n_own, self.n_ext), rp, lcol, M.val[a:b].contiguous())
        return out

    def ext_vector(self, dtype=torch.complex128):
        return torch.zeros(self.n_ext, dtype=dtype, device=self.ghost.device if self.ghost is not None else _hx.require_gpu())

    def own_slice(self, t):
        """Owned block of a global device vector / the columns of a (k, n_global) panel."""
        return t[..., self.r0:self.r1]


class ShardedCallbacks:
    """ctypes trampolines handed to hx_amg_set_sharded; ``bufs`` maps the library's buffer ids to torch tensors."""

    def __init__(self, shard, bufs):
        self.shard, self.bufs = shard, bufs
        self.error = None

        def halo(_ctx, buf):
            try:
                shard.halo(self.bufs[buf])
                return 0
            except Exception:                      # never let an exception cross the C boundary
                self.error = traceback.format_exc()
                return 1

        def allreduce(_ctx, buf, count):
            try:
                shard.all_reduce_(self.bufs[buf][:count])
                return 0
            except Exception:
                self.error = traceback.format_exc()
                return 1

        self.halo = _hx.HALO_FN(halo)
        self.allreduce = _hx.ALLREDUCE_FN(allreduce)


def bench_hot_path(case, st_type=None, nev=4, tol=1e-3, history=None):
    """bench.hot_path(..., sharded=True): one pass of the hot path on the ONE problem partitioned over the ranks of the
    default process group.  Leaves the ShardContext in case["shard"], this rank's rows of the shifted operator in
    case["K_local"] and the set-up mode in case["setup_mode"]."""
    import numpy as np
    from .acoustic_matrices import AcousticMatrices
    from .eigensolvers import fixed_point_iteration
    from .flame_matrices import PointwiseFlameMatrix
    from .parameters_utils import Q_multiple
    from .petsc import combine_csr
    P, mesh = case["P"], case["mesh"]
    mesh.assembler().rebuild_maps()
    matrices = AcousticMatrices(mesh, case["facet_tags"], {11: {'Robin': P.R_outlet}}, case["c"])
    h = case.get("h")
    if h is None:
        h = case["h"] = Q_multiple(mesh, case["subdomains"], P.N_sector).resident()
    D = PointwiseFlameMatrix(mesh, case["subdomains"], P.x_r(), h, P.rho_amb, P.q_0, P.u_b, case["ftf"])
    D.assemble_submatrices('direct')
    x = mesh.geometry.x
    th = np.arctan2(x[:, 1], x[:, 0])
    plane = int(np.argmax(np.abs(th - th[0]) > 1e-12)) if case.get("numbering", "native") == "native" else 1
    shard = case["shard"] = ShardContext(matrices.A.shape[0], align=max(plane, 1))
    E = fixed_point_iteration(matrices, D, P.target, i=0, nev=nev, tol=tol, st_type=st_type or "amg", history=history,
                              shard=shard)
    A, B, Cm = matrices.A, matrices.B, matrices.C
    K = combine_csr(A.shape, A.csr_terms + [(P.target * c, m) for c, m in B.csr_terms]
                    + [(P.target ** 2 * c, m) for c, m in Cm.csr_terms])
    case["K_local"] = shard.localize(K)
    case["setup_mode"] = "replicated (every rank assembles the global operators and builds the whole hierarchy)"
    return E, matrices
