"""Row-sharded building blocks for the multi-GPU path: one process per GPU,
``torch.distributed`` over RCCL (backend "nccl"); the same code runs under "gloo" for the CPU tests.

What is here (round 1):
  * contiguous row partition (theta-slabs for the annulus numbering, which is theta-major);
  * ``shard_mesh``: the cells incident to a rank's vertices + one ghost layer, renumbered owned-first, so the
    vertex-centric assembly kernels produce the rank's matrix rows with NO assembly communication
    (owner computes; the reference ships off-process rows through MatAssemblyBegin/End, acoustic_matrices.py:103);
  * ``HaloPlan``: who needs which of my entries; exchange = one batched isend/irecv with the (<= 2 for slabs)
    neighbouring ranks -- point-to-point over xGMI, not a ring collective;
  * ``ShardedOperator``: y_own = A_own,: [x_own; x_ghost]  (halo exchange + local hx_csr_spmv_z);
  * ``sharded_panel_dot``: local fused CGS panel + ONE all-reduce of (k+1) complex numbers per Krylov step.
The collective fixed-point solve built on these blocks lives in sharded.py.
"""
import numpy as np
import torch
import torch.distributed as dist


def partition_rows(n, world, align=1):
    """Balanced contiguous ranges [r0, r1) per rank, boundaries rounded to a multiple of ``align``
    (use align = (nr+1)*(nz+1) to cut the annulus between theta-planes)."""
    cuts = [0]
    for p in range(1, world):
        c = int(round(p * n / world / align)) * align
        cuts.append(min(max(c, cuts[-1]), n))
    cuts.append(n)
    return [(cuts[p], cuts[p + 1]) for p in range(world)]


def shard_mesh(x, cells, r0, r1):
    """Local mesh of the rank owning vertices [r0, r1).

    Returns dict(x, cells, n_own, local_to_global): vertices renumbered owned-first (ascending global id), then the
    ghost vertices (ascending global id); cells = all cells touching an owned vertex, in ascending global cell id,
    so every owned row receives exactly its global element contributions in the global order."""
    cells = np.asarray(cells)
    own_mask = (cells >= r0) & (cells < r1)
    keep = np.flatnonzero(own_mask.any(axis=1))
    lc = cells[keep]
    verts = np.unique(lc)
    owned = verts[(verts >= r0) & (verts < r1)]
    if owned.size != r1 - r0:                      # owned vertices not touched by any cell (isolated): keep them
        owned = np.arange(r0, r1)
    ghost = verts[(verts < r0) | (verts >= r1)]
    l2g = np.concatenate([owned, ghost])
    order = np.argsort(l2g, kind="stable")
    local_cells = order[np.searchsorted(l2g[order], lc.ravel())].reshape(lc.shape)
    return dict(x=np.ascontiguousarray(x[l2g]), cells=local_cells.astype(np.int32), n_own=int(owned.size),
                local_to_global=l2g, cell_ids=keep)


class HaloPlan:
    """Ghost columns of a rank, grouped by owner, and the matching send lists (built with one all_gather of the
    ghost id lists -- integer set-up work)."""

    def __init__(self, ranges, rank, ghost_global, group=None):
        self.rank, self.world, self.group = rank, len(ranges), group
        self.r0, self.r1 = ranges[rank]
        ghost_global = np.asarray(ghost_global, dtype=np.int64)
        assert np.all(np.diff(ghost_global) > 0), "ghost ids must be strictly ascending"
        starts = np.array([r[0] for r in ranges] + [ranges[-1][1]])
        owner = np.searchsorted(starts, ghost_global, side="right") - 1
        self.recv_from = {int(p): np.flatnonzero(owner == p) for p in np.unique(owner)}   # positions in ghost block
        wanted = [None] * self.world
        if self.world > 1:
            dist.all_gather_object(wanted, ghost_global, group=group)
        else:
            wanted = [ghost_global]
        self.send_to = {}
        for p, ids in enumerate(wanted):
            if p == rank or ids is None:
                continue
            mine = ids[(ids >= self.r0) & (ids < self.r1)]
            if mine.size:
                self.send_to[p] = (mine - self.r0).astype(np.int64)
        self.n_ghost = int(ghost_global.size)
        self._dev_idx = {}
        self._send_span = {}         # neighbour -> (a, b) if its send list is the contiguous owned range [a, b), else False
        self._send_buf = {}

    def exchange(self, x_ext, n_own):
        """Fill x_ext[n_own:] (ghost block) with the owners' current values of x_ext[:n_own] (complex64 or complex128).
        Ghost ids ascend and owners hold contiguous ranges, so each neighbour's entries form one contiguous slice of
        the ghost block: received in place, no scatter."""
        if self.world == 1 or (not self.send_to and not self.recv_from):
            return
        backend = dist.get_backend(self.group)
        stage = backend == "gloo" and x_ext.is_cuda       # gloo moves host buffers only
        ops, staged, keep = [], [], []
        for p, idx in sorted(self.send_to.items()):
            span = self._send_span.get(p)
            if span is None:                       # first use: is the send list one contiguous range of owned rows?
                span = (int(idx[0]), int(idx[-1]) + 1) if (len(idx) and idx[-1] - idx[0] + 1 == len(idx)
                                                           and np.all(np.diff(idx) == 1)) else False
                self._send_span[p] = span
            if span:                               # slabs cut between mesh planes: send the slice itself, no gather
                src = x_ext[span[0]:span[1]]
            else:
                key = (p, x_ext.device)
                if key not in self._dev_idx:
                    self._dev_idx[key] = torch.from_numpy(idx).to(x_ext.device)
                bkey = (p, x_ext.dtype, x_ext.device)
                if bkey not in self._send_buf:
                    self._send_buf[bkey] = torch.empty(len(idx), dtype=x_ext.dtype, device=x_ext.device)
                src = torch.index_select(x_ext[:n_own], 0, self._dev_idx[key], out=self._send_buf[bkey])
            buf = torch.view_as_real(src)
            buf = buf.cpu() if stage else buf
            keep.append(buf)
            ops.append(dist.P2POp(dist.isend, buf, p, group=self.group))
        for p, pos in sorted(self.recv_from.items()):
            a, b = int(pos[0]), int(pos[-1]) + 1
            assert b - a == len(pos), "ghost entries of one owner must be contiguous"
            dst = torch.view_as_real(x_ext[n_own + a:n_own + b])
            if stage:
                buf = torch.empty(dst.shape, dtype=dst.dtype, device="cpu")
                staged.append((dst, buf))
            else:
                buf = dst
            ops.append(dist.P2POp(dist.irecv, buf, p, group=self.group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        for dst, buf in staged:
            dst.copy_(buf)
        self.n_exchanges = getattr(self, "n_exchanges", 0) + 1


class ShardedOperator:
    """Rows [r0, r1) of a global operator as a local device CSR over [owned | ghost] columns."""

    def __init__(self, local_csr, n_own, plan):
        self.A, self.n_own, self.plan = local_csr, n_own, plan
        self.n_ext = local_csr.shape[1]

    def mult(self, x_ext, y_own):
        """y_own = A [x_own; x_ghost]; x_ext must have room for the ghost block (length n_own + n_ghost)."""
        self.plan.exchange(x_ext, self.n_own)
        self.A.spmv(x_ext, y_own)


def sharded_panel_dot(V, k, w, group=None):
    """Global [V_j^H w (j<k), w^H w] from row-sharded V (k x n_own) and w: local fused panel + one all-reduce."""
    from .petsc import panel_dot
    local = torch.from_numpy(np.ascontiguousarray(panel_dot(V, k, w)))
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        buf = torch.view_as_real(local).contiguous()
        if dist.get_backend(group) == "nccl":
            buf = buf.to(w.device)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        local = torch.view_as_complex(buf.cpu())
    return local.numpy()
