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


def rcb_parts(x, world):
    """Recursive coordinate bisection of points into ``world`` parts of (nearly) equal size: every split cuts the current set along
    the LONGEST axis of its bounding box at the weighted median (part counts need not be powers of two: a set destined for p parts
    is cut in the proportion floor(p/2) : ceil(p/2)).  part[i] in [0, world); deterministic (stable sorts, ties by index).
    What the reference gets from DOLFINx's graph partitioner when it reads a mesh on N ranks (io_utils.py:149-155); METIS / SCOTCH
    are not available here, and a geometric partition of a FEM mesh has compact parts with few neighbours, which is what the halo
    exchange wants (point-to-point over xGMI, one message per neighbour)."""
    x = np.asarray(x, dtype=np.float64)
    part = np.zeros(x.shape[0], dtype=np.int32)
    stack = [(np.arange(x.shape[0]), 0, int(world))]
    while stack:
        ids, first, p = stack.pop()
        if p <= 1 or ids.size == 0:
            part[ids] = first
            continue
        pl = p // 2
        box = x[ids].max(axis=0) - x[ids].min(axis=0)
        axis = int(np.argmax(box))
        order = ids[np.argsort(x[ids, axis], kind="stable")]
        cut = int(round(ids.size * pl / p))
        stack.append((order[:cut], first, pl))
        stack.append((order[cut:], first + pl, p - pl))
    return part


def rcb_renumbering(mesh, world, local_order="rcm"):
    """(perm, ranges): perm[old] = new vertex id such that part p of ``rcb_parts`` owns the contiguous range ranges[p] -- the form
    ShardContext / distribute_mesh take -- and, inside every part, the vertices follow the reverse Cuthill-McKee order of the
    part's own sub-graph (``local_order="rcm"``: neighbouring rows have neighbouring columns, which the SELL-64 kernels and their
    16-bit column deltas rely on) or keep their relative order (``"keep"``).  Integer host work at ingestion, identical on every
    rank.  Use with ``mesh.renumbered(perm)``."""
    import scipy.sparse as sp
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    x = mesh.geometry.x
    n = x.shape[0]
    part = rcb_parts(x, world)
    counts = np.bincount(part, minlength=world)
    starts = np.concatenate([[0], np.cumsum(counts)])
    ranges = [(int(starts[p]), int(starts[p + 1])) for p in range(world)]
    order = np.argsort(part, kind="stable")                       # order[new] = old, parts contiguous, relative order kept
    if local_order == "rcm":
        cells = np.asarray(mesh.topology.cells, dtype=np.int64)
        k = cells.shape[1]
        rows = np.repeat(cells, k, axis=1).ravel()
        cols = np.tile(cells, (1, k)).ravel()
        same = part[rows] == part[cols]                           # edges inside a part only
        G = sp.coo_matrix((np.ones(int(same.sum()), np.int8), (rows[same], cols[same])), shape=(n, n)).tocsr()
        del rows, cols
        for p in range(world):
            ids = order[starts[p]:starts[p + 1]]
            if ids.size > 1:
                sub = G[ids][:, ids]
                order[starts[p]:starts[p + 1]] = ids[reverse_cuthill_mckee(sub, symmetric_mode=True)]
    elif local_order != "keep":
        raise ValueError("local_order must be 'rcm' or 'keep'")
    perm = np.empty(n, dtype=np.int64)
    perm[order] = np.arange(n)
    return perm, ranges


def partition_quality(cells, ranges):
    """Per rank of a contiguous-range partition: ghost vertices (columns owned elsewhere that its rows touch) and neighbour ranks
    -- the size and the number of the messages of one halo exchange.  Host arrays; for bench records and tests."""
    cells = np.asarray(cells, dtype=np.int64)
    starts = np.array([r[0] for r in ranges] + [ranges[-1][1]])
    owner = np.searchsorted(starts, cells, side="right") - 1     # (nc, k)
    out = []
    for p in range(len(ranges)):
        touch = (owner == p).any(axis=1)
        v = cells[touch]
        o = owner[touch]
        ghost = np.unique(v[o != p])
        nb = np.unique(o[o != p])
        out.append(dict(rows=int(ranges[p][1] - ranges[p][0]), ghosts=int(ghost.size), neighbours=int(nb.size)))
    return out


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


class DistComm:
    """``mesh.comm`` of a distributed local mesh: the few host-side reductions the reference makes through
    ``MPI.COMM_WORLD.allreduce`` (boundary areas, flame volumes, normalisation integrals: acoustic_matrices.py:76-90,
    parameters_utils.py:216,236, dolfinx_utils.py:43,53) -- scalars or small arrays, summed (or max-ed) over the ranks."""

    def __init__(self, group=None):
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.size = dist.get_world_size(group) if dist.is_initialized() else 1

    def allreduce(self, v, op="sum"):
        if self.size == 1:
            return v
        a = np.atleast_1d(np.asarray(v))
        t = torch.from_numpy(np.ascontiguousarray(a.astype(np.complex128 if np.iscomplexobj(a) else np.float64)))
        buf = torch.view_as_real(t) if t.is_complex() else t
        if dist.get_backend(self.group) == "nccl":
            dev = buf.cuda()
            dist.all_reduce(dev, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM, group=self.group)
            buf.copy_(dev.cpu())
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM, group=self.group)
        out = t.numpy()
        return out[0].item() if np.ndim(v) == 0 else out.reshape(np.shape(v))


def distribute_mesh(mesh, ranges, rank, cell_tags=None, facet_tags=None, fields=(), group=None):
    """This rank's part of a mesh that every rank holds as host arrays (what ``XDMFFile.read_mesh`` hands each MPI rank of
    the reference: its cells plus one ghost layer, io_utils.py:149-155): the cells touching a vertex of [r0, r1), vertices
    renumbered [owned | ghost] (``shard_mesh``), the tagged facets with an owned vertex, tags and coefficient fields cut to
    the local entities.  The assembly kernels run on it unchanged and produce this rank's matrix ROWS over
    [owned | ghost] columns with no assembly communication (owner computes).

    Returns (local_mesh, local_cell_tags, local_facet_tags, local_fields).  ``local_mesh.part`` carries the maps:
    r0, r1, n_own, n_global, local_to_global, cell_ids, cell_owned (a cell belongs to the rank that owns its lowest
    vertex: used for volume / area sums), facet_ids, ranges."""
    from .fem import Function, FunctionSpace, Mesh, MeshTags
    r0, r1 = ranges[rank]
    x, cells = mesh.geometry.x, mesh.topology.cells
    sh = shard_mesh(x, cells, r0, r1)
    l2g, keep, n_own = sh["local_to_global"], sh["cell_ids"], sh["n_own"]
    g2l = np.full(x.shape[0], -1, dtype=np.int64)
    g2l[l2g] = np.arange(l2g.size)
    fv = mesh.facets
    fkeep = np.flatnonzero(((fv >= r0) & (fv < r1)).any(axis=1)) if fv.size else np.zeros(0, np.int64)
    lf = np.sort(g2l[fv[fkeep]], axis=1) if fkeep.size else np.zeros((0, fv.shape[1] if fv.ndim == 2 else 1), np.int64)
    fcell = np.searchsorted(keep, mesh.facet_cells[fkeep])
    assert fkeep.size == 0 or (np.array_equal(keep[fcell], mesh.facet_cells[fkeep]) and lf.min() >= 0)
    local = Mesh(sh["x"], sh["cells"], facets=lf.astype(np.int32), facet_cells=fcell.astype(np.int32))
    low = cells[keep].min(axis=1)
    local.part = dict(r0=r0, r1=r1, n_own=n_own, n_global=x.shape[0], local_to_global=l2g, cell_ids=keep,
                      cell_owned=(low >= r0) & (low < r1), facet_ids=fkeep, ranges=list(ranges), rank=rank,
                      n_cells_global=cells.shape[0])
    flow = fv[fkeep].min(axis=1) if fkeep.size else np.zeros(0, np.int64)
    local.part["facet_owned"] = (flow >= r0) & (flow < r1)
    local.comm = DistComm(group)

    def cut_tags(tags, ids):
        if tags is None:
            return None
        pos = np.searchsorted(ids, tags.indices)
        ok = (pos < ids.size)
        ok[ok] = ids[pos[ok]] == tags.indices[ok]
        return MeshTags(local, tags.dim, pos[ok], tags.values[ok])

    out_fields = []
    for f in fields:
        kind = f.function_space.kind
        g = Function(FunctionSpace(local, "DG", 0) if kind == "DG0" else FunctionSpace(local, "CG", 1), f.name)
        g.x.array[:] = f.x.array[keep] if kind == "DG0" else f.x.array[l2g]
        out_fields.append(g)
    return local, cut_tags(cell_tags, keep), cut_tags(facet_tags, fkeep), out_fields


# ---- torch.distributed work handles of this module
# Under the nccl backend a rank holds TWO NCCL communicators: torch's process group (the point-to-point messages and object
# collectives of the SET-UP below) and the library's own hx_comm (every exchange of the solves, csrc/comm.hip).  NCCL does not
# order operations of different communicators: a torch request still in flight when an hx_comm operation is enqueued may meet its
# peers in a different order on another rank -- a deadlock no single-GPU stand-in can show.  The rule: every request issued here is
# waited for before control can reach the library's communicator.  ``issue`` / ``wait_all`` keep the requests in a registry and
# ``assert_no_outstanding_work`` (called at every Python entry into hx_comm: ShardContext.halo / all_reduce_ / the sharded solves)
# raises when one was left behind.
_OUTSTANDING = []


def issue(ops):
    """dist.batch_isend_irecv with the returned work handles registered as outstanding."""
    reqs = dist.batch_isend_irecv(ops)
    _OUTSTANDING.extend(reqs)
    return reqs


def wait_all(reqs):
    for req in reqs:
        req.wait()
        for k, r in enumerate(_OUTSTANDING):
            if r is req:
                del _OUTSTANDING[k]
                break


def assert_no_outstanding_work(where):
    if _OUTSTANDING:
        raise RuntimeError("%d torch.distributed request(s) issued by the set-up are still outstanding at %s: a request must be "
                           "waited for before the library's own communicator (hx_comm) is used -- NCCL does not order two "
                           "communicators" % (len(_OUTSTANDING), where))


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
        self.n_exchanges = 0
        self._dev_idx = {}
        self._send_span = {}         # neighbour -> (a, b) if its send list is the contiguous owned range [a, b), else False
        self._send_buf = {}

    def exchange(self, x_ext, n_own):
        """Fill x_ext[n_own:] (ghost block) with the owners' current values of x_ext[:n_own] (complex64 / complex128 iterates;
        int32 / float64 vectors of the hierarchy set-up).
        Ghost ids ascend and owners hold contiguous ranges, so each neighbour's entries form one contiguous slice of
        the ghost block: received in place, no scatter."""
        if self.world == 1 or (not self.send_to and not self.recv_from):
            return
        ops, staged, keep = self.p2p_ops(x_ext, n_own, n_own)
        wait_all(issue(ops))
        for dst, buf in staged:
            dst.copy_(buf)
        self.n_exchanges += 1

    def p2p_ops(self, x_ext, n_own, ghost_base):
        """The point-to-point operations of one exchange (sends of listed owned entries, receives into the ghost block that starts
        at ``ghost_base``): (ops, [(device destination, host buffer)] to copy back under gloo, buffers to keep alive)."""
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
            buf = torch.view_as_real(src) if src.is_complex() else src
            buf = buf.cpu() if stage else buf
            keep.append(buf)
            ops.append(dist.P2POp(dist.isend, buf, p, group=self.group))
        for p, pos in sorted(self.recv_from.items()):
            a, b = int(pos[0]), int(pos[-1]) + 1
            assert b - a == len(pos), "ghost entries of one owner must be contiguous"
            dst = x_ext[ghost_base + a:ghost_base + b]
            dst = torch.view_as_real(dst) if dst.is_complex() else dst
            if stage:
                buf = torch.empty(dst.shape, dtype=dst.dtype, device="cpu")
                staged.append((dst, buf))
            else:
                buf = dst
            ops.append(dist.P2POp(dist.irecv, buf, p, group=self.group))
        return ops, staged, keep

    def native_arrays(self):
        """The plan as the HOST arrays hx_comm_set_plan takes: send lists (owned indices) per peer, receive ranges of the
        extended vector per peer (ghost entries of one owner are contiguous: ghost ids ascend, owners hold ranges)."""
        n_own = self.r1 - self.r0
        sp = sorted(self.send_to)
        send_peer = np.array(sp, dtype=np.int32)
        send_ptr = np.zeros(len(sp) + 1, dtype=np.int64)
        send_ptr[1:] = np.cumsum([len(self.send_to[p]) for p in sp])
        send_idx = np.concatenate([self.send_to[p] for p in sp]).astype(np.int32) if sp else np.zeros(0, np.int32)
        rp_ = sorted(self.recv_from)
        recv_peer = np.array(rp_, dtype=np.int32)
        recv_off = np.array([n_own + int(self.recv_from[p][0]) for p in rp_], dtype=np.int64)
        recv_cnt = np.array([len(self.recv_from[p]) for p in rp_], dtype=np.int64)
        for p in rp_:
            pos = self.recv_from[p]
            assert int(pos[-1]) - int(pos[0]) + 1 == len(pos), "ghost entries of one owner must be contiguous"
        return send_peer, send_ptr, send_idx, recv_peer, recv_off, recv_cnt

    def exchange_rows(self, rowptr, col, val):
        """Sparse-row halo of the set-up: CSR rows (device: int32 rowptr over the OWNED rows, int32 col, val) of this rank's
        owned entities listed in the send lists go to the ranks holding them as ghosts; returns the rows of MY ghost entities
        in ghost order as (rowptr, col, val) device tensors.  Three messages per neighbour (row lengths, columns, values):
        host-driven torch.distributed point-to-point -- set-up only, a handful of calls per hierarchy level."""
        dev = col.device
        backend = dist.get_backend(self.group) if dist.is_initialized() else "none"
        stage = backend == "gloo"
        lens_own = (rowptr[1:] - rowptr[:-1]).to(torch.int64)
        sends, keep = [], []
        for p, idx in sorted(self.send_to.items()):
            ii = torch.from_numpy(idx).to(dev)
            ln = lens_own[ii]
            start = rowptr[:-1].to(torch.int64)[ii]
            tot = int(ln.sum().item())
            # entry positions of the listed rows, row after row
            off = torch.cumsum(ln, 0) - ln
            pos = torch.repeat_interleave(start - off, ln) + torch.arange(tot, device=dev, dtype=torch.int64)
            sends.append((p, ln.to(torch.int32).contiguous(), col[pos].contiguous(), val[pos].contiguous()))
        # 1) row lengths
        ops, recv_len = [], {}
        for p, ln, _, _ in sends:
            b = ln.cpu() if stage else ln
            keep.append(b)
            ops.append(dist.P2POp(dist.isend, b, p, group=self.group))
        for p, pos in sorted(self.recv_from.items()):
            b = torch.empty(len(pos), dtype=torch.int32, device="cpu" if stage else dev)
            recv_len[p] = b
            ops.append(dist.P2POp(dist.irecv, b, p, group=self.group))
        if ops:
            wait_all(issue(ops))
        # 2) columns and values
        ops, recv_col, recv_val = [], {}, {}
        for p, _, c, v in sends:
            vr = torch.view_as_real(v) if v.is_complex() else v
            bc, bv = (c.cpu(), vr.cpu()) if stage else (c, vr)
            keep += [bc, bv]
            ops.append(dist.P2POp(dist.isend, bc, p, group=self.group))
            ops.append(dist.P2POp(dist.isend, bv, p, group=self.group))
        for p in sorted(self.recv_from):
            tot = int(recv_len[p].sum().item())
            bc = torch.empty(tot, dtype=torch.int32, device="cpu" if stage else dev)
            shape = (tot, 2) if val.is_complex() else (tot,)
            bv = torch.empty(shape, dtype=torch.float64 if val.dtype in (torch.complex128, torch.float64) else torch.float32,
                             device="cpu" if stage else dev)
            recv_col[p], recv_val[p] = bc, bv
            ops.append(dist.P2POp(dist.irecv, bc, p, group=self.group))
            ops.append(dist.P2POp(dist.irecv, bv, p, group=self.group))
        if ops:
            wait_all(issue(ops))
        order = sorted(self.recv_from, key=lambda p: int(self.recv_from[p][0]))      # ghost order = ascending position
        lens = torch.cat([recv_len[p].to(dev) for p in order]) if order else torch.zeros(0, dtype=torch.int32, device=dev)
        rp = torch.zeros(lens.numel() + 1, dtype=torch.int32, device=dev)
        if lens.numel():
            rp[1:] = torch.cumsum(lens.to(torch.int64), 0).to(torch.int32)
        gc = torch.cat([recv_col[p].to(dev) for p in order]) if order else torch.zeros(0, dtype=torch.int32, device=dev)
        if order:
            gv = torch.cat([recv_val[p].to(dev) for p in order])
            gv = torch.view_as_complex(gv.contiguous()) if val.is_complex() else gv
        else:
            gv = torch.zeros(0, dtype=val.dtype, device=dev)
        return rp, gc, gv


class ShardedOperator:
    """Rows [r0, r1) of a global operator as a local device CSR over [owned | ghost] columns."""

    def __init__(self, local_csr, n_own, plan):
        self.A, self.n_own, self.plan = local_csr, n_own, plan
        self.n_ext = local_csr.shape[1]

    def mult(self, x_ext, y_own):
        """y_own = A [x_own; x_ghost]; x_ext must have room for the ghost block (length n_own + n_ghost)."""
        self.plan.exchange(x_ext, self.n_own)
        self.A.spmv(x_ext, y_own)


class DeepHaloPlan:
    """SEVERAL ghost layers of a level in ONE exchange (the deep-halo cycle of a row-sharded level, amg.hip: vcycle_deep): layer 1 =
    the level's ordinary ghost set (the off-rank columns of the rank's operator rows, positions [n_own, n_own + n1) of an extended
    vector), layer k + 1 = the off-rank columns of the OPERATOR ROWS OF LAYER k that are in no earlier layer (positions behind
    it).  ``layers``: one HaloPlan per layer (each built on the ascending global ids of its layer).  All layers travel in one batch
    of point-to-point messages -- per peer one message per layer, issued in layer order on both sides -- i.e. ONE exchange in the
    counters where the classic schedule needs one per smoothing step."""

    def __init__(self, layers, group=None):
        self.layers = list(layers)
        self.p1 = self.layers[0]
        self.sizes = [p.n_ghost for p in self.layers]
        self.n_ghost = int(sum(self.sizes))
        self.n_exchanges = 0
        self.world, self.group = self.p1.world, group
        self.send_to = {(k, p): v for k, pl in enumerate(self.layers) for p, v in pl.send_to.items()}
        self.recv_from = {(k, p): v for k, pl in enumerate(self.layers) for p, v in pl.recv_from.items()}

    def exchange(self, x_ext, n_own):
        if self.world == 1 or (not self.send_to and not self.recv_from):
            return
        ops, staged, keep, base = [], [], [], n_own
        for pl, nk in zip(self.layers, self.sizes):
            o, st, kp = pl.p2p_ops(x_ext, n_own, base)
            ops += o
            staged += st
            keep += kp
            base += nk
        if ops:
            wait_all(issue(ops))
        for dst, buf in staged:
            dst.copy_(buf)
        self.n_exchanges += 1

    def native_arrays(self):
        """All layers as ONE plan of hx_comm_set_plan: the peers of layer 1, then those of layer 2, ...  A peer appears once per
        layer; hx_comm_set_plan concatenates its lists in this order on both sides into ONE message per peer and direction (the
        receiving side scatters it to the layers' ranges), so nothing rests on how the communication library matches several
        messages between one pair of ranks."""
        parts = [pl.native_arrays() for pl in self.layers]
        send_peer = np.concatenate([a[0] for a in parts])
        send_ptr, off = [np.zeros(1, np.int64)], 0
        for a in parts:
            send_ptr.append(off + a[1][1:])
            off += int(a[1][-1])
        send_ptr = np.concatenate(send_ptr)
        send_idx = np.concatenate([a[2] for a in parts])
        recv_peer = np.concatenate([a[3] for a in parts])
        shift = np.concatenate([[0], np.cumsum(self.sizes)[:-1]])
        recv_off = np.concatenate([a[4] + int(sh) for a, sh in zip(parts, shift)])
        recv_cnt = np.concatenate([a[5] for a in parts])
        return send_peer, send_ptr, send_idx, recv_peer, recv_off, recv_cnt


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
