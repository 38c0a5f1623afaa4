"""Row-sharded fixed-point solve: one process per GPU, RCCL over xGMI.

What the reference gets from DOLFINx / PETSc / SLEPc under ``mpirun -np N`` (a distributed mesh, row-distributed Mat, KSP and
BV; the scripts run unchanged on N ranks: numerical_examples/AnnularCombustor/Micca/fullAnnulus/runAll.sh:9-10) is provided
here by a ``ShardContext``:

  * every rank owns a contiguous row range [r0, r1) (theta-slabs for the annulus numbering) and holds only ITS part of the
    mesh (``distributed.distribute_mesh``: its cells + one ghost layer); the unchanged assembly kernels produce its rows of
    A, B, C over [owned | ghost] columns with no assembly communication (owner computes, acoustic_matrices.py:101-103);
  * operator products  y_own = K[r0:r1, :] x  use ONE halo exchange (point-to-point with the ranks owning the ghost columns:
    two neighbours for slabs) followed by the local CSR / SELL-64 kernel;
  * inner products (CGS2 panels of the Krylov-Schur basis and of GMRES, COCG scalars, Woodbury sparse dots) are local fused
    kernels + one small all-reduce;
  * the AMG hierarchy is BUILT sharded (amg.build_hierarchy_sharded: uncoupled aggregation, prolongator slabs, Galerkin
    products from owned rows + sparse-row halos) and its large levels STAY sharded; only the levels below
    ``replicate_max_rows`` are replicated (one all-reduce of that level's right-hand side per V-cycle).

Transport: under the ``nccl`` backend the exchanges are issued BY THE LIBRARY on the compute stream (``hx_comm``: RCCL
ncclSend/ncclRecv groups and ncclAllReduce, csrc/comm.hip) -- no Python frame and no host synchronisation per exchange; the
communicator id travels once through torch.distributed.  Under ``gloo`` (CPU tests, single-GPU rehearsals) the same plans
run through torch.distributed with host staging, via the callback interface of include/hx.h.
"""
import ctypes as C
import os
import traceback

import numpy as np
import torch
import torch.distributed as dist

from . import _hx
from .distributed import DeepHaloPlan, HaloPlan, assert_no_outstanding_work, issue, partition_rows, wait_all
from .petsc import CSR


class _DevView:
    """Raw device pointer -> torch tensor (``__cuda_array_interface__``): the gloo callback path wraps the library's buffers."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


_ELEM = {4: torch.float32, 8: torch.complex64, 16: torch.complex128}


class ShardContext:
    """Row partition + communicator + one exchange plan per sharded hierarchy level.
    ``align`` cuts between mesh planes (annulus: (nr+1)*(nz+1) rows per plane)."""

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
        self.ranks_on_device = _hx.count_device_sharing(group)      # (collective) sizes the device-buffer pool's cap
        self.plans = {}             # level -> HaloPlan
        self.level_n = {0: (self.n_own, self.n_own)}      # level -> (n_own, n_ext)
        self.level_ranges = {0: self.ranges}
        self.ghost = None           # level 0: ascending global ids of the ghost columns (device, int64)
        self.n_ext = self.n_own
        self.n_allreduce = 0
        self._views = {}
        self._cb = None
        # RCCL issued by the library: under nccl, or forced (HX_NATIVE_COMM=1 with HX_RCCL_LIB: the test double)
        want = os.environ.get("HX_NATIVE_COMM")
        self.comm = None
        self.comm_library = None
        if (self.backend == "nccl" and want != "0") or want in ("1", "try"):
            self._create_native_comm(required=(want == "1"))

    # ---- native communicator
    def _create_native_comm(self, required=False):
        """The library's own RCCL communicator (hx_comm).  Collective: the ranks agree on the outcome -- if librccl cannot be bound
        or ncclCommInitRank fails on ANY rank, every rank drops its communicator and the exchanges go through torch.distributed
        (same plans, driven from Python: slower, same results), with a warning; ``required`` (HX_NATIVE_COMM=1) raises instead."""
        ident, err, h = [None], None, None
        if self.rank == 0:
            try:
                buf = (C.c_char * _hx.COMM_ID_BYTES)()
                _hx.check(_hx.lib.hx_comm_unique_id(buf, _hx.COMM_ID_BYTES))
                ident = [bytes(buf.raw)]
            except Exception as e:                        # (librccl not loadable, symbol missing, ...)
                err = e
        dist.broadcast_object_list(ident, src=0, group=self.group)
        if ident[0] is not None:
            try:
                _hx.require_gpu()
                h = C.c_void_p()
                _hx.check(_hx.lib.hx_comm_create(C.byref(h), self.rank, self.world, ident[0], _hx.COMM_ID_BYTES))
            except Exception as e:
                err, h = e, None
        ok = torch.tensor([1.0 if h is not None else 0.0], device=_hx.require_gpu() if self.backend == "nccl" else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
        if float(ok.item()) < 1.0:
            if h is not None:
                _hx.lib.hx_comm_destroy(h)
            if required:
                raise RuntimeError("HX_NATIVE_COMM=1: the library's RCCL communicator could not be created on every rank: %r" % (err,))
            import sys
            print("helmholtz_x.sharded: WARNING rank %d: no library-issued RCCL communicator (%r): the exchanges go through "
                  "torch.distributed" % (self.rank, err), file=sys.stderr)
            return
        self.comm = h
        self.comm_library = _hx.lib.hx_comm_library().decode()

    def destroy(self):
        if self.comm is not None:
            _hx.lib.hx_comm_destroy(self.comm)
            self.comm = None

    @property
    def plan(self):
        return self.plans.get(0)

    # ---- collectives
    def all_reduce_(self, t, op="sum"):
        """In-place SUM (or MAX) over ranks of a (complex or real, float32/64) device tensor; same bits on every rank."""
        self.n_allreduce += 1
        v = torch.view_as_real(t) if t.is_complex() else t
        if self.comm is not None and v.is_cuda:
            assert v.is_contiguous() and v.dtype in (torch.float32, torch.float64)
            assert_no_outstanding_work("ShardContext.all_reduce_ (hx_comm_allreduce)")
            _hx.check(_hx.lib.hx_comm_allreduce(self.comm, _hx.ptr(v), v.numel(), int(v.dtype == torch.float64),
                                                int(op == "max"), _hx.stream()))
            return t
        rop = dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM
        if self.backend == "gloo" and v.is_cuda:
            h = v.cpu()
            dist.all_reduce(h, op=rop, group=self.group)
            v.copy_(h)
        else:
            dist.all_reduce(v, op=rop, group=self.group)
        return t

    def halo(self, x_ext, level=0):
        """Fill the ghost block of a level's extended vector (any 4-, 8- or 16-byte element type)."""
        if self.comm is not None and x_ext.is_cuda:
            assert_no_outstanding_work("ShardContext.halo (hx_comm_halo)")
            _hx.check(_hx.lib.hx_comm_halo(self.comm, level, _hx.ptr(x_ext), x_ext.element_size(), _hx.stream()))
            return
        self.plans[level].exchange(x_ext, self.level_n[level][0])

    def reset_counters(self):
        self.n_allreduce = 0
        for p in self.plans.values():
            p.n_exchanges = 0
        if self.comm is not None:
            _hx.check(_hx.lib.hx_comm_counters(self.comm, None, None, 1))

    def counters(self):
        out = {"halo_exchanges": sum(p.n_exchanges for p in self.plans.values()), "all_reduces": self.n_allreduce,
               "issued_by": "torch.distributed (%s), host-driven" % self.backend}
        if self.comm is not None:
            ne, na = C.c_int64(0), C.c_int64(0)
            _hx.check(_hx.lib.hx_comm_counters(self.comm, C.byref(ne), C.byref(na), 0))
            out = {"halo_exchanges": out["halo_exchanges"] + ne.value, "all_reduces": na.value,
                   "halos_overlapped_with_interior_slices": int(_hx.lib.hx_comm_overlapped(self.comm)),
                   "communicator_stream_switches": int(_hx.lib.hx_comm_stream_switches(self.comm)),
                   "issued_by": "libhx on the compute stream (%s)" % self.comm_library}
        return out

    # ---- plans
    def set_level_plan(self, level, ranges, ghost_global):
        """Exchange plan of a hierarchy level: ``ranges`` = the ranks' row ranges on that level, ``ghost_global`` = ascending
        global ids of this rank's ghost columns there (host int64 array)."""
        ghost_global = np.asarray(ghost_global, dtype=np.int64)
        plan = HaloPlan(ranges, self.rank, ghost_global, self.group)
        self.plans[level] = plan
        n_own = ranges[self.rank][1] - ranges[self.rank][0]
        self.level_n[level] = (n_own, n_own + int(ghost_global.size))
        self.level_ranges[level] = list(ranges)
        if self.comm is not None:
            sp, sptr, sidx, rp_, roff, rcnt = plan.native_arrays()
            _hx.check(_hx.lib.hx_comm_set_plan(self.comm, level, len(sp), _hx.nptr(sp), _hx.nptr(sptr), _hx.nptr(sidx),
                                               len(rp_), _hx.nptr(rp_), _hx.nptr(roff), _hx.nptr(rcnt)))
        return plan

    DEEP = 32         # exchange plan of a level's TWO ghost layers: number DEEP + level (csrc/amg.hip: vcycle_deep)

    def layer_plan(self, level, ghost_global):
        """HaloPlan of one further ghost layer of a sharded level (ascending global ids, disjoint from the earlier layers): used by
        the set-up to fetch the operator rows of that layer before the deep plan is registered.  Collective."""
        return HaloPlan(self.level_ranges[level], self.rank, np.asarray(ghost_global, dtype=np.int64), self.group)

    def set_level_deep_plan(self, level, extra_layers):
        """Multi-layer exchange of a sharded level: its ordinary ghost set (layer 1) + ``extra_layers`` (HaloPlans from
        ``layer_plan``, in layer order), as ONE plan numbered DEEP + level.  Extended vectors of the level are then
        [owned | layer 1 | layer 2 | ...]."""
        plan = DeepHaloPlan([self.plans[level]] + list(extra_layers), self.group)
        key = self.DEEP + level
        self.plans[key] = plan
        n_own = self.level_n[level][0]
        self.level_n[key] = (n_own, n_own + plan.n_ghost)
        if self.comm is not None:
            sp, sptr, sidx, rp_, roff, rcnt = plan.native_arrays()
            _hx.check(_hx.lib.hx_comm_set_plan(self.comm, key, len(sp), _hx.nptr(sp.astype(np.int32)), _hx.nptr(sptr.astype(np.int64)),
                                               _hx.nptr(sidx.astype(np.int32)), len(rp_), _hx.nptr(rp_.astype(np.int32)),
                                               _hx.nptr(roff.astype(np.int64)), _hx.nptr(rcnt.astype(np.int64))))
        return plan

    def set_pattern(self, K):
        """Fix the level-0 ghost set from the widest fine pattern of a GLOBAL operator (K = A + sigma B + sigma^2 C contains
        all the others)."""
        col = K.col[int(K.rowptr[self.r0]):int(K.rowptr[self.r1])].long()
        ghost = torch.unique(col[(col < self.r0) | (col >= self.r1)])          # sorted ascending
        self.set_ghosts(ghost.cpu().numpy())

    def set_ghosts(self, ghost_global):
        """Level-0 ghost set given directly (distributed mesh: the ghost vertices of the local mesh, ascending)."""
        ghost_global = np.asarray(ghost_global, dtype=np.int64)
        dev = _hx.require_gpu() if torch.cuda.is_available() else torch.device("cpu")
        self.ghost = torch.from_numpy(ghost_global).to(dev)
        self.set_level_plan(0, self.ranges, ghost_global)
        self.n_ext = self.n_own + int(ghost_global.size)

    def distribute(self, mesh, cell_tags=None, facet_tags=None, fields=()):
        """This rank's part of ``mesh`` (distributed.distribute_mesh) wired to this context: the level-0 exchange plan is
        the ghost layer of the local mesh.  Returns (local_mesh, local_cell_tags, local_facet_tags, local_fields); operators
        assembled on ``local_mesh`` are this rank's rows and go straight into ``fixed_point_iteration(..., shard=self)``."""
        from .distributed import distribute_mesh
        local, ct, ft, fl = distribute_mesh(mesh, self.ranges, self.rank, cell_tags, facet_tags, fields, group=self.group)
        self.set_ghosts(local.part["local_to_global"][local.part["n_own"]:])
        local.shard = self
        return local, ct, ft, fl

    # ---- operators
    def localize(self, M):
        """Rows [r0, r1) of the global device CSR ``M`` over [owned | ghost] columns (integer work on the device); an
        operator that is local already (distributed assembly) passes through."""
        if getattr(M, "local", False):
            return M
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
        out.local = True
        return out

    def ext_vector(self, dtype=torch.complex128, level=0):
        dev = self.ghost.device if self.ghost is not None else _hx.require_gpu()
        return torch.zeros(self.level_n[level][1], dtype=dtype, device=dev)

    def own_slice(self, t):
        """Owned block of a global device vector / the columns of a (k, n_global) panel."""
        return t[..., self.r0:self.r1]

    # ---- global cyclic shift of a row-sharded vector (translated Woodbury guesses of identical flames)
    def cyclic_shift(self, src_own, shift):
        """dst[i] = src[(i - shift) mod n_global] on the GLOBAL vector whose owned block is ``src_own``; returns this rank's
        owned block of dst.  Segments are cut by the ranks' ranges; every rank computes the same list."""
        n, s = self.n_global, int(shift) % self.n_global
        dst = torch.empty_like(src_own)
        stage = self.backend == "gloo" and src_own.is_cuda
        ops, keep, staged = [], [], []

        def pieces(lo, hi):             # global [lo, hi) (no wrap) cut by owner ranges -> (owner, a, b)
            out = []
            for p, (a, b) in enumerate(self.ranges):
                u, v = max(lo, a), min(hi, b)
                if u < v:
                    out.append((p, u, v))
            return out

        def src_ranges(d0, d1):         # source index ranges (no wrap) feeding destination [d0, d1): (lo, hi, first dst index)
            lo, hi = d0 - s, d1 - s
            if hi <= 0:
                return [(lo + n, hi + n, d0)]
            if lo < 0:
                return [(lo + n, n, d0), (0, hi, d0 - lo)]
            return [(lo, hi, d0)]

        local_copies = []
        for (lo, hi, dpos) in src_ranges(self.r0, self.r1):          # what I receive, by source owner
            for (p, u, v) in pieces(lo, hi):
                d_a = dpos + (u - lo) - self.r0
                if p == self.rank:
                    local_copies.append((u - self.r0, v - self.r0, d_a))
                else:
                    view = torch.view_as_real(dst[d_a:d_a + (v - u)])
                    if stage:
                        buf = torch.empty(view.shape, dtype=view.dtype)
                        staged.append((view, buf))
                    else:
                        buf = view
                    ops.append((p, "recv", buf, u))
        for q, (q0, q1) in enumerate(self.ranges):                    # what I send: every other rank's destination range
            if q == self.rank:
                continue
            for (lo, hi, _dpos) in src_ranges(q0, q1):
                for (p, u, v) in pieces(lo, hi):
                    if p == self.rank:
                        view = torch.view_as_real(src_own[u - self.r0:v - self.r0].contiguous())
                        buf = view.cpu() if stage else view
                        keep.append(buf)
                        ops.append((q, "send", buf, u))
        for (a, b, d_a) in local_copies:
            dst[d_a:d_a + (b - a)].copy_(src_own[a:b])
        # same matching order on both sides: by peer, then by the global source offset of the segment
        p2p = [dist.P2POp(dist.isend if kind == "send" else dist.irecv, buf, peer, group=self.group)
               for peer, kind, buf, _u in sorted(ops, key=lambda o: (o[0], o[3]))]
        if p2p:
            wait_all(issue(p2p))
        for view, buf in staged:
            view.copy_(buf)
        return dst

    # ---- callbacks for hx_amg_set_sharded
    def callbacks(self):
        """(halo_fn, allreduce_fn, ctx) for the library: its own RCCL entry points with the communicator as context, or --
        gloo path -- ctypes trampolines into torch.distributed that wrap the raw device pointers."""
        if self.comm is not None:
            return (C.cast(_hx.lib.hx_comm_halo, _hx.HALO_FN), C.cast(_hx.lib.hx_comm_allreduce, _hx.ALLREDUCE_FN),
                    self.comm)
        if self._cb is None:
            self._cb = ShardedCallbacks(self)
        return self._cb.halo, self._cb.allreduce, None

    def callback_error(self):
        return None if self._cb is None else self._cb.error

    def view(self, ptr, nbytes, dtype):
        """Device buffer of the library as a torch tensor of ``dtype`` (no copy)."""
        key = (int(ptr), int(nbytes), dtype)
        t = self._views.get(key)
        if t is None:
            t = torch.as_tensor(_DevView(ptr, nbytes), device=_hx.require_gpu()).view(dtype)
            if len(self._views) > 256:
                self._views.clear()
            self._views[key] = t
        return t


class ShardedCallbacks:
    """ctypes trampolines of the gloo path (hx_halo_fn / hx_allreduce_fn): the library hands raw device pointers."""

    def __init__(self, shard):
        self.shard = shard
        self.error = None

        def halo(_ctx, level, ptr, elem_bytes, _stream):
            try:
                n_ext = shard.level_n[level][1]
                shard.halo(shard.view(ptr, n_ext * elem_bytes, _ELEM[elem_bytes]), level)
                return 0
            except Exception:                      # never let an exception cross the C boundary
                self.error = traceback.format_exc()
                return 1

        def allreduce(_ctx, ptr, count, dtype, op, _stream):
            try:
                t = shard.view(ptr, count * (8 if dtype else 4), torch.float64 if dtype else torch.float32)
                shard.all_reduce_(t, "max" if op else "sum")
                return 0
            except Exception:
                self.error = traceback.format_exc()
                return 1

        self.halo = _hx.HALO_FN(halo)
        self.allreduce = _hx.ALLREDUCE_FN(allreduce)


def distributed_case(case, group=None):
    """Partition the synthetic annulus case over the ranks: ShardContext + this rank's local mesh, tags and input fields
    (built once per case, like the single-GPU mesh upload: inputs are resident before the timed region)."""
    from .parameters_utils import Q_multiple
    d = case.get("dist")
    if d is not None:
        return d
    if case.get("generator") is not None:
        # the analytic annulus in its generator's numbering: every rank GENERATES its theta-slab (+ the one plane either side),
        # nobody holds the global mesh (synthetic.annulus_part = distribute_mesh of the global mesh, bit for bit)
        import time
        from .synthetic import annulus_part, micca_sound_speed
        t0 = time.perf_counter()
        nr, nth, nz = case["generator"]
        plane = (nr + 1) * (nz + 1)
        shard = ShardContext(nth * plane, group=group, align=plane)
        local, sub_l, ft_l = annulus_part(nr, nth, nz, shard.ranges, shard.rank, group)
        shard.set_ghosts(local.part["local_to_global"][local.part["n_own"]:])      # level-0 exchange plan = the mesh's ghost layer
        local.shard = shard
        c_l = micca_sound_speed(local)
        mine = dict(rows=int(local.part["n_own"]), ghosts=int(local.num_vertices - local.part["n_own"]),
                    neighbours=int(len({(shard.rank - 1) % shard.world, (shard.rank + 1) % shard.world} - {shard.rank})),
                    host_setup_s=round(time.perf_counter() - t0, 3))
        try:
            import resource
            mine["host_rss_gib"] = round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2 ** 20, 2)
        except Exception:
            pass
        per_rank = [None] * shard.world
        dist.all_gather_object(per_rank, mine, group=group)
        case["partition"] = dict(kind="theta-slabs of the generator's numbering, every rank generates its own slab", per_rank=per_rank)
        local.assembler()
        c_l.resident()
        h_l = Q_multiple(local, sub_l, case["P"].N_sector).resident()
        d = case["dist"] = dict(shard=shard, mesh=local, subdomains=sub_l, facet_tags=ft_l, c=c_l, h=h_l)
        case["shard"] = shard
        return d
    mesh = case["mesh"]
    x = mesh.geometry.x
    world = dist.get_world_size(group)
    sub, ft, c = case["subdomains"], case["facet_tags"], case["c"]
    if case.get("numbering", "native") == "native":
        # the generator's theta-major numbering: theta-slabs cut between mesh planes (two neighbours per rank)
        th = np.arctan2(x[:, 1], x[:, 0])
        plane = max(int(np.argmax(np.abs(th - th[0]) > 1e-12)), 1)     # rows per theta-plane
        shard = ShardContext(mesh.num_vertices, group=group, align=plane)
        kind = "theta-slabs of the generator's numbering"
    else:
        # any other mesh: recursive coordinate bisection, reverse Cuthill-McKee inside every part (what DOLFINx's graph
        # partitioner + dof reordering do for the reference, io_utils.py:149-155); HX_PARTITION=rows keeps contiguous row ranges
        # of the numbering as it is (strips of the global RCM order: fewer neighbours, 2-3 x more ghost rows)
        from .distributed import rcb_renumbering
        from .fem import Function, FunctionSpace, MeshTags
        if os.environ.get("HX_PARTITION", "rcb") == "rcb":
            perm, ranges = rcb_renumbering(mesh, world)
            old = mesh
            mesh = old.renumbered(perm)
            sub = MeshTags(mesh, sub.dim, sub.indices, sub.values)
            ft = MeshTags(mesh, ft.dim, ft.indices, ft.values)
            if c.function_space.kind != "DG0":               # nodal field: follows its vertices
                c2 = Function(FunctionSpace(mesh, "CG", 1), c.name)
                c2.x.array[perm] = c.x.array
                c = c2
            else:
                c2 = Function(FunctionSpace(mesh, "DG", 0), c.name)
                c2.x.array[:] = c.x.array
                c = c2
            shard = ShardContext(mesh.num_vertices, group=group, ranges=ranges)
            kind = "recursive coordinate bisection + reverse Cuthill-McKee inside the parts"
        else:
            shard = ShardContext(mesh.num_vertices, group=group)
            kind = "contiguous row ranges of the mesh numbering"
    from .distributed import partition_quality
    case["partition"] = dict(kind=kind, per_rank=partition_quality(mesh.topology.cells, shard.ranges))
    local, sub_l, ft_l, (c_l,) = shard.distribute(mesh, sub, ft, [c])
    local.assembler()                                  # local mesh arrays -> HBM
    c_l.resident()
    h_l = Q_multiple(local, sub_l, case["P"].N_sector).resident()
    d = case["dist"] = dict(shard=shard, mesh=local, subdomains=sub_l, facet_tags=ft_l, c=c_l, h=h_l)
    case["shard"] = shard
    return d


def bench_hot_path(case, st_type=None, nev=4, tol=1e-3, history=None):
    """bench.hot_path(..., sharded=True): one pass of the hot path on the ONE problem partitioned over the ranks of the
    default process group -- local assembly of this rank's rows, local flame vectors, collective fixed-point solve.
    Leaves the ShardContext in case["shard"], this rank's rows of the shifted operator in case["K_local"]."""
    from .acoustic_matrices import AcousticMatrices
    from .eigensolvers import fixed_point_iteration
    from .flame_matrices import PointwiseFlameMatrix
    from .petsc import combine_csr
    P = case["P"]
    d = distributed_case(case)
    mesh, shard = d["mesh"], d["shard"]
    mesh.assembler().rebuild_maps()
    matrices = AcousticMatrices(mesh, d["facet_tags"], {11: {'Robin': P.R_outlet}}, d["c"])
    D = PointwiseFlameMatrix(mesh, d["subdomains"], P.x_r(), d["h"], P.rho_amb, P.q_0, P.u_b, case["ftf"])
    D.assemble_submatrices('direct')
    E = fixed_point_iteration(matrices, D, P.target, i=0, nev=nev, tol=tol, st_type=st_type or "amg", history=history,
                              shard=shard)
    A, B, Cm = matrices.A, matrices.B, matrices.C
    case["K_local"] = combine_csr(A.shape, A.csr_terms + [(P.target * c, m) for c, m in B.csr_terms]
                                  + [(P.target ** 2 * c, m) for c, m in Cm.csr_terms])
    case["setup_mode"] = ("sharded: every rank assembles its own rows on its part of the mesh and builds its slab of the "
                          "hierarchy (uncoupled aggregation, Galerkin products from owned rows + sparse-row halos)")
    return E, matrices
