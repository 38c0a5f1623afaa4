"""Minimal stand-ins for the DOLFINx objects the hot path receives and the
device-side P1 assembly driver.

DOLFINx is not a dependency of this package: the reference uses it for meshes, P1 function spaces and
form assembly (acoustic_matrices.py:25,101-123; flame_matrices.py:24,63,141-156,199-200).  Here
  * ``Mesh`` / ``MeshTags`` / ``FunctionSpace`` / ``Function`` expose the attributes the reference
    code touches (``mesh.topology.dim``, ``mesh.geometry.x/.dim/.dofmap``, ``tags.indices/.values/.find``,
    ``f.x.array``, ``f.name``, ``f.copy()``, ``f.function_space``);
  * ``P1Assembler`` owns the device copies of the mesh, the vertex->cell adjacency and the CSR pattern,
    and launches the libhx assembly kernels (one thread per matrix row, deterministic).
"""
import os
import ctypes as C
from itertools import combinations

import numpy as np
import torch

from . import _hx
from .petsc import CSR, Mat, new_pattern_id, register_coords, to_device


# ----------------------------------------------------------------------------- host-side objects
class _Topology:
    def __init__(self, dim, cells):
        self.dim = dim
        self.cells = cells

    def cell_name(self):
        return {1: "interval", 2: "triangle", 3: "tetrahedron"}[self.dim]

    def create_connectivity(self, d0, d1):
        pass


class _Geometry:
    def __init__(self, x, dim, dofmap):
        self.x = x
        self.dim = dim
        self.dofmap = dofmap


class _Comm:
    rank, size = 0, 1

    def allreduce(self, v, op=None):
        return v


class Mesh:
    """Simplicial mesh: x (n,3) float64, cells (nc, d+1) int32; P1 dofmap == vertex ids.

    ``facets`` (nf, d) lists the tagged / exterior facets by vertex ids; a facet *index* (as found in
    ``MeshTags.indices`` of dimension d-1) is a row of that table; ``facet_cells`` is the adjacent cell."""

    def __init__(self, x, cells, facets=None, facet_cells=None):
        x = np.ascontiguousarray(x, dtype=np.float64)
        if x.shape[1] != 3:
            x = np.concatenate([x, np.zeros((x.shape[0], 3 - x.shape[1]))], axis=1)
        cells = np.ascontiguousarray(cells, dtype=np.int32)
        dim = cells.shape[1] - 1
        self.topology = _Topology(dim, cells)
        self.geometry = _Geometry(x, dim, cells)
        self.comm = _Comm()
        if facets is None:
            facets, facet_cells = exterior_facets(cells)
        self.facets = np.ascontiguousarray(facets, dtype=np.int32)
        self.facet_cells = np.ascontiguousarray(facet_cells, dtype=np.int32)
        self._assembler = None

    @property
    def num_vertices(self):
        return self.geometry.x.shape[0]

    @property
    def num_cells(self):
        return self.topology.cells.shape[0]

    def renumbered(self, perm):
        """The same mesh with vertex ``old`` renamed ``perm[old]`` (cells and tagged facets keep their order, so MeshTags
        indices carry over: build them again on the new mesh with the same indices / values).  DOLFINx renumbers the
        dofs of every mesh it reads for locality (reverse Cuthill-McKee); here the numbering of the file is kept unless
        the caller asks -- ``locality_permutation`` gives an ordering along a space-filling curve."""
        perm = np.asarray(perm, dtype=np.int64)
        x_old = self.geometry.x
        if perm.shape != (x_old.shape[0],) or not np.array_equal(np.sort(perm), np.arange(x_old.shape[0])):
            raise ValueError("perm must be a permutation of the vertex ids")
        x = np.empty_like(x_old)
        x[perm] = x_old
        return Mesh(x, perm[self.topology.cells], facets=np.sort(perm[self.facets], axis=1), facet_cells=self.facet_cells)

    def locality_renumbered(self, kind="rcm"):
        """The same mesh with its vertices renumbered for locality -- ``kind`` "rcm": reverse Cuthill-McKee of the vertex graph
        (``bandwidth_permutation``; what DOLFINx applies to the dofs of every mesh the reference reads), "morton": along a
        space-filling curve (``locality_permutation``).  ``input_vertex[new]`` of the result = the vertex id in the numbering
        this mesh was first given in (DOLFINx: ``geometry.input_global_indices``), composed over repeated renumberings, so
        nodal results can be reported in the numbering of the mesh file.  Cells and tagged facets keep their order (rebuild
        MeshTags on the new mesh with the same indices / values)."""
        if kind == "rcm":
            perm = bandwidth_permutation(self)
        elif kind == "morton":
            perm = locality_permutation(self.geometry.x)
        else:
            raise ValueError("kind must be 'rcm' or 'morton'")
        out = self.renumbered(perm)
        old = getattr(self, "input_vertex", None)
        new_to_old = np.empty(perm.size, dtype=np.int64)
        new_to_old[perm] = np.arange(perm.size)
        out.input_vertex = new_to_old if old is None else np.asarray(old)[new_to_old]
        return out

    def assembler(self):
        if self._assembler is None:
            self._assembler = P1Assembler(self)
        return self._assembler


def locality_permutation(x, bits=21):
    """perm[old] = new vertex id along the Morton (Z-order) curve of the coordinates: neighbours in space become
    neighbours in memory, which is what the SpMV gathers and the 16-bit column deltas of the SELL-64 format want
    (a randomly numbered 1 M-DOF annulus runs 2.4 times slower than a locality-ordered one).  Pure integer work on the
    host, O(n log n)."""
    x = np.asarray(x, dtype=np.float64)
    lo, hi = x.min(axis=0), x.max(axis=0)
    span = np.where(hi > lo, hi - lo, 1.0)
    q = np.minimum(((x - lo) / span * (2 ** bits - 1)).astype(np.uint64), np.uint64(2 ** bits - 1))

    def spread(v):                         # insert two zero bits between the bits of v (21 bits -> 63 bits)
        v = v & np.uint64(0x1fffff)
        v = (v | (v << np.uint64(32))) & np.uint64(0x1f00000000ffff)
        v = (v | (v << np.uint64(16))) & np.uint64(0x1f0000ff0000ff)
        v = (v | (v << np.uint64(8))) & np.uint64(0x100f00f00f00f00f)
        v = (v | (v << np.uint64(4))) & np.uint64(0x10c30c30c30c30c3)
        v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
        return v

    code = spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1)) | (spread(q[:, 2]) << np.uint64(2))
    order = np.argsort(code, kind="stable")            # order[new] = old
    perm = np.empty(x.shape[0], dtype=np.int64)
    perm[order] = np.arange(x.shape[0])
    return perm


def vertex_bandwidth(cells):
    """max |i - j| over the vertex pairs sharing a cell: the half bandwidth of every P1 operator on the mesh."""
    cells = np.asarray(cells, dtype=np.int64)
    return int((cells.max(axis=1) - cells.min(axis=1)).max()) if cells.size else 0


def bandwidth_permutation(mesh, degree_window=None):
    """perm[old] = new vertex id in the reverse Cuthill-McKee order of the vertex graph (vertices sharing a cell are adjacent).
    A mesh file numbers vertices entity by entity (gmsh: points, curves, surfaces, volumes), which scatters the neighbours
    of a row over the whole index range: a randomly numbered 5 M-DOF annulus runs the hot path five times slower than the
    same mesh in this order (EXPERIMENTS.md, round 3), because the SpMV gathers lose their locality and the column
    deltas of the SELL-64 format no longer fit 16 bits.  The graph is the P1 sparsity pattern: taken from the assembler
    when it has been built on the device, else formed on the host from the cells.  One-off integer work at ingestion
    (scipy.sparse.csgraph, O(nnz))."""
    import scipy.sparse as sp
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    n = mesh.num_vertices
    if degree_window is None:
        degree_window = int(os.environ.get("HX_RCM_DEGREE_WINDOW", "0"))
    asm = getattr(mesh, "_assembler", None)
    if asm is not None and getattr(asm, "rowptr", None) is not None:
        rp, ci = asm.rowptr.cpu().numpy(), asm.col.cpu().numpy()
        G = sp.csr_matrix((np.ones(ci.size, np.int8), ci, rp), shape=(n, n))
    else:
        cells = np.asarray(mesh.topology.cells, dtype=np.int32)
        k = cells.shape[1]
        rows = np.repeat(cells, k, axis=1).ravel()
        cols = np.tile(cells, (1, k)).ravel()
        G = sp.coo_matrix((np.ones(rows.size, np.int8), (rows, cols)), shape=(n, n)).tocsr()
        del rows, cols
    order = reverse_cuthill_mckee(G, symmetric_mode=True)          # order[new] = old
    if degree_window and degree_window > 1:
        # Experiment, off by default (HX_RCM_DEGREE_WINDOW): SELL-64 pads every slice of 64 rows to its longest row -- on an
        # unstructured mesh (8 to 30+ neighbours per vertex) a quarter more entries than the operator has.  Sorting the vertices
        # by their number of neighbours inside windows of the RCM order puts rows of equal length into the same slices (1.9 M-DOF
        # refined MICCA mesh, window 512: 33.6 M -> 28.4 M stored entries) -- but the SELL kernels are no faster for it (operator
        # product 0.133 -> 0.133 ms, smoother step 0.064 -> 0.064: the padding repeats the last column, whose x is in cache, and the
        # gathers lose some locality) and the greedy aggregation, which follows the numbering, builds a hierarchy that needs 19.1
        # instead of 18.0 iterations per solve: 2.10 -> 2.27 s per call.
        deg = np.diff(G.indptr)[order]
        win = np.arange(n) // int(degree_window)
        order = order[np.lexsort((-deg, win))]
    perm = np.empty(n, dtype=np.int64)
    perm[order] = np.arange(n)
    return perm


class MeshTags:
    """dolfinx.mesh.meshtags stand-in: sorted entity ``indices`` with ``values``."""

    def __init__(self, mesh, dim, indices, values):
        indices = np.asarray(indices, dtype=np.int32)
        values = np.asarray(values, dtype=np.int32)
        order = np.argsort(indices, kind="stable")
        self.mesh, self.dim = mesh, dim
        self.indices, self.values = indices[order], values[order]

    def find(self, tag):
        return self.indices[self.values == tag]


def exterior_facets(cells):
    """(facets (nf, d) with ascending vertex ids, adjacent cell (nf,)) of a simplicial mesh."""
    cells = np.asarray(cells)
    k = cells.shape[1]
    if k == 2:
        fl = cells.reshape(-1, 1)
        owner = np.repeat(np.arange(cells.shape[0]), 2)
    else:
        combos = list(combinations(range(k), k - 1))
        fl = np.sort(np.concatenate([cells[:, c] for c in combos], axis=0), axis=1)
        owner = np.tile(np.arange(cells.shape[0]), len(combos))
    uniq, first, counts = np.unique(fl, axis=0, return_index=True, return_counts=True)
    m = counts == 1
    return uniq[m].astype(np.int32), owner[first[m]].astype(np.int32)


TET_EDGES = np.array([(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)])
TRI_EDGES = np.array([(0, 1), (0, 2), (1, 2)])


def p2_dofmap(n_vertices, cells):
    """P2 dofs on tetrahedra: the mesh vertices first (P1 numbering), then one dof per edge, edges sorted by their
    (low, high) vertex pair.  Returns (cell_dofs (nc, 10): 4 vertices + the edges in TET_EDGES order, edges (ne, 2)).
    (DOLFINx numbers differently; eigenvalues do not depend on it.)"""
    cells = np.asarray(cells, dtype=np.int64)
    pairs = np.sort(cells[:, TET_EDGES], axis=2).reshape(-1, 2)
    key = pairs[:, 0] * n_vertices + pairs[:, 1]
    uniq, inv = np.unique(key, return_inverse=True)
    edges = np.stack([uniq // n_vertices, uniq % n_vertices], axis=1)
    cell_dofs = np.concatenate([cells, n_vertices + inv.reshape(-1, 6)], axis=1)
    return np.ascontiguousarray(cell_dofs, dtype=np.int32), np.ascontiguousarray(edges, dtype=np.int32)


def p2_dofmap_distributed(mesh):
    """P2 dofs of a DISTRIBUTED local mesh (``mesh.part``, distributed.distribute_mesh) -- what DOLFINx's dofmap builder does for
    the reference under mpirun (acoustic_matrices.py:25 with degree = 2).  An edge belongs to the rank that owns its lower GLOBAL
    vertex: every cell containing the edge touches that vertex, so the owner holds all of them and the assembled row of the
    edge dof is complete there (owner computes, no assembly communication -- as for the vertices).  Global numbering: rank after
    rank, a rank's owned vertices (ascending global vertex id) followed by its owned edges (ascending (low, high) pair); local
    numbering [owned vertices | owned edges | ghost dofs ascending global id] -- the [owned | ghost] layout of the row-sharded
    solve.  The global ids of ghost edges come from their owners: two small object collectives (the keys asked for, the positions
    answered).  Collective over ``mesh.comm``.

    Returns dict(cell_dofs (nc, 10) int32 local, edges (ne, 2) local vertex pairs, edge_keys (ne,) ascending low * n_global + high
    of GLOBAL vertex ids, vertex_dof (nv,), edge_dof (ne,), n_own, dof_l2g, dof_ranges, n_global_dofs)."""
    import torch.distributed as dist
    part = mesh.part
    N, r0, r1, n_vo, rank = int(part["n_global"]), int(part["r0"]), int(part["r1"]), int(part["n_own"]), int(part["rank"])
    l2g = np.asarray(part["local_to_global"], dtype=np.int64)
    cells = np.asarray(mesh.topology.cells, dtype=np.int64)
    pl = cells[:, TET_EDGES].reshape(-1, 2)
    pg = l2g[pl]
    swap = pg[:, 0] > pg[:, 1]
    pl[swap] = pl[swap][:, ::-1]
    pg[swap] = pg[swap][:, ::-1]
    uniq, first, inv = np.unique(pg[:, 0] * N + pg[:, 1], return_index=True, return_inverse=True)
    edges = pl[first]
    own_e = (uniq // N >= r0) & (uniq // N < r1)
    own_keys = uniq[own_e]
    n_eo, n_own = int(own_keys.size), n_vo + int(own_keys.size)
    group, world = mesh.comm.group, mesh.comm.size
    v_ranges = np.asarray(part["ranges"], dtype=np.int64).reshape(world, 2)
    if world > 1:
        counts = [None] * world
        dist.all_gather_object(counts, n_own, group=group)
    else:
        counts = [n_own]
    starts = np.concatenate([[0], np.cumsum(np.asarray(counts, dtype=np.int64))])
    gv = l2g[n_vo:]
    owner_v = np.searchsorted(v_ranges[:, 1], gv, side="right")
    gv_dof = starts[owner_v] + (gv - v_ranges[owner_v, 0])
    ghost_keys = uniq[~own_e]
    owner_e = np.searchsorted(v_ranges[:, 1], ghost_keys // N, side="right")
    ge_dof = np.empty(ghost_keys.size, dtype=np.int64)
    if world > 1:
        asked = [None] * world
        dist.all_gather_object(asked, {int(q): ghost_keys[owner_e == q] for q in np.unique(owner_e)}, group=group)
        mine = {}
        for r, req in enumerate(asked):
            if rank in req:
                pos = np.searchsorted(own_keys, req[rank])
                if not np.array_equal(own_keys[np.minimum(pos, max(n_eo - 1, 0))], req[rank]):
                    raise ValueError("rank %d asked rank %d for an edge it does not hold" % (r, rank))
                mine[r] = pos
        answered = [None] * world
        dist.all_gather_object(answered, mine, group=group)
        for q in np.unique(owner_e):
            q = int(q)
            ge_dof[owner_e == q] = starts[q] + (v_ranges[q, 1] - v_ranges[q, 0]) + answered[q][rank]
    elif ghost_keys.size:
        raise ValueError("ghost edges on a one-rank mesh")
    ghost_global = np.concatenate([gv_dof, ge_dof])
    order = np.argsort(ghost_global, kind="stable")
    ghost_local = np.empty(ghost_global.size, dtype=np.int64)
    ghost_local[order] = n_own + np.arange(ghost_global.size)
    vertex_dof = np.concatenate([np.arange(n_vo, dtype=np.int64), ghost_local[:gv.size]])
    edge_dof = np.empty(uniq.size, dtype=np.int64)
    edge_dof[own_e] = n_vo + np.arange(n_eo)
    edge_dof[~own_e] = ghost_local[gv.size:]
    cell_dofs = np.concatenate([vertex_dof[cells], edge_dof[inv].reshape(-1, 6)], axis=1)
    return dict(cell_dofs=np.ascontiguousarray(cell_dofs, dtype=np.int32), edges=np.ascontiguousarray(edges, dtype=np.int32),
                edge_keys=uniq, vertex_dof=vertex_dof, edge_dof=edge_dof, n_own=n_own,
                dof_l2g=np.concatenate([starts[rank] + np.arange(n_own), ghost_global[order]]),
                dof_ranges=[(int(starts[p]), int(starts[p + 1])) for p in range(world)], n_global_dofs=int(starts[-1]))


def collapsed_gauss(dim, n=4):
    """Barycentric quadrature table (nq, dim + 2): points + weight (sum 1) on the reference simplex -- tensor
    Gauss-Legendre in collapsed coordinates, exact to total degree 2n - 1 - (dim - 1) (n = 4: 5 on tetrahedra, 6 on
    triangles: every polynomial form of the degree-2 operators)."""
    g, w = np.polynomial.legendre.leggauss(n)
    g, w = 0.5 * (g + 1), 0.5 * w
    rows = []
    if dim == 2:
        for a, wa in zip(g, w):
            for b, wb in zip(g, w):
                x1, x2 = a, b * (1 - a)
                rows.append([1 - x1 - x2, x1, x2, wa * wb * (1 - a)])
    else:
        for a, wa in zip(g, w):
            for b, wb in zip(g, w):
                for c, wc in zip(g, w):
                    x1, x2, x3 = a, b * (1 - a), c * (1 - a) * (1 - b)
                    rows.append([1 - x1 - x2 - x3, x1, x2, x3, wa * wb * wc * (1 - a) ** 2 * (1 - b)])
    t = np.array(rows)
    t[:, -1] /= t[:, -1].sum()
    return np.ascontiguousarray(t)


class FunctionSpace:
    def __init__(self, mesh, family="Lagrange", degree=1):
        fam = family.upper()
        if fam in ("LAGRANGE", "CG", "P") and degree == 1:
            self.kind, self.size = "P1", mesh.num_vertices
        elif fam in ("DG", "DISCONTINUOUS LAGRANGE") and degree == 0:
            self.kind, self.size = "DG0", mesh.num_cells
        elif fam in ("LAGRANGE", "CG", "P") and degree == 2:
            if mesh.topology.dim != 3:
                raise NotImplementedError("quadratic elements are provided on tetrahedra (the reference's degree-2 case)")
            self.kind = "P2"
            nv = mesh.num_vertices
            if getattr(mesh, "part", None) is not None:
                # distributed local mesh: one numbering (and one exchange plan) per mesh, shared by every quadratic space on it
                d = mesh.__dict__.get("_p2_dist")
                if d is None:
                    d = mesh.__dict__["_p2_dist"] = p2_dofmap_distributed(mesh)
                    # the row-sharding context of the quadratic dofs is made HERE, where every rank is (the numbering above is
                    # collective too) -- not on first use, which need not happen on all ranks at the same point
                    from .sharded import ShardContext
                    d["shard"] = ShardContext(d["n_global_dofs"], group=mesh.comm.group, ranges=d["dof_ranges"])
                    d["shard"].set_ghosts(d["dof_l2g"][d["n_own"]:])
                self._dist = d
                self.cell_dofs, self.edges, self.edge_keys = d["cell_dofs"], d["edges"], d["edge_keys"]
                self.vertex_dof, self.edge_dof, self.n_own = d["vertex_dof"], d["edge_dof"], d["n_own"]
                self.size = nv + self.edges.shape[0]
            else:
                self.cell_dofs, self.edges = p2_dofmap(nv, mesh.topology.cells)
                self.edge_keys = self.edges[:, 0].astype(np.int64) * nv + self.edges[:, 1]
                self.vertex_dof, self.edge_dof = np.arange(nv, dtype=np.int64), nv + np.arange(self.edges.shape[0], dtype=np.int64)
                self.size = self.n_own = nv + self.edges.shape[0]
        else:
            raise NotImplementedError("P1, P2 (tetrahedra) and DG0 spaces are on the hot path (got %s %d)" % (family, degree))
        self.mesh = mesh
        if self.kind != "P2":
            part = getattr(mesh, "part", None)
            self.n_own = part["n_own"] if (part is not None and self.kind == "P1") else self.size
        if self.kind == "P1":
            self.dofmap = mesh.topology.cells
        elif self.kind == "P2":
            self.dofmap = self.cell_dofs
        else:
            self.dofmap = np.arange(mesh.num_cells, dtype=np.int32)[:, None]

    def assembler(self):
        """Device assembler of this space (P1: the mesh's; P2: cached on the mesh per space kind)."""
        if self.kind == "P2":
            cache = self.mesh.__dict__.setdefault("_p2_assembler", None)
            if cache is None:
                cache = self.mesh.__dict__["_p2_assembler"] = P2Assembler(self)
            return cache
        return self.mesh.assembler()

    @property
    def shard(self):
        """Row-sharding context of this space on a distributed local mesh (``fixed_point_iteration(..., shard=V.shard)``): the
        mesh's own for P1 (ShardContext.distribute), one over the quadratic dofs for P2 (made with the mesh's first quadratic
        space)."""
        d = getattr(self, "_dist", None)
        return getattr(self.mesh, "shard", None) if d is None else d["shard"]

    def tabulate_dof_coordinates(self):
        if self.kind == "P1":
            return self.mesh.geometry.x
        if self.kind == "P2":
            x = self.mesh.geometry.x
            out = np.empty((self.size, x.shape[1]), dtype=x.dtype)
            out[self.vertex_dof] = x
            out[self.edge_dof] = 0.5 * (x[self.edges[:, 0]] + x[self.edges[:, 1]])
            return out
        return self.mesh.geometry.x[self.mesh.topology.cells].mean(axis=1)


def functionspace(mesh, element):
    return FunctionSpace(mesh, element[0], element[1])


class _X:
    """``Function.x``: the dof array (host numpy complex128, like a complex DOLFINx build)."""

    def __init__(self, array, space=None):
        self.array = array
        self._space = space

    def scatter_forward(self):
        pass

    @property
    def petsc_vec(self):
        from .petsc import Vec
        v = Vec(self.array)
        # on a distributed local mesh the vector is row-sharded: dots are sums over the ranks' owned entries
        v.shard = getattr(self._space, "shard", None)
        return v


class Function:
    def __init__(self, V, name="f"):
        self.function_space = V
        self.x = _X(np.zeros(V.size, dtype=np.complex128), V)
        self.name = name

    def copy(self):
        g = Function(self.function_space, self.name)
        g.x.array[:] = self.x.array
        return g

    def interpolate(self, fn):
        pts = self.function_space.tabulate_dof_coordinates()
        was_resident = getattr(self, "_dev", None) is not None
        self.release()
        self.x.array[:] = fn(pts.T)
        if was_resident:
            self.resident()

    def resident(self):
        """Keep a copy of the values in HBM for the assembly kernels (an input field that stays the same over several
        assemblies is then uploaded once).  The host array is FROZEN while the device copy exists (numpy raises on
        ``f.x.array[:] = ...``), so the copy can never go stale silently: call ``release()`` before changing the values
        (``interpolate`` does both by itself).  Returns self."""
        self.x.array.flags.writeable = True
        self._dev = to_device(self.x.array.astype(np.complex128))
        self.x.array.flags.writeable = False
        return self

    def release(self):
        """Drop the device copy made by ``resident()`` and make the host array writable again.  Returns self."""
        self._dev = None
        self.x.array.flags.writeable = True
        return self


# ----------------------------------------------------------------------------- device assembly
class P1Assembler:
    """Device mesh + adjacency + CSR pattern; wraps the hx_assemble_p1_* kernels."""

    def __init__(self, mesh):
        dev = _hx.require_gpu()
        self.mesh = mesh
        self.dim = mesh.topology.dim
        self.n = mesh.num_vertices
        self.nc = mesh.num_cells
        self.x = to_device(mesh.geometry.x)
        self.cells = to_device(mesh.topology.cells)
        self.v2c_ptr, self.v2c_idx = self._adjacency(self.cells, self.nc, self.dim + 1)
        self.rowptr, self.col = self._pattern(self.cells, self.nc, self.dim + 1, self.v2c_ptr, self.v2c_idx)
        self.pattern_id = new_pattern_id()
        if getattr(mesh, "part", None) is None:
            register_coords(self.pattern_id, mesh.geometry.x)
        self._h_pattern = None
        self.device = dev

    def _adjacency(self, ents, n_ent, k):
        ptr = torch.empty(self.n + 1, dtype=torch.int32, device=self.x.device)
        idx = torch.empty(max(n_ent * k, 1), dtype=torch.int32, device=self.x.device)
        _hx.check(_hx.lib.hx_build_v2e(self.n, n_ent, k, _hx.ptr(ents), _hx.ptr(ptr), _hx.ptr(idx), _hx.stream()))
        return ptr, idx

    def _pattern(self, ents, n_ent, k, ptr, idx):
        rowptr = torch.empty(self.n + 1, dtype=torch.int32, device=self.x.device)
        scratch = torch.empty(max(n_ent * k * k, 1), dtype=torch.int32, device=self.x.device)
        nnz = C.c_int64(0)
        _hx.check(_hx.lib.hx_p1_pattern_count(self.n, n_ent, k, _hx.ptr(ents), _hx.ptr(ptr), _hx.ptr(idx),
                                              _hx.ptr(rowptr), _hx.ptr(scratch), C.byref(nnz), _hx.stream()))
        col = torch.empty(max(nnz.value, 1), dtype=torch.int32, device=self.x.device)[:nnz.value]
        _hx.check(_hx.lib.hx_p1_pattern_fill(self.n, k, _hx.ptr(ptr), _hx.ptr(rowptr), _hx.ptr(scratch),
                                             _hx.ptr(col), _hx.stream()))
        return rowptr, col

    def _csr(self, val):
        out = CSR((self.n, self.n), self.rowptr, self.col, val, pattern_id=self.pattern_id)
        out.symmetric = True            # Galerkin P1 operators are (complex) symmetric: transpose() is the identity
        return self._own_rows(out)

    def _own_rows(self, csr):
        """Distributed local mesh (``mesh.part``: vertices numbered [owned | ghost]): the assembled rows of the OWNED
        vertices are complete (every cell touching them is local) -- keep those, over [owned | ghost] columns.  The ghost
        rows (incomplete) are dropped: the owner computes them (acoustic_matrices.py:101-103 ships such rows instead)."""
        part = getattr(self.mesh, "part", None)
        if part is None:
            return csr
        n_own = part["n_own"]
        key = ("own", csr.pattern_id)
        cache = self.__dict__.setdefault("_own_cache", {})
        if key not in cache:
            nnz = int(csr.rowptr[n_own].item())
            cache.clear()
            cache[key] = (csr.rowptr[:n_own + 1].contiguous(), nnz)
        rp, nnz = cache[key]
        out = CSR((n_own, csr.shape[1]), rp, csr.col[:nnz], None if csr.val is None else csr.val[:nnz],
                  pattern_id=("own", csr.pattern_id))
        out.symmetric = getattr(csr, "symmetric", False)
        out.local = True
        return out

    @staticmethod
    def _field(f):
        """Function -> (device complex tensor, is_cellwise) for a coefficient; a pair already in that form passes through
        (a caller that assembles several operators with one coefficient uploads it once)."""
        if isinstance(f, tuple) and len(f) == 2 and isinstance(f[0], torch.Tensor):
            return f
        if isinstance(f, Function):
            dev = getattr(f, "_dev", None)                      # Function.resident(): already in HBM
            if dev is None or dev.numel() != f.x.array.size:
                dev = to_device(f.x.array.astype(np.complex128))
            return dev, f.function_space.kind == "DG0"
        raise TypeError("coefficient must be a Function")

    # ---- operators
    def stiffness_mass(self, c=None, want_A=True, want_C=True):
        """(A, C) as Mats: A = -int c^2 grad.grad, C = mass (acoustic_matrices.py:101,121)."""
        A_val = torch.empty(self.col.numel(), dtype=torch.complex128, device=self.x.device) if want_A else None
        C_val = torch.empty(self.col.numel(), dtype=torch.complex128, device=self.x.device) if want_C else None
        cf, cellwise = (self._field(c) if c is not None else (None, False))
        _hx.check(_hx.lib.hx_assemble_p1_stiff_mass(self.dim, self.n, _hx.ptr(self.x), _hx.ptr(self.cells),
                                                    _hx.ptr(self.v2c_ptr), _hx.ptr(self.v2c_idx),
                                                    _hx.ptr(self.rowptr), _hx.ptr(self.col), _hx.ptr(cf),
                                                    int(cellwise), _hx.ptr(A_val), _hx.ptr(C_val), _hx.stream()))
        return (Mat.from_csr(self._csr(A_val)) if want_A else None,
                Mat.from_csr(self._csr(C_val)) if want_C else None)

    def apply_dirichlet(self, mat, dofs, diag=1.0):
        """DOLFINx ``assemble_matrix(bcs=...)``: rows and columns of the listed dofs zeroed, ``diag`` on their diagonal.
        Distributed mesh (``mesh.part``): the operator holds this rank's rows over [owned | ghost] columns; an owned
        Dirichlet vertex is always known here (every tagged facet with an owned vertex is local), a ghost one may lie on
        facets this rank does not hold -- its flag comes from its owner by one halo exchange (PETSc: the ghost update
        ``MatZeroRowsColumns`` does for the reference)."""
        csr = mat.csr_part()
        n_rows, n_cols = csr.shape
        flags = np.zeros(n_cols, dtype=np.uint8)
        flags[np.asarray(dofs, dtype=np.int64)] = 1
        part = getattr(self.mesh, "part", None)
        if part is not None:
            f = torch.zeros(n_cols, dtype=torch.int32, device=csr.val.device)
            f[:part["n_own"]] = to_device(flags[:part["n_own"]].astype(np.int32))
            self.mesh.shard.halo(f)
            d_flags = f.to(torch.uint8)
        else:
            d_flags = to_device(flags)
        _hx.check(_hx.lib.hx_csr_apply_dirichlet(n_rows, _hx.ptr(csr.rowptr), _hx.ptr(csr.col), _hx.ptr(csr.val),
                                                 _hx.ptr(d_flags), float(diag), _hx.stream()))

    def robin(self, facet_ids, coef_per_facet, c):
        """B = sum_f coef_f int_F c phi_k phi_j ds over the listed facets (acoustic_matrices.py:71,108-110).
        Returns a Mat on B's own (boundary-only) pattern."""
        facet_ids = np.asarray(facet_ids, dtype=np.int64)
        if facet_ids.size == 0:
            # no such facet here (a rank of a distributed mesh whose part does not touch the boundary): an operator without
            # entries -- the Mat still exists, so that `if not B` chooses the same branch on every rank
            part = getattr(self.mesh, "part", None)
            rows = part["n_own"] if part is not None else self.n
            dev = self.x.device
            out = CSR((rows, self.n), torch.zeros(rows + 1, dtype=torch.int32, device=dev),
                      torch.zeros(0, dtype=torch.int32, device=dev), torch.zeros(0, dtype=torch.complex128, device=dev))
            out.symmetric, out.local = True, part is not None
            return Mat.from_csr(out)
        facets = to_device(self.mesh.facets[facet_ids])
        nf, k = facets.shape
        fcell = to_device(self.mesh.facet_cells[facet_ids])
        coef = to_device(np.asarray(coef_per_facet, dtype=np.complex128))
        ptr, idx = self._adjacency(facets, nf, k)
        rowptr, col = self._pattern(facets, nf, k, ptr, idx)
        cf, cellwise = self._field(c)
        val = torch.empty(col.numel(), dtype=torch.complex128, device=self.x.device)
        _hx.check(_hx.lib.hx_assemble_p1_robin(self.dim, self.n, _hx.ptr(self.x), _hx.ptr(facets), _hx.ptr(ptr),
                                               _hx.ptr(idx), _hx.ptr(rowptr), _hx.ptr(col), _hx.ptr(cf),
                                               int(cellwise), _hx.ptr(fcell), _hx.ptr(coef), _hx.ptr(val),
                                               _hx.stream()))
        out = CSR((self.n, self.n), rowptr, col, val)
        out.symmetric = True            # boundary mass matrix: B^T = B, so B_adj = B^H is the conjugate (no host transpose)
        part = getattr(self.mesh, "part", None)
        if part is not None:            # distributed local mesh: owned rows only
            n_own = part["n_own"]
            nnz = int(rowptr[n_own].item())
            own = CSR((n_own, self.n), rowptr[:n_own + 1].contiguous(), col[:nnz], val[:nnz])
            own.symmetric, own.local = True, True
            return Mat.from_csr(own)
        return Mat.from_csr(out)

    # ---- load vectors (device tensors of length n)
    def load(self, kind, fa, fb=None, scale=1.0, cell_tag=None, tag=0):
        """cell_tag: optional int32 device tensor (n_cells); only cells with cell_tag == tag contribute."""
        fa_t = fa if isinstance(fa, torch.Tensor) else to_device(np.asarray(fa, dtype=np.complex128))
        fb_t = None if fb is None else (fb if isinstance(fb, torch.Tensor) else to_device(np.asarray(fb, np.complex128)))
        out = torch.empty(self.n, dtype=torch.complex128, device=self.x.device)
        _hx.check(_hx.lib.hx_assemble_p1_load(kind, self.dim, self.n, _hx.ptr(self.x), _hx.ptr(self.cells),
                                              _hx.ptr(self.v2c_ptr), _hx.ptr(self.v2c_idx), _hx.ptr(fa_t),
                                              _hx.ptr(fb_t), _hx.z2(scale), _hx.ptr(cell_tag), int(tag), _hx.ptr(out),
                                              _hx.stream()))
        return out

    def load_rows(self, kind, rows, fa, fb=None, scale=1.0, cell_tag=None, row_tag=None):
        """Entries ``rows`` (int32 array) of the load vector of ``load``; with ``cell_tag`` only the cells tagged row_tag[t]
        contribute to entry t.  Returns a device tensor of len(rows) values."""
        rows_t = to_device(np.ascontiguousarray(rows, dtype=np.int32))
        tag_t = None if row_tag is None else to_device(np.ascontiguousarray(row_tag, dtype=np.int32))
        fa_t = fa if isinstance(fa, torch.Tensor) else to_device(np.asarray(fa, dtype=np.complex128))
        fb_t = None if fb is None else (fb if isinstance(fb, torch.Tensor) else to_device(np.asarray(fb, np.complex128)))
        out = torch.empty(max(rows_t.numel(), 1), dtype=torch.complex128, device=self.x.device)[:rows_t.numel()]
        _hx.check(_hx.lib.hx_assemble_p1_load_rows(kind, self.dim, rows_t.numel(), _hx.ptr(rows_t), _hx.ptr(tag_t),
                                                   _hx.ptr(self.x), _hx.ptr(self.cells), _hx.ptr(self.v2c_ptr),
                                                   _hx.ptr(self.v2c_idx), _hx.ptr(fa_t), _hx.ptr(fb_t), _hx.z2(scale),
                                                   _hx.ptr(cell_tag), _hx.ptr(out), _hx.stream()))
        return out

    def cell_tag_array(self, tags):
        """Dense int32 device array of cell tags (-1 = untagged) from a MeshTags, cached per MeshTags object."""
        cache = self.__dict__.setdefault("_tag_cache", {})
        if id(tags) not in cache:
            dense = np.full(self.nc, -1, dtype=np.int32)
            dense[tags.indices] = tags.values
            cache[id(tags)] = (tags, to_device(dense))
        return cache[id(tags)][1]

    def rebuild_maps(self):
        """Recompute adjacency and CSR pattern from the resident mesh arrays (what a fresh assembly pays)."""
        self.v2c_ptr, self.v2c_idx = self._adjacency(self.cells, self.nc, self.dim + 1)
        self.rowptr, self.col = self._pattern(self.cells, self.nc, self.dim + 1, self.v2c_ptr, self.v2c_idx)
        self.pattern_id = new_pattern_id()
        self._h_pattern = None
        if getattr(self.mesh, "part", None) is None:
            register_coords(self.pattern_id, self.mesh.geometry.x)

    def lumped(self):
        """m_j = int phi_j dx, so that int f dx = m . f for nodal P1 f."""
        if not hasattr(self, "_lumped"):
            self._lumped = self.load(0, np.ones(self.nc, dtype=np.complex128))
        return self._lumped

    def integrate_p1(self, f_array):
        """int f dx for a nodal array (dolfinx_utils.py:42-43 ``normalize``)."""
        from .petsc import panel_dot
        f = to_device(np.asarray(f_array, np.complex128))
        part = getattr(self.mesh, "part", None)
        if part is not None:            # distributed mesh: the owned vertices' lumped weights are complete; ranks add up
            n_own = part["n_own"]
            local = panel_dot(self.lumped()[:n_own].reshape(1, -1), 1, f[:n_own].contiguous())[0]
            return complex(self.mesh.comm.allreduce(complex(local)))
        return complex(panel_dot(self.lumped().reshape(1, -1), 1, f)[0])   # lumped weights are real: m^H f = m.f

    def cell_volumes(self):
        """|K_c| (device kernel, cached; returned as a host array for the per-flame volume sums)."""
        if not hasattr(self, "_vol"):
            vol = torch.empty(self.nc, dtype=torch.float64, device=self.x.device)
            _hx.check(_hx.lib.hx_p1_cell_volumes(self.dim, self.nc, _hx.ptr(self.x), _hx.ptr(self.cells), _hx.ptr(vol),
                                                 _hx.stream()))
            self._vol = vol.cpu().numpy()
        return self._vol

    def locate_points(self, points, tol=1e-10, must_find=True):
        """Lowest-index cell containing each point (n_pts x 3) -> int32 host array (raises if a point is outside; with
        ``must_find=False`` -- distributed local meshes -- such points get -1 instead)."""
        pts = np.ascontiguousarray(np.asarray(points, dtype=np.float64).reshape(-1, 3))
        d_pts = to_device(pts)
        out = torch.full((pts.shape[0],), 2 ** 31 - 1, dtype=torch.int32, device=self.x.device)
        _hx.check(_hx.lib.hx_p1_locate_points(self.dim, self.nc, _hx.ptr(self.x), _hx.ptr(self.cells), pts.shape[0],
                                              _hx.ptr(d_pts), float(tol), _hx.ptr(out), _hx.stream()))
        cells = out.cpu().numpy()
        if (cells == 2 ** 31 - 1).any():
            if not must_find:
                return np.where(cells == 2 ** 31 - 1, -1, cells)
            raise ValueError("reference point(s) %s outside the mesh" % (pts[cells == 2 ** 31 - 1],))
        return cells


class P2Assembler:
    """Quadratic Lagrange space on tetrahedra with P1 coefficient fields: device dof map, dof -> cell adjacency, CSR
    pattern and the hx_assemble_p2_* kernels (the reference's ``degree = 2``: acoustic_matrices.py:25,
    flame_matrices.py:199-200).  Same owner-computes scheme as P1Assembler."""

    def __init__(self, V):
        assert V.kind == "P2"
        self.V, self.mesh = V, V.mesh
        base = self.mesh.assembler()
        self.x, self.cells = base.x, base.cells
        self.n, self.nc = V.size, self.mesh.num_cells
        self.cdofs = to_device(V.cell_dofs)
        self.quad3 = to_device(collapsed_gauss(3))
        self.quad2 = to_device(collapsed_gauss(2))
        self.d2c_ptr, self.d2c_idx = self._adjacency(self.cdofs, self.nc, 10)
        self.rowptr, self.col = self._pattern(self.cdofs, self.nc, 10, self.d2c_ptr, self.d2c_idx)
        self.pattern_id = new_pattern_id()
        if getattr(self.mesh, "part", None) is None and hasattr(V, "tabulate_dof_coordinates"):
            register_coords(self.pattern_id, V.tabulate_dof_coordinates())

    _adjacency = P1Assembler._adjacency
    _pattern = P1Assembler._pattern

    def _csr(self, val):
        out = CSR((self.n, self.n), self.rowptr, self.col, val, pattern_id=self.pattern_id)
        out.symmetric = True
        return self._own_rows(out)

    def _own_rows(self, csr):
        """Distributed local mesh: the rows of the OWNED dofs (vertices and edges, complete: ``p2_dofmap_distributed``) over
        [owned | ghost] columns, as P1Assembler._own_rows."""
        if getattr(self.mesh, "part", None) is None:
            return csr
        n_own = self.V.n_own
        nnz = int(csr.rowptr[n_own].item())
        out = CSR((n_own, csr.shape[1]), csr.rowptr[:n_own + 1].contiguous(), csr.col[:nnz],
                  None if csr.val is None else csr.val[:nnz], pattern_id=("own", csr.pattern_id))
        out.symmetric = getattr(csr, "symmetric", False)
        out.local = True
        return out

    @staticmethod
    def _p1_field(f):
        if not (isinstance(f, Function) and f.function_space.kind == "P1"):
            raise NotImplementedError("coefficient fields of the quadratic space must be P1 Functions")
        return to_device(f.x.array.astype(np.complex128))

    def stiffness_mass(self, c=None, want_A=True, want_C=True):
        A_val = torch.empty(self.col.numel(), dtype=torch.complex128, device=self.x.device) if want_A else None
        C_val = torch.empty(self.col.numel(), dtype=torch.complex128, device=self.x.device) if want_C else None
        cf = self._p1_field(c) if c is not None else None
        _hx.check(_hx.lib.hx_assemble_p2_stiff_mass(self.n, _hx.ptr(self.x), _hx.ptr(self.cells), _hx.ptr(self.cdofs),
                                                    _hx.ptr(self.d2c_ptr), _hx.ptr(self.d2c_idx), _hx.ptr(self.rowptr),
                                                    _hx.ptr(self.col), _hx.ptr(cf), self.quad3.shape[0],
                                                    _hx.ptr(self.quad3), _hx.ptr(A_val), _hx.ptr(C_val), _hx.stream()))
        return (Mat.from_csr(self._csr(A_val)) if want_A else None,
                Mat.from_csr(self._csr(C_val)) if want_C else None)

    def apply_dirichlet(self, mat, dofs, diag=1.0):
        """DOLFINx ``assemble_matrix(bcs=...)`` for the quadratic space: rows and columns of the listed dofs (vertices and edge
        midpoints of the boundary facets: ``facet_dofs``) zeroed, ``diag`` on their diagonal."""
        csr = mat.csr_part()
        n_rows = csr.shape[0]
        flags = np.zeros(self.n, dtype=np.uint8)
        flags[np.asarray(dofs, dtype=np.int64)] = 1
        if getattr(self.mesh, "part", None) is not None:
            # distributed: an owned boundary dof is always known here (a tagged facet holding an owned vertex -- or an owned edge,
            # whose lower vertex is owned -- is local); the flags of the ghost dofs come from their owners by one exchange
            f = torch.zeros(self.n, dtype=torch.int32, device=csr.val.device)
            f[:n_rows] = to_device(flags[:n_rows].astype(np.int32))
            self.V.shard.halo(f)
            flags_d = f.to(torch.uint8)
        else:
            flags_d = to_device(flags)          # (bound to a name: the launch below is asynchronous)
        _hx.check(_hx.lib.hx_csr_apply_dirichlet(n_rows, _hx.ptr(csr.rowptr), _hx.ptr(csr.col), _hx.ptr(csr.val),
                                                 _hx.ptr(flags_d), float(diag), _hx.stream()))

    def facet_dofs(self, facets):
        """(nf, 6) P2 dofs of boundary triangles: 3 vertices + the edges in TRI_EDGES order."""
        V = self.V
        facets = np.asarray(facets, dtype=np.int64)
        part = getattr(self.mesh, "part", None)
        if part is None:
            g, N = facets, self.mesh.num_vertices
        else:                                  # edge keys are pairs of GLOBAL vertex ids there
            g, N = np.asarray(part["local_to_global"], dtype=np.int64)[facets], int(part["n_global"])
        pairs = np.sort(g[:, TRI_EDGES], axis=2)
        key = pairs[..., 0] * N + pairs[..., 1]
        pos = np.searchsorted(V.edge_keys, key)
        if not np.array_equal(V.edge_keys[np.minimum(pos, len(V.edge_keys) - 1)], key):
            raise ValueError("a boundary facet edge is not a mesh edge")
        return np.ascontiguousarray(np.concatenate([V.vertex_dof[facets], V.edge_dof[pos]], axis=1), dtype=np.int32)

    def robin(self, facet_ids, coef_per_facet, c):
        facet_ids = np.asarray(facet_ids, dtype=np.int64)
        part = getattr(self.mesh, "part", None)
        if facet_ids.size == 0:               # (a part that does not touch the boundary: an operator without entries, as for P1)
            rows = self.V.n_own if part is not None else self.n
            dev = self.x.device
            out = CSR((rows, self.n), torch.zeros(rows + 1, dtype=torch.int32, device=dev),
                      torch.zeros(0, dtype=torch.int32, device=dev), torch.zeros(0, dtype=torch.complex128, device=dev))
            out.symmetric, out.local = True, part is not None
            return Mat.from_csr(out)
        fv = self.mesh.facets[facet_ids]
        facets = to_device(np.ascontiguousarray(fv, dtype=np.int32))
        fdofs = to_device(self.facet_dofs(fv))
        nf = fv.shape[0]
        coef = to_device(np.asarray(coef_per_facet, dtype=np.complex128))
        ptr, idx = self._adjacency(fdofs, nf, 6)
        rowptr, col = self._pattern(fdofs, nf, 6, ptr, idx)
        val = torch.empty(col.numel(), dtype=torch.complex128, device=self.x.device)
        cf = self._p1_field(c)
        _hx.check(_hx.lib.hx_assemble_p2_robin(self.n, _hx.ptr(self.x), _hx.ptr(facets), _hx.ptr(fdofs), _hx.ptr(ptr),
                                               _hx.ptr(idx), _hx.ptr(rowptr), _hx.ptr(col), _hx.ptr(cf), _hx.ptr(coef),
                                               self.quad2.shape[0], _hx.ptr(self.quad2), _hx.ptr(val), _hx.stream()))
        out = CSR((self.n, self.n), rowptr, col, val)
        out.symmetric = True            # boundary mass matrix
        return Mat.from_csr(self._own_rows(out))

    def load_cells(self, h_cell, cell_tag, tag, scale=1.0):
        """scale * sum over the cells tagged ``tag`` of h_cell int phi_j dx (h: DG0 array; cell_tag: device int32 per cell)."""
        h_t = h_cell if isinstance(h_cell, torch.Tensor) else to_device(np.asarray(h_cell, dtype=np.complex128))
        out = torch.empty(self.n, dtype=torch.complex128, device=self.x.device)
        _hx.check(_hx.lib.hx_assemble_p2_load_cells(self.n, _hx.ptr(self.x), _hx.ptr(self.cells), _hx.ptr(self.cdofs),
                                                    _hx.ptr(self.d2c_ptr), _hx.ptr(self.d2c_idx), _hx.ptr(h_t), _hx.ptr(cell_tag),
                                                    int(tag), _hx.z2(scale), self.quad3.shape[0], _hx.ptr(self.quad3),
                                                    _hx.ptr(out), _hx.stream()))
        return out

    def load(self, kind, fa, fb, scale=1.0):
        """kind 1: scale * int fa fb phi_j dx; kind 2: scale * int (d phi_j / dz) fb / fa dx (fa, fb: P1 nodal arrays)."""
        fa_t = to_device(np.asarray(fa, dtype=np.complex128))
        fb_t = to_device(np.asarray(fb, dtype=np.complex128))
        out = torch.empty(self.n, dtype=torch.complex128, device=self.x.device)
        _hx.check(_hx.lib.hx_assemble_p2_load(kind, self.n, _hx.ptr(self.x), _hx.ptr(self.cells), _hx.ptr(self.cdofs),
                                              _hx.ptr(self.d2c_ptr), _hx.ptr(self.d2c_idx), _hx.ptr(fa_t), _hx.ptr(fb_t),
                                              _hx.z2(scale), self.quad3.shape[0], _hx.ptr(self.quad3), _hx.ptr(out),
                                              _hx.stream()))
        return out
