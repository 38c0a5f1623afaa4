"""Bloch-periodic reduction of a 1/N sector (reference: helmholtz_x/bloch_operator.py:6-123).

Same class / constructor / properties (``A, B, B_adj, C, D, remapper, blochify``).  The reduction
``M_bloch = NB . M . BN`` is a one-shot sparse triple product.  The reference forms it with scipy on the host
(bloch_operator.py:104-111); here it is two device sparse products ``NB (M BN)`` by the hash-based SpGEMM of the
hierarchy set-up (hx_spgemm_*, csrc/amg_setup.hip): M never leaves HBM (round 2 copied every operator to the host and
back: 1.2 s of a 3.2 s sector run at 1.9 M DOF).  ``HX_BLOCH_HOST=1`` keeps the scipy product (tests compare the two).

Differences, on purpose:
  * master/slave DOFs are paired GEOMETRICALLY (rotate the master face by 2 pi / N about the z axis) by default;
    the reference pairs them by sorted DOF index (:40), which is only right if the numbering happens to line up
    (on its own MICCA sector mesh it does not: paired nodes end up to 0.069 m apart).  ``pairing='index'`` reproduces
    the reference rule, in the numbering given by ``index_order`` (tests pin the reference's Bloch logs this way).
  * the rank-r flame operator stays factored: blochify() of a low-rank ``Mat`` transforms the factors
    (u' = NB u, v' = BN^T v) instead of materialising a (x) b.
  * ``B_adj`` is provided (the reference leaves it None, :25,88-90, so its Bloch + Robin adjoint cannot run).

Row-sharded runs (round 4; BASELINE config 5 names four GPUs; the reference runs its Bloch case serially, Micca/README.md:29):
``Blochifier(geometry, ..., global_geometry=G)`` with ``geometry`` holding this rank's part of the sector mesh
(``ShardContext.distribute``) and ``G`` the whole sector mesh + tags that every rank holds as host arrays anyway.  The sector
must be partitioned so that every master/slave pair is RANK-LOCAL -- slabs in z or r of a numbering that is contiguous in that
direction (``sector_slab_numbering``), not theta-slabs.  The master/slave pairing is computed on the global arrays (integer host
work, identical on every rank); every rank then forms ITS rows of NB M BN from its own rows of M by the same two device sparse
products, over the [owned | ghost] columns of the REDUCED numbering (a ghost master column folds onto its slave partner, which
joins the ghost set), and ``self.shard`` is the ShardContext of the reduced problem (what ``fixed_point_iteration(..., shard=)``
takes).  The reduced operators are not complex symmetric, so the inner solver is the row-sharded flexible GMRES (hx_gmres_sharded).
"""
import numpy as np
import scipy.sparse as sp

from .petsc import CSR, LowRank, Mat
from .solver_utils import info


class Blochifier:

    def __init__(self, geometry, boundary_conditions, N, passive_matrices, active_matrix=None, pairing="geometric",
                 index_order=None, global_geometry=None):
        self.shard = None
        if getattr(geometry.mesh, "part", None) is not None:
            if global_geometry is None:
                raise ValueError("a Bloch reduction on a distributed mesh needs global_geometry= (the whole sector mesh and its "
                                 "facet tags, which every rank holds as host arrays)")
            self._init_distributed(geometry, global_geometry, boundary_conditions, N, passive_matrices, active_matrix)
            return
        self.passive_matrices = passive_matrices
        self.active_matrix = active_matrix
        self.b = 1.0                                  # Bloch wave number
        self.f_b = np.exp(self.b * 1j * 2 * np.pi / N)
        self.mesh = geometry.mesh
        self.facet_tags = geometry.facet_tags
        self.V = passive_matrices.V
        self._A = self._B = self._B_adj = self._C = self._D = None

        master_tag = list(boundary_conditions.keys())[list(boundary_conditions.values()).index("Master")]
        slave_tag = list(boundary_conditions.keys())[list(boundary_conditions.values()).index("Slave")]
        dofs_master = np.unique(self.mesh.facets[self.facet_tags.find(master_tag)])
        dofs_slave = np.unique(self.mesh.facets[self.facet_tags.find(slave_tag)])
        assert len(dofs_master) == len(dofs_slave)
        if pairing == "geometric":
            dofs_slave = self._pair_by_rotation(dofs_master, dofs_slave, 2 * np.pi / N)
        elif pairing == "index":
            # the reference rule: k-th smallest master index with k-th smallest slave index.  ``index_order[i]`` = the
            # index node i has in the numbering that rule is to be applied in (e.g. DOLFINx's, to reproduce its logs)
            if index_order is not None:
                order = np.asarray(index_order)
                dofs_master = dofs_master[np.argsort(order[dofs_master], kind="stable")]
                dofs_slave = dofs_slave[np.argsort(order[dofs_slave], kind="stable")]
        else:
            raise ValueError("pairing must be 'geometric' or 'index'")
        self.dofs_master, self.dofs_slave = dofs_master, dofs_slave

        n = self.V.size
        self.N = n                                    # the reference overwrites N with the DOF count (:43)
        keep = np.setdiff1d(np.arange(n), dofs_master)
        col_of = np.full(n, -1, dtype=np.int64)
        col_of[keep] = np.arange(keep.size)
        # BN: full = BN @ reduced.  Kept dofs copy through; a master dof takes f_b times its slave's value.
        rows = np.concatenate([keep, dofs_master])
        cols = np.concatenate([col_of[keep], col_of[dofs_slave]])
        vals = np.concatenate([np.ones(keep.size, dtype=np.complex128), np.full(dofs_master.size, self.f_b)])
        self.BN_csr = sp.csr_matrix((vals, (rows, cols)), shape=(n, keep.size))
        # NB: reduced residual = NB @ full residual; master rows are folded onto their slave with 1/f_b.
        vals = np.concatenate([np.ones(keep.size, dtype=np.complex128), np.full(dofs_master.size, 1 / self.f_b)])
        self.NB_csr = sp.csr_matrix((vals, (cols, rows)), shape=(keep.size, n))
        self._BN = Mat.from_csr(CSR.from_scipy(self.BN_csr))
        self._NB_dev = self._BN_dev = None           # device copies of the two maps for the SpGEMM route (built on first use)

        self._A = self.blochify(passive_matrices.A)
        if passive_matrices.B:
            self._B = self.blochify(passive_matrices.B)
            self._B_adj = self._B.copy()
            self._B_adj.transpose()
            self._B_adj.conjugate()
        self._C = self.blochify(passive_matrices.C)
        info("- Bloch reduction: %d -> %d DOFs (%d master/slave pairs)" % (n, keep.size, dofs_master.size))

    # ------------------------------------------------------------------ row-sharded reduction
    def _init_distributed(self, geometry, G, boundary_conditions, N, passive_matrices, active_matrix):
        import torch
        from .sharded import ShardContext
        self.passive_matrices, self.active_matrix = passive_matrices, active_matrix
        self.b = 1.0
        self.f_b = np.exp(self.b * 1j * 2 * np.pi / N)
        self.mesh, self.facet_tags, self.V = geometry.mesh, geometry.facet_tags, passive_matrices.V
        self._A = self._B = self._B_adj = self._C = self._D = None
        part = self.mesh.part
        full = self.mesh.shard                       # ShardContext of the full (unreduced) numbering
        n_g, n_own, l2g = part["n_global"], part["n_own"], np.asarray(part["local_to_global"], dtype=np.int64)
        ranges = part["ranges"]
        # global pairing: integer host work on arrays every rank holds (identical result on every rank)
        master_tag = list(boundary_conditions.keys())[list(boundary_conditions.values()).index("Master")]
        slave_tag = list(boundary_conditions.keys())[list(boundary_conditions.values()).index("Slave")]
        gm = np.unique(G.mesh.facets[G.facet_tags.find(master_tag)])
        gs = np.unique(G.mesh.facets[G.facet_tags.find(slave_tag)])
        assert len(gm) == len(gs)
        x_keep, self.mesh = self.mesh, G.mesh        # (_pair_by_rotation reads self.mesh.geometry.x)
        try:
            gs = self._pair_by_rotation(gm, gs, 2 * np.pi / N)
        finally:
            self.mesh = x_keep
        starts = np.array([r[0] for r in ranges] + [ranges[-1][1]])
        owner = lambda ids: np.searchsorted(starts, ids, side="right") - 1          # noqa: E731
        if not np.array_equal(owner(gm), owner(gs)):
            raise ValueError("Bloch reduction on a distributed mesh: every master/slave pair must be owned by ONE rank -- partition "
                             "the sector in slabs along z or r (bloch_operator.sector_slab_numbering), not along theta")
        slave_of = np.full(n_g, -1, dtype=np.int64)
        slave_of[gm] = gs
        is_master = slave_of >= 0
        red_id = np.full(n_g, -1, dtype=np.int64)                                      # full id -> reduced id (kept dofs, order kept)
        red_id[~is_master] = np.arange(int((~is_master).sum()))
        ranges_red = []
        for (a, b) in ranges:
            kept = np.flatnonzero(~is_master[a:b])
            lo = int(red_id[a + kept[0]]) if kept.size else (ranges_red[-1][1] if ranges_red else 0)
            ranges_red.append((lo, lo + int(kept.size)))
        R0, R1 = ranges_red[part["rank"]]
        self.N = n_g
        # reduced column each local full column folds onto (a master: its slave partner's), and the reduced ghost set
        tgt = np.where(is_master[l2g], red_id[np.where(is_master[l2g], slave_of[l2g], 0)], red_id[l2g])
        ghost_red = np.unique(tgt[(tgt < R0) | (tgt >= R1)])
        n_keep_own, n_ext_red = R1 - R0, (R1 - R0) + ghost_red.size
        loc_red = np.where((tgt >= R0) & (tgt < R1), tgt - R0, n_keep_own + np.searchsorted(ghost_red, tgt))
        n_ext_full = l2g.size
        vals = np.where(is_master[l2g], self.f_b, 1.0 + 0j)
        self.BN_csr = sp.csr_matrix((vals, (np.arange(n_ext_full), loc_red)), shape=(n_ext_full, n_ext_red))
        own_full = np.arange(n_own)
        # NB: rows = owned reduced dofs, columns = owned full dofs (both partners of a pair are owned)
        assert np.all(loc_red[:n_own] < n_keep_own)
        nb_vals = np.where(is_master[l2g[:n_own]], 1 / self.f_b, 1.0 + 0j)
        self.NB_csr = sp.csr_matrix((nb_vals, (loc_red[:n_own], own_full)), shape=(n_keep_own, n_own))
        self.dofs_master = np.flatnonzero(is_master[l2g[:n_own]])                       # local (owned) ids
        r0 = int(ranges[part["rank"]][0])                                               # owned vertices: local id = global id - r0
        assert n_own == 0 or (l2g[0] == r0 and l2g[n_own - 1] == r0 + n_own - 1)
        self.dofs_slave = (slave_of[l2g[self.dofs_master]] - r0).astype(np.int64)
        # remapper: owned full entries from OWNED reduced entries (a master's slave is owned)
        self._BN_own = sp.csr_matrix((vals[:n_own], (own_full, loc_red[:n_own])), shape=(n_own, n_keep_own))
        self._BN = Mat.from_csr(CSR.from_scipy(self._BN_own))
        self._NB_dev = self._BN_dev = None
        self.shard = ShardContext(int((~is_master).sum()), group=full.group, ranges=ranges_red)
        self.shard.set_ghosts(ghost_red)
        self.full_shard = full
        self._A = self.blochify(passive_matrices.A)
        if passive_matrices.B:
            self._B = self.blochify(passive_matrices.B)
            # adjoint: (NB B BN)^H = BN^H B^H NB^H needs rows of other ranks in general; B is a boundary mass matrix (complex
            # symmetric up to the scalar i c / Z, acoustic_matrices.py:107-119), so B^H = conj(B) and the reduction commutes:
            # NB conj(B) BN with the conjugated phases = conj of (conj(NB) B conj(BN)) -- formed the same way from local rows
            self._B_adj = self._blochify_conj(passive_matrices.B)
        self._C = self.blochify(passive_matrices.C)
        info("- Bloch reduction (rank %d of %d): %d -> %d owned DOFs (%d master/slave pairs), %d reduced ghost columns"
             % (part["rank"], len(ranges), n_own, n_keep_own, self.dofs_master.size, ghost_red.size))

    def _blochify_conj(self, matrix):
        """(NB M BN)^H for a complex SYMMETRIC M, from local rows: (NB M BN)^H = BN^H conj(M) NB^H, and with NB = BN^H (|f_b| = 1)
        this is NB' conj(M) BN' with NB' = BN^H = NB, BN' = NB^H = BN: the same reduction applied to conj(M)."""
        m = matrix.copy()
        m.conjugate()
        return self.blochify(m)

    def _pair_by_rotation(self, dofs_master, dofs_slave, angle):
        """For every master DOF the slave DOF at the same (r, z), one sector pitch further (either sense)."""
        from scipy.spatial import cKDTree
        x = self.mesh.geometry.x
        xm, xs = x[dofs_master], x[dofs_slave]
        tree = cKDTree(xs)
        scale = max(np.ptp(xs, axis=0).max(), 1e-300)
        for a in (angle, -angle):
            c, s = np.cos(a), np.sin(a)
            rot = np.stack([c * xm[:, 0] - s * xm[:, 1], s * xm[:, 0] + c * xm[:, 1], xm[:, 2]], axis=1)
            dist, idx = tree.query(rot)
            if dist.max() <= 1e-6 * scale and len(np.unique(idx)) == len(idx):
                return dofs_slave[idx]
        raise ValueError("master/slave faces are not images of each other under a rotation by 2 pi / N")

    @property
    def A(self):
        return self._A

    @property
    def B(self):
        return self._B

    @property
    def B_adj(self):
        return self._B_adj

    @property
    def C(self):
        return self._C

    @property
    def D(self):
        return self._D

    @property
    def remapper(self):
        return self._BN

    def _triple_product_device(self, K):
        """NB (K BN) on the device: two sparse products with sorted output rows (deterministic sums)."""
        from .amg import DevCSR, _dev_spgemm
        if self._NB_dev is None:
            def dev(M):
                M = sp.csr_matrix(M, dtype=np.complex128)
                M.sort_indices()
                c = CSR.from_scipy(M)
                return DevCSR(M.shape, c.rowptr, c.col, c.val)
            self._NB_dev, self._BN_dev = dev(self.NB_csr), dev(self.BN_csr)
        red = _dev_spgemm(self._NB_dev, _dev_spgemm(DevCSR(K.shape, K.rowptr, K.col, K.val), self._BN_dev))
        return CSR(red.shape, red.rowptr, red.col, red.val)

    def blochify(self, matrix):
        """NB . M . BN for a lazy Mat: CSR part by a host triple product, low-rank part factor-wise."""
        m = self.NB_csr.shape[0]
        csr_terms, lr_terms = [], []
        K = matrix.csr_part()
        if self.shard is not None:
            # this rank's rows of the reduced operator over the [owned | ghost] reduced columns
            if K is not None:
                red = self._triple_product_device(K)
                red.local = True
                csr_terms.append((1.0, red))
            n_own = self.NB_csr.shape[1]
            for coef, lr in matrix.lr_terms:
                if not lr.local:
                    raise ValueError("flame factors of a distributed Bloch run must be assembled on the distributed mesh")
                pairs = []
                for (ui, uv), (vi, vv) in lr.pairs:
                    u = self.NB_csr @ sp.csr_matrix((uv, (ui, np.zeros_like(ui))), shape=(n_own, 1))
                    v = self._BN_own.T @ sp.csr_matrix((vv, (vi, np.zeros_like(vi))), shape=(n_own, 1))
                    u, v = u.tocoo(), v.tocoo()
                    pairs.append(((u.row.astype(np.int32), u.data), (v.row.astype(np.int32), v.data)))
                lr_terms.append((coef, LowRank(m, pairs, local=True)))
            return Mat((m, self.BN_csr.shape[1]), csr_terms, lr_terms)
        if K is not None:
            import os
            if os.environ.get("HX_BLOCH_HOST", "0") != "0":
                red = (self.NB_csr @ K.to_scipy() @ self.BN_csr).tocsr()
                csr_terms.append((1.0, CSR.from_scipy(red)))
            else:
                csr_terms.append((1.0, self._triple_product_device(K)))
        for coef, lr in matrix.lr_terms:
            pairs = []
            for (ui, uv), (vi, vv) in lr.pairs:
                u = self.NB_csr @ sp.csr_matrix((uv, (ui, np.zeros_like(ui))), shape=(lr.n, 1))
                v = self.BN_csr.T @ sp.csr_matrix((vv, (vi, np.zeros_like(vi))), shape=(lr.n, 1))
                u, v = u.tocoo(), v.tocoo()
                pairs.append(((u.row.astype(np.int32), u.data), (v.row.astype(np.int32), v.data)))
            lr_terms.append((coef, LowRank(m, pairs)))
        return Mat((m, m), csr_terms, lr_terms)


def sector_slab_numbering(mesh, axis="z"):
    """perm[old] = new vertex id of a sector mesh such that the numbering runs slab by slab along ``axis`` ("z" or "r"):
    contiguous row ranges are then z- (r-) slabs, and every master/slave pair of a Bloch sector -- two vertices with equal (r, z)
    -- lies in ONE range when the cuts fall between the planes (``ShardContext(align=...)`` / equal keys never split: the sort is
    stable and the returned ``plane_starts`` lists the first id of every plane).  Use with ``mesh.renumbered(perm)``."""
    x = mesh.geometry.x
    key = x[:, 2] if axis == "z" else np.hypot(x[:, 0], x[:, 1])
    scale = max(float(np.ptp(key)), 1e-300)
    q = np.round((key - key.min()) / scale * 1e9).astype(np.int64)          # planes: equal keys up to rounding
    order = np.argsort(q, kind="stable")                                    # order[new] = old
    perm = np.empty(order.size, dtype=np.int64)
    perm[order] = np.arange(order.size)
    qs = q[order]
    plane_starts = np.flatnonzero(np.concatenate([[True], qs[1:] != qs[:-1]]))
    return perm, plane_starts
