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
"""
import numpy as np
import scipy.sparse as sp

from .petsc import CSR, LowRank, Mat
from .solver_utils import info


class Blochifier:

    def __init__(self, geometry, boundary_conditions, N, passive_matrices, active_matrix=None, pairing="geometric",
                 index_order=None):
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
