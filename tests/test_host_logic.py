"""CPU tests of the product's HOST logic (no GPU, no compute kernels): Krylov-Schur driver with a NumPy basis
double, native host set-up passes of the AMG hierarchy, mesh generators / readers, flame transfer functions,
C-ABI argument checking."""
import ctypes as C
import os

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from cases import Micca, Rijke


class NumpyBasis:
    """Test double for slepc.DeviceBasis: same interface, NumPy arithmetic (lives in tests/ only)."""

    def __init__(self, n, m, op, start=None):
        self.V = np.zeros((m, n), complex)
        self.m, self.n, self.op, self.start, self.n_ops = m, n, op, start, 0

    def set_start(self, j):
        if self.start is not None:
            v = self.start.copy()
        else:
            rng = np.random.default_rng(0)
            v = rng.standard_normal(self.n) + 1j * rng.standard_normal(self.n)
        self.V[j] = v / np.linalg.norm(v)

    def expand(self, j):
        w = self.op(self.V[j])
        self.n_ops += 1
        Vj = self.V[:j + 1]
        h1 = Vj.conj() @ w
        w = w - Vj.T @ h1
        h2 = Vj.conj() @ w
        w = w - Vj.T @ h2
        beta = np.linalg.norm(w)
        self.V[j + 1] = w / beta
        return h1 + h2, beta

    def rotate(self, Q, k):
        mact = Q.shape[0]
        new = Q.T @ self.V[:mact]
        res = self.V[mact].copy() if mact < self.m else None
        self.V[:k] = new
        if res is not None:
            self.V[k] = res


def _box_problem():
    from oracle import fem, meshes
    m = meshes.rectangle(48, 12, 0.4, 0.1)
    c = np.full(m.x.shape[0], 450.0)
    return fem.assemble_A(m.x, m.cells, c), fem.assemble_C(m.x, m.cells)


@pytest.mark.parametrize("nev,ncv,target_hz", [(2, 17, 600.0), (4, 8, 2500.0), (1, 16, 1200.0)])
def test_krylov_schur_matches_arpack(nev, ncv, target_hz):
    from helmholtz_x.krylov import krylov_schur
    from oracle import eigen
    A, Cm = _box_problem()
    n = A.shape[0]
    target = target_hz * 2 * np.pi
    sigma = target ** 2
    lu = spla.splu((A + sigma * Cm).tocsc())
    mC = (-Cm).tocsr()
    basis = NumpyBasis(n, ncv + 1, lambda v: lu.solve(mC @ v))
    res = krylov_schur(basis, nev, ncv)
    assert res.nconv >= nev
    lam = sigma + 1.0 / res.theta[:nev]
    ref = eigen.eps_solve(A, Cm, target, nev).eigs
    assert np.max(np.abs(np.sort_complex(lam) - np.sort_complex(ref)) / np.abs(ref).max()) < 1e-10
    for i in range(nev):                      # basis rows are unit Ritz VECTORS
        x = basis.V[i]
        assert abs(np.linalg.norm(x) - 1) < 1e-8
        r = A @ x + lam[i] * (Cm @ x)
        assert np.linalg.norm(r) <= 1e-9 * abs(A).max() * np.linalg.norm(x)


def test_krylov_schur_warm_start_needs_fewer_operator_applications():
    from helmholtz_x.krylov import krylov_schur
    A, Cm = _box_problem()
    n = A.shape[0]
    sigma = (600.0 * 2 * np.pi) ** 2
    lu = spla.splu((A + sigma * Cm).tocsc())
    mC = (-Cm).tocsr()
    cold = NumpyBasis(n, 18, lambda v: lu.solve(mC @ v))
    r0 = krylov_schur(cold, 1, 17)
    # slightly perturbed operator, started from the previous Ritz vector
    lu2 = spla.splu((A + 1.001 * sigma * Cm).tocsc())
    warm = NumpyBasis(n, 18, lambda v: lu2.solve(mC @ v), start=cold.V[0])
    r1 = krylov_schur(warm, 1, 17)
    assert r1.nconv >= 1 and warm.n_ops < cold.n_ops
    assert abs((1.001 * sigma + 1 / r1.theta[0]) - (sigma + 1 / r0.theta[0])) < 1e-6 * abs(sigma)


def test_native_amg_setup_passes_match_numpy():
    from helmholtz_x import amg
    from oracle import fem, meshes
    m = meshes.rectangle(30, 9, 0.4, 0.1)
    rng = np.random.default_rng(1)
    m.x[:, 1] *= 3.0                                        # anisotropy => some weak couplings
    S = -fem.assemble_A(m.x, m.cells, 300.0 + 50 * rng.random(m.x.shape[0])).real.tocsr()
    theta = 0.25
    Lf = amg._filter((-S).astype(np.complex128) + 0.3j * sp.eye(S.shape[0]), -1.0, theta)   # L = sign * Re(K), read in place
    d = np.abs(S.diagonal())
    coo = S.tocoo()
    off = coo.row != coo.col
    strong = off & (np.abs(coo.data) >= theta * np.sqrt(d[coo.row] * d[coo.col]))
    weak = off & ~strong
    assert weak.sum() > 0 and strong.sum() > 0
    ref = sp.csr_matrix((coo.data[~weak], (coo.row[~weak], coo.col[~weak])), shape=S.shape) \
        + sp.diags(np.bincount(coo.row[weak], weights=coo.data[weak], minlength=S.shape[0]))
    assert abs(Lf - ref).max() < 1e-12 * abs(ref).max()
    assert np.all(np.diff(Lf.indices)[np.diff(np.repeat(np.arange(S.shape[0]), np.diff(Lf.indptr))) == 0] > 0)  # sorted rows
    assert abs(amg._spectral_radius_dinv(Lf, iters=60) - np.max(np.abs(np.linalg.eigvals((sp.diags(1 / Lf.diagonal()) @ Lf).toarray())))) < 2e-2
    agg, na = amg._aggregate(Lf)
    assert agg.min() == 0 and agg.max() == na - 1 and len(np.unique(agg)) == na and na < 0.5 * S.shape[0]
    # every aggregate is connected through strong couplings to its root neighbourhood: members share a strong edge
    G = sp.csr_matrix((np.ones(Lf.nnz), Lf.indices, Lf.indptr), shape=Lf.shape)
    same = sp.csr_matrix((np.ones(len(agg)), (np.arange(len(agg)), agg)))
    assert ((G @ same).multiply(same).sum(axis=1) > 0).all()          # each node has a neighbour (or itself) in its aggregate
    omega = 0.6
    P = amg._prolongator(Lf, agg, na, omega)
    cnt = np.bincount(agg, minlength=na).astype(float)
    P0 = sp.csr_matrix((1 / np.sqrt(cnt[agg]), (np.arange(len(agg)), agg)), shape=(len(agg), na))
    refP = P0 - omega * (sp.diags(1 / Lf.diagonal()) @ Lf @ P0)
    assert abs(P - refP).max() < 1e-13
    lv = amg.build_hierarchy(sp.csr_matrix(-S + (50.0 + 5j) * sp.eye(S.shape[0])), coarse_size=40)
    assert len(lv) >= 2 and lv[-1]["n"] <= 40 or len(lv) == 8
    for a, b in zip(lv, lv[1:]):
        assert abs(b["K"] - a["R"] @ a["K"] @ a["P"]).max() < 1e-9 * abs(b["K"]).max()      # Galerkin


def test_cabi_argument_checks_return_codes_without_gpu():
    from helmholtz_x import _hx
    lib = _hx.lib
    one = (C.c_double * 2)(1.0, 0.0)
    assert lib.hx_zaxpy(-1, one, None, None, None) == -1                     # HX_ERR_ARG
    assert b"hx_zaxpy" in lib.hx_last_error()
    assert lib.hx_csr_spmv_z(5, 3, None, None, None, None, None, one, one, 0, None) == -1
    assert lib.hx_assemble_p1_stiff_mass(4, 10, None, None, None, None, None, None, None, 0, None, None, None) == -1
    # non-triangular factor is rejected by the host analysis before anything touches the device
    rp = np.array([0, 1, 2], np.int32)
    ci = np.array([1, 0], np.int32)
    va = np.ones(2, np.complex128)
    inv = np.ones(2, np.complex128)
    h = C.c_void_p()
    assert lib.hx_sptrsv_create(C.byref(h), 2, 1, _hx.nptr(rp), _hx.nptr(ci), _hx.nptr(va), _hx.nptr(inv)) == -1
    assert b"lower" in lib.hx_last_error()
    with pytest.raises(_hx.HxError):
        _hx.check(-1)


def test_annulus_generator_geometry_and_tags():
    from helmholtz_x.synthetic import annulus, box, micca_sound_speed
    from oracle import fem
    mesh, sub, ft = annulus(6, 48, 8)
    x, cells = mesh.geometry.x, mesh.topology.cells
    assert x.shape[0] == 7 * 48 * 9 and cells.shape[0] == 6 * 48 * 8 * 6
    _, vol = fem.cell_geometry(x, cells)
    assert vol.min() > 0
    exact = np.pi * (0.21 ** 2 - 0.14 ** 2) * 0.29
    assert abs(vol.sum() - exact) / exact < 0.01                      # polygonal ring
    area = fem.facet_measure(x, mesh.facets[ft.find(11)]).sum()
    assert abs(area - np.pi * (0.21 ** 2 - 0.14 ** 2)) / area < 0.01
    assert np.allclose(x[mesh.facets[ft.find(11)]][:, :, 2], 0.2)
    assert sorted(np.unique(sub.values)) == list(range(16))
    counts = np.bincount(sub.values)
    assert counts.min() == counts.max() > 0                           # 16 identical burners
    c = micca_sound_speed(mesh).x.array
    zc = x[cells].mean(axis=1)[:, 2]
    assert np.allclose(c, Micca.c_of_z(zc))
    bm, _, bft = box(3, 2, 2)
    assert bm.num_vertices == 4 * 3 * 3 and set(np.unique(bft.values)) == {1, 2, 3, 4, 5, 6}


def test_gmsh_reader_agrees_with_oracle_reader(golden_dir):
    from helmholtz_x.io_utils import XDMFReader
    from oracle import meshes
    for name, dim in (("rijke2d", 2), ("rijke3d", 3), ("micca_bloch_sector", 3)):
        r = XDMFReader(os.path.join(golden_dir, "meshes", name))
        mesh, ct, ft = r.getAll()
        o = meshes.read_gmsh41(os.path.join(golden_dir, "meshes", name + ".msh"), dim)
        assert np.array_equal(mesh.topology.cells, o.cells) and np.allclose(mesh.geometry.x, o.x)
        assert np.array_equal(mesh.facets, o.facets) and np.array_equal(mesh.facet_cells, o.facet_cells)
        assert np.array_equal(ft.values, o.facet_tags) and np.array_equal(ct.values, o.cell_tags)
        assert r.dimension == dim


def test_builtin_setups_and_fields_match_oracle():
    from helmholtz_x.dolfinx_utils import OneDimensionalSetup, RectangleSetup
    from helmholtz_x.parameters_utils import (c_step, density_step, gamma_function, gaussian, rho_step,
                                              sound_speed_variable_gamma, temperature_step)
    from oracle import fem, meshes
    mesh, sub, ft = OneDimensionalSetup(50)
    o = meshes.interval(50)
    assert np.array_equal(mesh.topology.cells, o.cells) and np.allclose(mesh.geometry.x, o.x)
    assert np.array_equal(mesh.facets[ft.find(1)].ravel(), [0]) and np.array_equal(mesh.facets[ft.find(2)].ravel(), [50])
    assert np.array_equal(sub.indices, np.flatnonzero(o.cell_tags == 0))
    mesh, sub, ft = RectangleSetup(8, 3, 0.4, 0.1)
    o = meshes.rectangle(8, 3, 0.4, 0.1)
    assert np.array_equal(np.sort(mesh.topology.cells, axis=1)[np.lexsort(np.sort(mesh.topology.cells, axis=1).T[::-1])],
                          np.sort(o.cells, axis=1)[np.lexsort(np.sort(o.cells, axis=1).T[::-1])])
    for tag in (1, 2, 3, 4):
        assert len(ft.find(tag)) == (o.facet_tags == tag).sum()
    T = temperature_step(mesh, Rijke.x_f(2), Rijke.T_u, Rijke.T_d)
    assert T.name == "temperature"
    assert np.allclose(sound_speed_variable_gamma(mesh, T).x.array, fem.sound_speed_variable_gamma(T.x.array))
    assert np.allclose(gamma_function(T).x.array, fem.gamma_of_T(T.x.array))
    assert np.isclose(gamma_function(300.0), fem.gamma_of_T(300.0))
    assert c_step(mesh, Rijke.x_f(2), 1.0, 2.0).name == "soundspeed"
    rho = rho_step(mesh, Rijke.x_f(2), Rijke.a_f, Rijke.rho_d, Rijke.rho_u)
    assert np.allclose(rho.x.array, density_step(mesh.geometry.x[:, 0], 0.25, Rijke.a_f, Rijke.rho_d, Rijke.rho_u))
    pts = mesh.geometry.x
    assert np.allclose(gaussian(pts.T, Rijke.x_r(2), 0.025, 2), fem.gaussian(pts, Rijke.x_r(2)[0], 0.025, 2))


def test_flame_transfer_functions_match_oracle(golden_dir):
    from scipy.io import loadmat
    from helmholtz_x.flame_transfer_function import nTau, stateSpace
    from oracle import eigen
    mat = loadmat(os.path.join(golden_dir, "micca_ftf.mat"))
    a, b = stateSpace(mat['A'], mat['b'], mat['c'], mat['d']), eigen.StateSpace(mat['A'], mat['b'], mat['c'], mat['d'])
    for w in (3225.12 + 481j, 1000.0, 50 - 20j):
        assert np.isclose(a(w), b(w), rtol=1e-13) and np.isclose(a.derivative(w), b.derivative(w), rtol=1e-13)
    n1, n2 = nTau(0.7, 1.5e-3), eigen.NTau(0.7, 1.5e-3)
    for w in (1242.2 + 6.7j, 3.4):
        assert np.isclose(n1(w), n2(w)) and np.isclose(n1.derivative(w), n2.derivative(w))


def test_h5lite_reads_reference_mesh_files(golden_dir):
    """The reference's meshes ship as XDMF + HDF5 (meshio output, chunked + deflate).  The package's own minimal HDF5
    reader must recover them exactly: node/cell counts from the XDMF header, tag ranges, positive cell volumes."""
    from helmholtz_x._h5lite import H5File
    from helmholtz_x.io_utils import read_xdmf
    base = os.path.join(golden_dir, "meshes", "micca_full")
    h = H5File(os.path.join(base, "mesh.h5"))
    assert sorted(h.datasets) == ["data0", "data1", "data2"]
    x = h.read("data0")
    cells = h.read("data1")
    tags = h.read("data2")
    assert x.shape == (34787, 3) and x.dtype == np.float64
    assert cells.shape == (163141, 4) and cells.min() == 0 and cells.max() == 34786
    assert tags.shape == (163141,) and set(np.unique(tags)) == set(range(18))
    e = x[cells[:, 1:]] - x[cells[:, :1]]
    vol = np.abs(np.linalg.det(e)) / 6
    assert vol.min() > 0 and abs(vol.sum() - 0.016595325942) < 1e-12      # plenum + 16 burners/flames + chamber [m^3]
    mesh, sub, ft = read_xdmf(os.path.join(base, "mesh"))
    assert mesh.geometry.x.shape == (34787, 3) and len(sub.values) == 163141
    assert len(ft.values) == 32908 and set(np.unique(ft.values)) == set(range(1, 12))
    assert len(ft.find(11)) > 0
    with pytest.raises(KeyError):
        h.read("nope")


def test_locate_slots_of_a_sub_pattern():
    """Host integer logic behind combine_csr: positions of a boundary-only pattern (B) inside the volume pattern (A/C)."""
    from helmholtz_x.petsc import locate_slots
    rng = np.random.default_rng(5)
    n = 400
    base = sp.random(n, n, density=0.03, random_state=3, format="csr") + sp.eye(n, format="csr")
    base.sort_indices()
    keep = rng.random(base.nnz) < 0.2
    keep[base.indptr[50]:base.indptr[300]] = False                     # whole row ranges untouched
    rows = np.repeat(np.arange(n), np.diff(base.indptr))
    sub = sp.csr_matrix((np.ones(keep.sum()), (rows[keep], base.indices[keep])), shape=(n, n))
    sub.sort_indices()
    slots = locate_slots(n, base.indptr.astype(np.int32), base.indices.astype(np.int32),
                         sub.indptr.astype(np.int32), sub.indices.astype(np.int32))
    assert slots.dtype == np.int32 and np.array_equal(slots, np.flatnonzero(keep))
    # an entry that is not in the base pattern -> None (caller falls back to the union pattern)
    extra = sub.tolil()
    r = 10
    missing = np.setdiff1d(np.arange(n), base.indices[base.indptr[r]:base.indptr[r + 1]])[0]
    extra[r, missing] = 1.0
    extra = extra.tocsr()
    extra.sort_indices()
    assert locate_slots(n, base.indptr.astype(np.int32), base.indices.astype(np.int32),
                        extra.indptr.astype(np.int32), extra.indices.astype(np.int32)) is None
    empty = sp.csr_matrix((n, n))
    assert locate_slots(n, base.indptr.astype(np.int32), base.indices.astype(np.int32),
                        empty.indptr.astype(np.int32), empty.indices.astype(np.int32)).size == 0


def test_harmonic_ritz_recycle_pair_algebra():
    """Host algebra of the Krylov-subspace recycling (amg.harmonic_ritz_recycle): from the Arnoldi relation of one GMRES
    cycle, K U = C and C^H C = I must hold, span(U) must contain the eigenvectors of smallest |eigenvalue|, and a GCRO
    solve that starts from the Galerkin guess on span(U) and stays orthogonal to C needs fewer steps."""
    from helmholtz_x.amg import harmonic_ritz_recycle
    rng = np.random.default_rng(11)
    n, m, k = 300, 40, 4
    lam = np.concatenate([[0.002, 0.004 + 0.001j, 0.007, 0.011], 1.0 + 0.3 * rng.random(n - 4) + 0.2j * rng.random(n - 4)])
    X = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))[0]
    K = X @ np.diag(lam) @ X.conj().T                    # normal matrix with four tiny eigenvalues

    def arnoldi(b, steps, Cm=None):
        V = [b / np.linalg.norm(b)]
        H = np.zeros((steps + 1, steps), dtype=complex)
        for j in range(steps):
            w = K @ V[j]
            if Cm is not None:
                w = w - Cm @ (Cm.conj().T @ w)
                w = w - Cm @ (Cm.conj().T @ w)
            for _ in range(2):
                h = np.array([np.vdot(v, w) for v in V])
                w = w - np.array(V).T @ h
                H[:j + 1, j] += h
            H[j + 1, j] = np.linalg.norm(w)
            V.append(w / H[j + 1, j])
        return np.array(V).T, H

    b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    V, Hb = arnoldi(b, m)
    Uc, Cc, theta = harmonic_ritz_recycle(Hb, k)
    U, Cm = V[:, :m] @ Uc, V @ Cc                        # unpreconditioned: Z = V
    assert np.abs(Cm.conj().T @ Cm - np.eye(k)).max() < 1e-12
    assert np.abs(K @ U - Cm).max() < 1e-10
    assert np.all(theta < 0.05)                          # the four outliers were found
    small = X[:, :4]
    Qu = np.linalg.qr(U)[0]
    assert np.linalg.norm(small - Qu @ (Qu.conj().T @ small)) < 1e-3

    def gmres_steps(b2, Cm=None, U=None, tol=1e-10):
        x0 = np.zeros(n, complex)
        r = b2.copy()
        if Cm is not None:
            t = Cm.conj().T @ r
            x0, r = U @ t, r - Cm @ t
        for steps in range(2, 80, 2):
            Vs, Hs = arnoldi(r, steps, Cm)
            e1 = np.zeros(steps + 1, complex)
            e1[0] = np.linalg.norm(r)
            y = np.linalg.lstsq(Hs, e1, rcond=None)[0]
            if np.linalg.norm(e1 - Hs @ y) <= tol * np.linalg.norm(b2):
                return steps
        return 80

    b2 = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    plain, recycled = gmres_steps(b2), gmres_steps(b2, Cm, U)
    assert recycled <= plain - 4, (plain, recycled)


def test_xdmf_writer_round_trip(tmp_path, golden_dir):
    """Minimal results writer (inline XML heavy data): values and mesh come back exactly."""
    import re
    from helmholtz_x.fem import Function, FunctionSpace
    from helmholtz_x.io_utils import read_gmsh, xdmf_writer
    mesh, _, _ = read_gmsh(os.path.join(golden_dir, "meshes", "rijke2d.msh"))
    p = Function(FunctionSpace(mesh, "CG", 1), "p")
    rng = np.random.default_rng(0)
    p.x.array[:] = rng.standard_normal(mesh.num_vertices) + 1j * rng.standard_normal(mesh.num_vertices)
    xdmf_writer(str(tmp_path / "out" / "p"), mesh, p)
    text = open(tmp_path / "out" / "p.xdmf").read()
    items = re.findall(r'<DataItem Dimensions="(\d+) (\d+)"[^>]*>\s*(.*?)\s*</DataItem>', text, flags=re.S)
    arrays = [np.array(body.split(), dtype=float).reshape(int(a), int(b)) for a, b, body in items]
    assert np.array_equal(arrays[0].astype(np.int64), mesh.topology.cells)
    assert np.array_equal(arrays[1], mesh.geometry.x)
    assert np.array_equal(arrays[2].ravel(), p.x.array.real) and np.array_equal(arrays[3].ravel(), p.x.array.imag)
    assert 'Name="real_f"' in text and 'Name="imag_f"' in text and 'TopologyType="Triangle"' in text


def test_amg_hierarchy_retries_and_refuses_a_huge_dense_coarse_level():
    """A strength threshold that classifies every coupling as weak stalls the aggregation; the builder relaxes it and
    retries instead of silently ending with one giant 'coarsest' level (which would be inverted densely)."""
    from helmholtz_x import amg
    from oracle import fem, meshes
    m = meshes.rectangle(60, 30, 0.4, 0.1)
    K = sp.csr_matrix(-fem.assemble_A(m.x, m.cells, np.full(m.x.shape[0], 300.0)) + (40.0 + 4j) * sp.eye(m.x.shape[0]))
    lv = amg.build_hierarchy(K, theta=0.9, coarse_size=50)       # 0.9: nothing is strong at first
    assert len(lv) >= 2 and lv[-1]["n"] <= amg.MAX_DENSE_COARSE
    old = amg.MAX_DENSE_COARSE
    try:
        amg.MAX_DENSE_COARSE = 10
        with pytest.raises(ArithmeticError):
            amg.build_hierarchy(sp.csr_matrix(sp.eye(500, dtype=np.complex128)), coarse_size=5)   # no couplings at all
    finally:
        amg.MAX_DENSE_COARSE = old


def test_mesh_renumbering_and_locality_order():
    """Mesh.renumbered keeps geometry, cells and tagged facets consistent; locality_permutation is a permutation that puts
    neighbours in space next to each other in memory (bandwidth of a randomly numbered grid drops by orders of magnitude)."""
    from helmholtz_x.fem import Mesh, locality_permutation
    from helmholtz_x.synthetic import annulus
    mesh, subdomains, facet_tags = annulus(4, 24, 5)
    n = mesh.num_vertices
    rng = np.random.default_rng(0)
    perm = rng.permutation(n)
    m2 = mesh.renumbered(perm)
    assert np.allclose(m2.geometry.x[perm], mesh.geometry.x)
    assert np.array_equal(np.sort(m2.geometry.x[m2.topology.cells].reshape(-1, 12), axis=1),
                          np.sort(mesh.geometry.x[mesh.topology.cells].reshape(-1, 12), axis=1))       # same tetrahedra
    assert np.array_equal(m2.facets, np.sort(perm[mesh.facets], axis=1)) and np.all(np.diff(m2.facets, axis=1) > 0)
    with pytest.raises(ValueError):
        mesh.renumbered(np.zeros(n, dtype=int))
    p = locality_permutation(m2.geometry.x)
    assert np.array_equal(np.sort(p), np.arange(n))
    m3 = m2.renumbered(p)

    def bandwidth(m):
        c = m.topology.cells.astype(np.int64)
        return int(np.abs(c[:, :, None] - c[:, None, :]).max(axis=(1, 2)).mean())
    assert bandwidth(m3) < 0.35 * bandwidth(m2)


def test_rcm_renumbering_at_ingestion(golden_dir, monkeypatch):
    """bandwidth_permutation (reverse Cuthill-McKee of the vertex graph) undoes a random numbering; locality_renumbered keeps
    the map back to the input numbering, also over repeated renumberings; XDMFReader applies it to meshes of HX_RENUMBER_MIN
    vertices or more (what DOLFINx does to every mesh the reference reads) with cell / facet tags carried over, and leaves
    smaller meshes -- all reference fixtures at the default limit -- in file numbering."""
    from helmholtz_x.fem import bandwidth_permutation, vertex_bandwidth
    from helmholtz_x.io_utils import XDMFReader
    from helmholtz_x.synthetic import annulus
    mesh, _, _ = annulus(4, 24, 5)
    n = mesh.num_vertices
    shuffled = mesh.renumbered(np.random.default_rng(0).permutation(n))
    perm = bandwidth_permutation(shuffled)
    assert np.array_equal(np.sort(perm), np.arange(n))
    m2 = shuffled.locality_renumbered("rcm")
    assert vertex_bandwidth(m2.topology.cells) < 0.25 * vertex_bandwidth(shuffled.topology.cells)
    assert np.array_equal(m2.geometry.x, shuffled.geometry.x[m2.input_vertex])
    m3 = m2.locality_renumbered("morton")
    assert np.array_equal(m3.geometry.x, shuffled.geometry.x[m3.input_vertex])
    with pytest.raises(ValueError):
        mesh.locality_renumbered("hilbert")
    # the reader
    path = os.path.join(golden_dir, "meshes", "rijke3d")
    mesh0, sub0, ft0 = XDMFReader(path).getAll()
    assert not hasattr(mesh0, "input_vertex")                                   # below the default limit: file numbering
    monkeypatch.setenv("HX_RENUMBER_MIN", "1")
    mesh1, sub1, ft1 = XDMFReader(path).getAll()
    assert hasattr(mesh1, "input_vertex")
    assert vertex_bandwidth(mesh1.topology.cells) < vertex_bandwidth(mesh0.topology.cells)
    assert np.array_equal(mesh1.geometry.x, mesh0.geometry.x[mesh1.input_vertex])
    inv = np.empty(mesh0.num_vertices, dtype=np.int64)
    inv[mesh1.input_vertex] = np.arange(mesh0.num_vertices)
    assert np.array_equal(mesh1.topology.cells, inv[mesh0.topology.cells])      # same cells, same order
    assert np.array_equal(mesh1.facets, np.sort(inv[mesh0.facets], axis=1))
    for a, b in ((sub0, sub1), (ft0, ft1)):
        assert np.array_equal(a.indices, b.indices) and np.array_equal(a.values, b.values) and b.mesh is mesh1
    monkeypatch.setenv("HX_RENUMBER_MIN", "0")
    assert not hasattr(XDMFReader(path).getAll()[0], "input_vertex")


def test_uniform_refinement_of_the_real_micca_mesh(golden_dir):
    """synthetic.refine_tets on the reference's own unstructured mesh: conforming (every face belongs to one or two cells,
    boundary faces = 4 x the parent's), volume and tagged areas preserved to rounding, every child an eighth of its parent
    with the parent's orientation, tags inherited, every child facet on a face of its adjacent child cell."""
    from itertools import combinations
    from helmholtz_x.io_utils import XDMFReader
    from helmholtz_x.synthetic import refine_tets
    mesh, sub, ft = XDMFReader(os.path.join(golden_dir, "meshes", "micca_full", "mesh")).getAll()
    fine, sub1, ft1 = refine_tets(mesh, sub, ft)

    def signed_volumes(m):
        X = m.geometry.x[m.topology.cells]
        return np.linalg.det(X[:, 1:] - X[:, :1]) / 6

    def faces_by_count(m):
        c = m.topology.cells.astype(np.int64)
        f = np.sort(np.concatenate([c[:, list(k)] for k in combinations(range(4), 3)]), axis=1)
        return np.bincount(np.unique(f, axis=0, return_counts=True)[1], minlength=3)

    def area(m, idx):
        X = m.geometry.x[m.facets[idx]]
        return 0.5 * np.linalg.norm(np.cross(X[:, 1] - X[:, 0], X[:, 2] - X[:, 0]), axis=1).sum()

    v0, v1 = signed_volumes(mesh), signed_volumes(fine)
    assert fine.num_cells == 8 * mesh.num_cells and fine.num_vertices > 7 * mesh.num_vertices
    assert np.allclose(v1.reshape(-1, 8), v0[:, None] / 8, rtol=1e-10, atol=0)
    n0, n1 = faces_by_count(mesh), faces_by_count(fine)
    assert n1[0] == 0 and len(n1) == 3 and n1[1] == 4 * n0[1]
    for tag in np.unique(ft.values):
        assert abs(area(fine, ft1.find(tag)) - area(mesh, ft.find(tag))) < 1e-12
    for tag in np.unique(sub.values):
        assert abs(np.abs(v1[sub1.find(tag)]).sum() - np.abs(v0[sub.find(tag)]).sum()) < 1e-15
    adj = np.sort(fine.topology.cells[fine.facet_cells], axis=1)
    assert np.all([(fine.facets[:, k, None] == adj).any(axis=1) for k in range(3)])


def test_translation_shifts_of_identical_flames():
    """Host logic behind the warm-started Woodbury columns: sparse vectors that are cyclic translates of the first one (with
    a support that wraps around the end of the numbering) are recognised with their shifts; one deviating value or a
    support of another shape switches the detection off altogether."""
    from helmholtz_x.slepc import translation_shifts
    n = 1000
    i0 = np.array([5, 6, 7, 990, 995], np.int32)
    v0 = np.array([1, 2 + 1j, 3, 4, 5 - 2j], complex)

    def tr(s):
        t = (i0.astype(np.int64) + s) % n
        o = np.argsort(t)
        return t[o].astype(np.int32), v0[o]

    assert translation_shifts([(i0, v0), tr(100), tr(20), tr(997)], n) == [0, 100, 20, 997]
    off = tr(20)
    assert translation_shifts([(i0, v0), tr(100), (off[0], off[1] * 1.001)], n) is None
    assert translation_shifts([(i0, v0), (np.array([1, 2, 3, 4, 9], np.int32), v0)], n) is None
    assert translation_shifts([(i0, v0), (i0[:4], v0[:4])], n) is None
    assert translation_shifts([(np.zeros(0, np.int32), np.zeros(0, complex)), tr(3)], n) is None


def _run_bench(args, env_extra):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **env_extra)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=600)


def test_bench_gpus_n_launches_n_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it (the way the driver runs N = 1) must start the two ranks
    itself, forward rank 0's one JSON line and exit 0 (reference: fullAnnulus/runAll.sh:9-10 runs
    `mpirun -np 8`).  The launch-check hook replaces the GPU work by one gloo all-reduce that counts the ranks."""
    import json
    p = _run_bench(["--gpus", "2", "--steps", "1"], {"HX_BENCH_LAUNCH_CHECK": "1"})
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1                               # exactly one line on stdout
    rec = json.loads(lines[0])
    assert rec["launch_check"] is True and rec["n_gpus"] == 2 and rec["gpus_arg"] == 2


def test_bench_failed_rank_gives_nonzero_exit_status():
    """A rank that fails (here: the last one returns 7 after the all-reduce; and, without the hook, every rank failing for
    want of a GPU) must surface in the exit status of `python bench.py --gpus N`."""
    p = _run_bench(["--gpus", "2"], {"HX_BENCH_LAUNCH_CHECK": "fail"})
    assert p.returncode != 0
    import torch
    if not torch.cuda.is_available():
        q = _run_bench(["--gpus", "2"], {})
        assert q.returncode != 0 and not q.stdout.decode().strip()
        assert "needs an MI355X" in q.stderr.decode()


def test_resident_field_is_frozen_until_released():
    """Function.resident() keeps a device copy for the assembly kernels; the host array is read-only while that copy
    exists, so a later `f.x.array[:] = ...` cannot leave it stale silently."""
    import torch
    from helmholtz_x.fem import Function, FunctionSpace
    from oracle import meshes
    from helmholtz_x.fem import Mesh
    m = meshes.rectangle(3, 2, 1.0, 1.0)
    f = Function(FunctionSpace(Mesh(m.x, m.cells), "DG", 0))
    f.x.array[:] = 2.0
    if not torch.cuda.is_available():
        f.x.array.flags.writeable = False           # what resident() does after its upload
        with pytest.raises(ValueError):
            f.x.array[:] = 3.0
        f.release()
        f.x.array[:] = 3.0
        assert f.x.array[0] == 3.0
        return
    f.resident()
    with pytest.raises(ValueError):
        f.x.array[:] = 3.0
    f.interpolate(lambda x: 0 * x[0] + 5.0)          # releases, writes, uploads again
    assert f.x.array[0] == 5.0 and complex(f._dev[0].item()) == 5.0
    with pytest.raises(ValueError):
        f.x.array[:] = 1.0


def test_sector_slab_numbering_keeps_bloch_pairs_in_one_row_range():
    """bloch_operator.sector_slab_numbering: renumbered z-plane by z-plane, a contiguous row partition cut between planes owns both
    vertices of every master/slave pair of a Bloch sector (equal r and z) -- the precondition of the row-sharded Bloch reduction;
    theta-slabs of the generator's numbering do not."""
    from helmholtz_x.bloch_operator import sector_slab_numbering
    from helmholtz_x.fem import MeshTags
    from helmholtz_x.synthetic import annulus
    half = np.pi / 16
    mesh, sub, ft = annulus(4, 6, 10, n_flames=1, sector=(-half, half))
    n = mesh.num_vertices
    perm, planes = sector_slab_numbering(mesh, "z")
    assert np.array_equal(np.sort(perm), np.arange(n)) and len(planes) == 11 and planes[0] == 0
    m2 = mesh.renumbered(perm)
    z = m2.geometry.x[:, 2]
    assert np.all(np.diff(z) >= -1e-12)
    ft2 = MeshTags(m2, ft.dim, ft.indices, ft.values)
    master = np.unique(m2.facets[ft2.find(12)])
    slave = np.unique(m2.facets[ft2.find(13)])
    cut = int(planes[5])
    x = m2.geometry.x
    key = lambda ids: np.round(np.stack([np.hypot(x[ids, 0], x[ids, 1]), x[ids, 2]], axis=1), 9)      # noqa: E731
    km, ks = key(master), key(slave)
    own_m, own_s = master < cut, slave < cut
    assert {tuple(r) for r in km[own_m]} == {tuple(r) for r in ks[own_s]}        # the same (r, z) set on either side of the cut
    # the generator's theta-major numbering: master plane first, slave plane last -> a two-way row split separates every pair
    m_old = np.unique(mesh.facets[ft.find(12)])
    s_old = np.unique(mesh.facets[ft.find(13)])
    assert m_old.max() < n // 2 <= s_old.min()


def test_set_from_options_maps_the_reference_option_names(monkeypatch):
    """E.setFromOptions() / Q.setFromOptions() (reference: eigensolvers.py:60,109) read -eps_* / -pep_* / -st_* from the options
    database (PETSC_OPTIONS + the command line, as petsc4py.init(sys.argv) does) and -hx_* for the library's own registry."""
    import sys
    from helmholtz_x import _hx
    from helmholtz_x.petsc import Options, PETSc
    from helmholtz_x.slepc import EPS, PEP
    monkeypatch.setattr(Options, "_db", None)                   # re-seed the process-wide database for this test
    monkeypatch.setenv("PETSC_OPTIONS", "-eps_nev 6 -eps_tol 1e-9 -st_pc_type lu")
    monkeypatch.setattr(sys, "argv", ["driver.py", "-nopopup", "-target", "1000", "-eps_ncv", "40", "-eps_max_it", "7",
                                      "-pep_nev", "3", "-pep_type", "toar", "-st_ksp_rtol", "1e-12", "-eps_target", "-1.5e3+2i",
                                      "-hx_cocg_single_reduction", "2", "-eps_two_sided"])
    db = PETSc.Options()
    assert db.getString("target") == "1000" and db.hasName("nopopup") and db.getInt("eps_ncv") == 40
    assert db.getReal("eps_tol") == 1e-9 and db.getScalar("eps_target") == -1500 + 2j       # "-1.5e3+2i" is a value, not a name
    E = EPS().create()
    E.setDimensions(2, -1)
    E.setTolerances(1e-15)
    E.setFromOptions()
    assert (E._nev, E._ncv, E._tol, E._max_it) == (6, 40, 1e-9, 7)
    assert E._target == -1500 + 2j and E._two_sided and E.st_type == "lu" and E._ksp_opts["rtol"] == 1e-12
    assert _hx.get_option("cocg_single_reduction") == "2"
    _hx.set_option("cocg_single_reduction", None)
    Q = PEP().create()
    Q.setDimensions(4, -1)
    Q.setFromOptions()
    assert Q._nev == 3 and Q._ncv is None and Q._tol == 1e-15 and Q.st_type == "lu"
    # the code-level argument of the drop-in functions is the default, the database overrides it -- and names that are not
    # on this path fail loudly instead of being ignored
    db["st_pc_type"] = "gamg"
    Q.setFromOptions()
    assert Q.st_type == "amg"
    db["pep_type"] = "qarnoldi"
    with pytest.raises(NotImplementedError):
        Q.setFromOptions()
    db.delValue("pep_type")
    db["st_type"] = "cayley"
    with pytest.raises(NotImplementedError):
        Q.setFromOptions()
    Options(prefix="st_").clear()
    assert not db.hasName("st_type") and db.hasName("eps_nev")


# ------------------------------------------------------------------------------------------------- direct solver: host analysis
def _splu_fields(hx, h):
    out = {}
    for name, dt in [("perm", np.int32), ("iperm", np.int32), ("sfirst", np.int32), ("sparent", np.int32), ("sptr", np.int64),
                     ("srow", np.int32), ("rel", np.int32), ("slevel", np.int32), ("level_ptr", np.int32), ("level_list", np.int32),
                     ("child_ptr", np.int32), ("child_list", np.int32), ("loff", np.int64), ("uoff", np.int64), ("soff", np.int64),
                     ("amap", np.int64)]:
        cnt = C.c_int64()
        hx.check(hx.lib.hx_splu_symbolic(h, name.encode(), None, 0, C.byref(cnt)))
        a = np.empty(cnt.value, dt)
        hx.check(hx.lib.hx_splu_symbolic(h, name.encode(), hx.nptr(a), a.nbytes, C.byref(cnt)))
        out[name] = a
    return out


@pytest.mark.parametrize("with_coords", [True, False])
@pytest.mark.parametrize("case", ["box3d", "path", "two_components", "unsymmetric_pattern", "tiny"])
def test_direct_solver_analysis_structure_contains_the_fill(case, with_coords):
    """hx_splu_analyse (host only): the ordering is a permutation, supernodes tile the columns, every supernode's parent owns
    its first border row, the extend-add positions point at the same global rows, every stored entry has its own slot in the
    right panel, and the supernodal structure CONTAINS the fill of an elimination without pivoting (dense symbolic elimination of
    the permuted pattern) -- with and without coordinates."""
    from helmholtz_x import _hx as hx
    from oracle import fem, meshes
    rng = np.random.default_rng(3)
    xy = None
    if case == "box3d":
        from helmholtz_x import synthetic
        m = synthetic.annulus(3, 24, 4)[0]
        x, cells = m.geometry.x, m.topology.cells
        K = fem.assemble_C(x, cells).tocsr()
        xy = x
    elif case == "path":
        n = 700
        K = sp.diags([np.ones(n - 1), np.ones(n), np.ones(n - 1)], [-1, 0, 1]).tocsr()
        xy = np.linspace(0, 1, n)[:, None]
    elif case == "two_components":
        m = meshes.rectangle(12, 9, 1.0, 1.0)
        A = fem.assemble_C(m.x, m.cells).tocsr()
        K = sp.block_diag([A, A]).tocsr()
        xy = np.vstack([m.x[:, :2], m.x[:, :2] + [3.0, 0.0]])
    elif case == "unsymmetric_pattern":
        K = sp.random(400, 400, density=0.01, random_state=rng, format="csr") + sp.eye(400)
        K = sp.csr_matrix(sp.triu(K) + sp.random(400, 400, density=0.002, random_state=rng))
        xy = rng.random((400, 2))
    else:
        K = sp.csr_matrix(np.array([[2.0, 1.0], [0.0, 3.0]]))
        xy = np.array([[0.0], [1.0]])
    K = sp.csr_matrix(K)
    K.sort_indices()
    n = K.shape[0]
    rp, ci = K.indptr.astype(np.int32), K.indices.astype(np.int32)
    xyc = np.ascontiguousarray(xy, np.float64) if with_coords else None
    h = C.c_void_p()
    hx.check(hx.lib.hx_splu_analyse(C.byref(h), n, hx.nptr(rp), hx.nptr(ci), None if xyc is None else hx.nptr(xyc),
                                    0 if xyc is None else xyc.shape[1], 16, 0))
    try:
        f = _splu_fields(hx, h)
        io, do = (C.c_int64 * 16)(), (C.c_double * 8)()
        hx.check(hx.lib.hx_splu_info(h, io, do))
        assert io[0] == n and io[1] == len(f["sfirst"]) - 1
        perm, iperm, sfirst, sptr, srow = f["perm"], f["iperm"], f["sfirst"], f["sptr"], f["srow"]
        assert np.array_equal(np.sort(perm), np.arange(n)) and np.array_equal(iperm[perm], np.arange(n))
        assert sfirst[0] == 0 and sfirst[-1] == n and np.all(np.diff(sfirst) > 0)
        nsup = len(sfirst) - 1
        super_of = np.repeat(np.arange(nsup), np.diff(sfirst))
        for s in range(nsup):
            rows = srow[sptr[s]:sptr[s + 1]]
            assert np.all(np.diff(rows) > 0) and (len(rows) == 0 or rows[0] >= sfirst[s + 1])
            p = f["sparent"][s]
            assert (p == -1) == (len(rows) == 0)
            if p >= 0:
                assert super_of[rows[0]] == p and f["slevel"][p] > f["slevel"][s]
                front_p = np.concatenate([np.arange(sfirst[p], sfirst[p + 1]), srow[sptr[p]:sptr[p + 1]]])
                assert np.array_equal(front_p[f["rel"][sptr[s]:sptr[s + 1]]], rows)
        dest = np.where(f["amap"] < 0, -f["amap"] - 1, f["amap"])          # (entries of U panels are stored as -(destination + 1))
        assert len(np.unique(dest)) == K.nnz and dest.min() >= 0 and dest.max() < io[4]
        # dense symbolic elimination of the permuted, symmetrised pattern
        P = ((abs(K) + abs(K).T) != 0).toarray()[np.ix_(perm, perm)]
        np.fill_diagonal(P, True)
        for k in range(n):
            below = np.nonzero(P[k + 1:, k])[0] + k + 1
            P[np.ix_(below, below)] = True
        for s in range(nsup):
            allowed = np.zeros(n, bool)
            allowed[sfirst[s]:sfirst[s + 1]] = True
            allowed[srow[sptr[s]:sptr[s + 1]]] = True
            for j in range(sfirst[s], sfirst[s + 1]):
                assert not np.any(P[j:, j] & ~allowed[j:]), (case, s, j)
    finally:
        hx.check(hx.lib.hx_splu_destroy(h))


def test_every_rank_generates_its_own_slab_of_the_annulus():
    """synthetic.annulus_part (what `bench.py --gpus N` uses on the analytic workload: no rank holds the global mesh) equals
    distributed.distribute_mesh of the global mesh BIT FOR BIT -- coordinates, cells, facets, maps, flame tags, sound speed -- for
    1, 2, 4 and 5 ranks; the owned ranges tile the vertex set and the owned cells tile the cell set."""
    from helmholtz_x import synthetic
    from helmholtz_x.distributed import distribute_mesh, partition_rows
    shape = (3, 24, 4)
    mesh, sub, ft = synthetic.annulus(*shape)
    c = synthetic.micca_sound_speed(mesh)
    plane = (shape[0] + 1) * (shape[2] + 1)
    for world in (1, 2, 4, 5):
        ranges = partition_rows(mesh.num_vertices, world, plane)
        owned_cells = np.zeros(mesh.num_cells, dtype=int)
        rows = 0
        for rank in range(world):
            a, sa, fa, (ca,) = distribute_mesh(mesh, ranges, rank, sub, ft, [c])
            b, sb, fb = synthetic.annulus_part(*shape, ranges, rank)
            cb = synthetic.micca_sound_speed(b)
            assert np.array_equal(a.geometry.x, b.geometry.x) and np.array_equal(a.topology.cells, b.topology.cells)
            assert np.array_equal(a.facets, b.facets) and np.array_equal(a.facet_cells, b.facet_cells)
            assert sorted(a.part) == sorted(b.part)
            for k in a.part:
                assert np.array_equal(np.asarray(a.part[k]), np.asarray(b.part[k])), (k, world, rank)
            assert np.array_equal(sa.indices, sb.indices) and np.array_equal(sa.values, sb.values)
            assert np.array_equal(fa.indices, fb.indices) and np.array_equal(fa.values, fb.values)
            assert np.array_equal(ca.x.array, cb.x.array)
            owned_cells[b.part["cell_ids"][b.part["cell_owned"]]] += 1
            rows += b.part["n_own"]
        assert rows == mesh.num_vertices and np.all(owned_cells == 1)
    with pytest.raises(ValueError):
        synthetic.annulus_part(*shape, [(0, 7), (7, mesh.num_vertices)], 0)        # not cut between theta-planes
