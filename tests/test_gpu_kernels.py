"""GPU parity of the libhx kernels against NumPy/SciPy and the CPU oracle (called through the C-ABI)."""
import ctypes as C
import os

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hx():
    import torch
    assert torch.cuda.is_available(), "gpu-marked tests need the MI355X"
    from helmholtz_x import _hx
    return _hx


def _dev(a):
    from helmholtz_x.petsc import to_device
    return to_device(a)


def _rand_c(rng, *shape):
    return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)


def _rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(np.asarray(b))), 1e-300))


@pytest.mark.parametrize("n,density", [(1, 1.0), (257, 0.05), (5000, 0.003), (20000, 0.0008)])
def test_spmv_random_all_lane_widths(hx, n, density):
    from helmholtz_x.petsc import CSR
    import torch
    rng = np.random.default_rng(n)
    M = sp.random(n, n, density=density, random_state=rng, format="csr", dtype=np.float64)
    M = M + 1j * sp.random(n, n, density=density, random_state=rng, format="csr") + sp.eye(n)
    M = sp.csr_matrix(M)
    A = CSR.from_scipy(M)
    x = _rand_c(rng, n)
    y0 = _rand_c(rng, n)
    alpha, beta = 0.7 - 0.2j, -1.3 + 0.4j
    ref = alpha * (M @ x) + beta * y0
    for lanes in (0, 1, 2, 4, 8, 16, 32, 64):
        y, xd = _dev(y0.copy()), _dev(x)
        A.spmv(xd, y, alpha, beta, lanes)
        got = y.cpu().numpy()
        assert np.max(np.abs(got - ref)) <= 1e-13 * np.max(np.abs(ref)), lanes
    y = _dev(np.full(n, np.nan + 0j))        # beta == 0 must ignore the previous content of y
    xd = _dev(x)
    A.spmv(xd, y)
    assert np.allclose(y.cpu().numpy(), M @ x, rtol=1e-13, atol=1e-13)
    torch.cuda.synchronize()


def test_spmv_empty_rows_and_ragged(hx):
    from helmholtz_x.petsc import CSR
    rng = np.random.default_rng(3)
    n = 1000
    rows = rng.integers(0, n // 2, 4000) * 2          # odd rows are empty
    cols = rng.integers(0, n, 4000)
    M = sp.csr_matrix((_rand_c(rng, 4000), (rows, cols)), shape=(n, n))
    M[10, :] = _rand_c(rng, n)                          # one very long row
    M = sp.csr_matrix(M)
    x = _rand_c(rng, n)
    for lanes in (0, 1, 8, 64):
        y, xd, Ad = _dev(np.zeros(n, complex)), _dev(x), CSR.from_scipy(M)
        Ad.spmv(xd, y, lanes=lanes)
        assert np.allclose(y.cpu().numpy(), M @ x, rtol=1e-12, atol=1e-12)


def test_locate_slots_on_the_device_matches_the_host_map(hx):
    """hx_csr_locate_slots (behind combine_csr: B's boundary pattern inside the volume pattern, `A + omega*B`,
    eigensolvers.py:174-176) against the host integer map; a missing entry and an empty pattern."""
    from helmholtz_x.petsc import CSR, combine_csr, locate_slots, locate_slots_device
    rng = np.random.default_rng(5)
    n = 3000
    base = sp.random(n, n, density=0.004, random_state=3, format="csr") + sp.eye(n, format="csr")
    base.sort_indices()
    keep = rng.random(base.nnz) < 0.2
    keep[base.indptr[500]:base.indptr[2200]] = False
    rows = np.repeat(np.arange(n), np.diff(base.indptr))
    sub = sp.csr_matrix((_rand_c(rng, int(keep.sum())), (rows[keep], base.indices[keep])), shape=(n, n))
    sub.sort_indices()
    cb = CSR.from_scipy(sp.csr_matrix(base, dtype=np.complex128))
    cs = CSR.from_scipy(sub)
    ref = locate_slots(n, base.indptr.astype(np.int32), base.indices.astype(np.int32),
                       sub.indptr.astype(np.int32), sub.indices.astype(np.int32))
    got = locate_slots_device(cb, cs)
    assert got.dtype == __import__("torch").int32 and np.array_equal(got.cpu().numpy(), ref)
    K = combine_csr((n, n), [(1.0, cb), (0.5 - 2j, cs)])
    want = sp.csr_matrix(base, dtype=np.complex128) + (0.5 - 2j) * sub
    assert abs(K.to_scipy() - want).max() < 1e-13
    extra = sub.tolil()
    r = 10
    absent = np.setdiff1d(np.arange(n), base.indices[base.indptr[r]:base.indptr[r + 1]])[0]
    extra[r, absent] = 1.0
    extra = extra.tocsr()
    extra.sort_indices()
    ce = CSR.from_scipy(sp.csr_matrix(extra, dtype=np.complex128))
    assert locate_slots_device(cb, ce) is None
    K2 = combine_csr((n, n), [(1.0, cb), (1.0, ce)])                   # union pattern on the host path
    assert abs(K2.to_scipy() - (sp.csr_matrix(base, dtype=np.complex128) + extra)).max() < 1e-13
    empty = CSR.from_scipy(sp.csr_matrix((n, n), dtype=np.complex128))
    assert locate_slots_device(cb, empty).numel() == 0


def test_vector_and_panel_kernels(hx):
    import torch
    from helmholtz_x.petsc import panel_dot
    rng = np.random.default_rng(0)
    n, m = 70001, 21
    V = _rand_c(rng, m, n)
    w = _rand_c(rng, n)
    Vd, wd = _dev(V), _dev(w)
    for k in (0, 1, 7, 8, 9, 20, 21):
        got = panel_dot(Vd, k, wd)
        ref = np.concatenate([V[:k].conj() @ w, [np.vdot(w, w)]])
        assert np.allclose(got, ref, rtol=1e-12, atol=1e-9), k
    got2 = panel_dot(Vd, 9, wd)
    assert np.array_equal(got2, panel_dot(Vd, 9, wd))          # deterministic reduction
    h = _rand_c(rng, m)
    w2 = _dev(w.copy())
    hd = _dev(h)
    hx.check(hx.lib.hx_zpanel_sub(n, m, hx.ptr(Vd), n, hx.ptr(hd), hx.ptr(w2), hx.stream()))
    assert np.allclose(w2.cpu().numpy(), w - V.T @ h, rtol=1e-12, atol=1e-12)
    Q = _rand_c(rng, m, 6)
    out = torch.zeros((6, n), dtype=torch.complex128, device=Vd.device)
    Qd = _dev(np.ascontiguousarray(Q.T))
    hx.check(hx.lib.hx_zpanel_mul(n, m, 6, hx.ptr(Vd), n, hx.ptr(Qd), hx.ptr(out), n, hx.stream()))
    assert np.allclose(out.cpu().numpy(), Q.T @ V, rtol=1e-12, atol=1e-11)
    # axpy / scal / lincomb / gather / scatter
    a = 0.3 - 1.1j
    y = _dev(w.copy())
    hx.check(hx.lib.hx_zaxpy(n, hx.z2(a), hx.ptr(Vd[3]), hx.ptr(y), hx.stream()))
    assert np.allclose(y.cpu().numpy(), w + a * V[3])
    hx.check(hx.lib.hx_zscal(n, hx.z2(a), hx.ptr(y), hx.stream()))
    assert np.allclose(y.cpu().numpy(), a * (w + a * V[3]))
    arr = (C.c_void_p * 3)(Vd[0].data_ptr(), Vd[1].data_ptr(), Vd[2].data_ptr())
    cf = (C.c_double * 6)(1, 0, 0, 2, -1, 0.5)
    hx.check(hx.lib.hx_zlincomb(n, 3, cf, arr, hx.ptr(y), hx.stream()))
    assert np.allclose(y.cpu().numpy(), V[0] + 2j * V[1] + (-1 + 0.5j) * V[2])
    idx = rng.permutation(n).astype(np.int32)
    idx_d = _dev(idx)
    hx.check(hx.lib.hx_zgather(n, hx.ptr(wd), hx.ptr(idx_d), hx.ptr(y), hx.stream()))
    assert np.array_equal(y.cpu().numpy(), w[idx])
    dst = _dev(V[5].copy())
    hx.check(hx.lib.hx_zscatter_axpy(n, hx.z2(a), hx.ptr(wd), hx.ptr(idx_d), hx.ptr(dst), hx.stream()))
    ref = V[5].copy()
    ref[idx] += a * w
    assert np.allclose(dst.cpu().numpy(), ref)
    # sparse dots
    ptr = np.array([0, 5, 5, 400], np.int32)
    sidx = rng.integers(0, n, 400).astype(np.int32)
    sval = _rand_c(rng, 400)
    out3 = torch.zeros(3, dtype=torch.complex128, device=Vd.device)
    ptr_d, sidx_d, sval_d = _dev(ptr), _dev(sidx), _dev(sval)     # keep alive: temporaries would be recycled
    hx.check(hx.lib.hx_zsparse_dots(3, hx.ptr(ptr_d), hx.ptr(sidx_d), hx.ptr(sval_d), hx.ptr(wd), hx.ptr(out3),
                                    hx.stream()))
    ref3 = [np.sum(sval[ptr[f]:ptr[f + 1]] * w[sidx[ptr[f]:ptr[f + 1]]]) for f in range(3)]
    assert np.allclose(out3.cpu().numpy(), ref3, rtol=1e-12, atol=1e-12)


def test_vec_mat_wrappers(hx):
    from helmholtz_x.petsc import CSR, LowRank, Mat, Vec
    rng = np.random.default_rng(5)
    n = 300
    M1 = sp.random(n, n, 0.05, random_state=rng, format="csr") + sp.eye(n)
    M2 = sp.random(n, n, 0.02, random_state=rng, format="csr") * 1j
    A, B = Mat.from_csr(CSR.from_scipy(M1)), Mat.from_csr(CSR.from_scipy(M2))
    lr = LowRank(n, [((np.array([1, 5, 9], np.int32), _rand_c(rng, 3)), (np.array([0, 2], np.int32), _rand_c(rng, 2))),
                     ((np.array([7], np.int32), _rand_c(rng, 1)), (np.arange(20, dtype=np.int32), _rand_c(rng, 20)))])
    D = Mat((n, n), lr_terms=[(1.0, lr)])
    w = np.complex128(3.0 - 2.0j)
    T = A + (w * B) - D * (0.5 + 0.1j)
    ref = M1 + w * M2 - (0.5 + 0.1j) * lr.to_scipy()
    assert np.allclose(T.to_scipy().toarray(), ref.toarray())
    x = Vec(_rand_c(rng, n))
    y = Vec.zeros(n)
    T.mult(x, y)
    assert np.allclose(y.array, ref @ x.array, rtol=1e-12, atol=1e-12)
    Bh = B.copy()
    Bh.transpose()
    Bh.conjugate()
    assert np.allclose(Bh.to_scipy().toarray(), M2.conj().T.toarray())
    # PETSc VecDot(x, y) = y^H x
    z = Vec(_rand_c(rng, n))
    assert np.isclose(x.dot(z), np.vdot(z.array, x.array))
    assert np.isclose(x.norm(), np.linalg.norm(x.array))


@pytest.mark.parametrize("n,lower", [(50, 1), (4000, 1), (4000, 0)])
def test_sptrsv_against_scipy(hx, n, lower):
    import torch
    rng = np.random.default_rng(n + lower)
    T = sp.random(n, n, density=min(1.0, 8.0 / n), random_state=rng, format="csr")
    T = T + 1j * sp.random(n, n, density=min(1.0, 8.0 / n), random_state=rng, format="csr")
    # add a long dependency chain and a dense trailing block, like LU factors have
    chain = sp.diags([np.ones(n - 1)], [-1]) * 0.3
    dense = sp.lil_matrix((n, n), dtype=complex)
    dense[n - 40:, n - 40:] = _rand_c(rng, 40, 40) * 0.1
    T = sp.tril(T + chain + dense.tocsr(), -1).tocsr()
    if not lower:
        T = T.T.tocsr()
    T.sort_indices()
    diag = 2.0 + _rand_c(rng, n) * 0.2
    full = (T + sp.diags(diag)).tocsr()
    b = _rand_c(rng, n)
    ref = spla.spsolve_triangular(full, b, lower=bool(lower))
    h = C.c_void_p()
    rp, ci = T.indptr.astype(np.int32), T.indices.astype(np.int32)
    va, inv = T.data.astype(np.complex128), (1.0 / diag).astype(np.complex128)
    hx.check(hx.lib.hx_sptrsv_create(C.byref(h), n, lower, hx.nptr(rp), hx.nptr(ci), hx.nptr(va), hx.nptr(inv)))
    xb = _dev(b.copy())
    hx.check(hx.lib.hx_sptrsv_solve(h, hx.ptr(xb), hx.ptr(xb), hx.stream()))     # in place
    torch.cuda.synchronize()
    got = xb.cpu().numpy()
    assert np.max(np.abs(got - ref)) <= 1e-11 * np.max(np.abs(ref))
    hx.check(hx.lib.hx_sptrsv_destroy(h))


def test_lu_solver_matches_splu(hx):
    from helmholtz_x.petsc import CSR
    from helmholtz_x.slepc import HostLUSolver, LUSolver
    from oracle import fem, meshes
    m = meshes.rectangle(60, 20, 0.4, 0.1)
    c = np.full(m.x.shape[0], 450.0)
    K = (fem.assemble_A(m.x, m.cells, c) + (3000.0 + 200j) ** 2 * fem.assemble_C(m.x, m.cells)).tocsr()
    rng = np.random.default_rng(1)
    b = _rand_c(rng, K.shape[0])
    ref = spla.splu(K.tocsc()).solve(b)
    for cls in (LUSolver, HostLUSolver):          # device multifrontal LU (the product route) / round-1 host SuperLU route
        S = cls(CSR.from_scipy(K))
        x, bd = _dev(np.zeros_like(b)), _dev(b)
        S.solve(bd, x)
        assert np.max(np.abs(x.cpu().numpy() - ref)) <= 1e-10 * np.max(np.abs(ref)), cls.__name__
        S.destroy()


def _micca_K(golden_dir, level):
    """K = A + sigma B + sigma^2 C of the reference's MICCA case at its own target (oracle assembly on the host), mesh
    refined ``level`` times; returns (K csr, coordinates)."""
    import os
    from cases import Micca
    from helmholtz_x import synthetic
    from helmholtz_x.io_utils import read_xdmf
    from oracle import fem
    mesh, sub, ft = read_xdmf(os.path.join(golden_dir, "meshes", "micca_full", "mesh"))
    for _ in range(level):
        mesh, sub, ft = synthetic.refine_tets(mesh, cell_tags=sub, facet_tags=ft)
    x, cells = mesh.geometry.x, mesh.topology.cells
    c = Micca.c_of_z(x[cells].mean(axis=1)[:, 2])
    sel = ft.find(11)
    B = fem.assemble_B(x, x.shape[0], [(mesh.facets[sel], fem.robin_Z(Micca.R_outlet), c, mesh.facet_cells[sel])])
    s = Micca.target
    K = (fem.assemble_A(x, cells, c, c_is_cellwise=True) + s * B + s * s * fem.assemble_C(x, cells)).tocsr()
    K.sort_indices()
    return K, x


_SUPERLU = {}


@pytest.mark.parametrize("level,with_coords,flags", [(0, True, 1), (0, True, 0), (0, False, 1), (1, True, 1)])
def test_device_sparse_lu_on_the_real_micca_operator(hx, golden_dir, level, with_coords, flags):
    """hx_splu_* on the shifted operator of the reference's flagship case: 34,787 DOF against SciPy's SuperLU (solution and
    transposed solution to 1e-10, both orderings), 249,497 DOF (one uniform refinement; SuperLU would need minutes and tens of
    GB there) against a manufactured solution b = K x and its residual.  No refinement step is allowed to hide anything: the
    raw solves of the C-ABI are compared.  flags = 1 (the pattern is symmetric; what LUSolver passes for Galerkin operators): the
    values are found symmetric on the device and the factorisation runs in symmetric mode; flags = 0: the general algorithm."""
    import torch
    K, xyz = _micca_K(golden_dir, level)
    n = K.shape[0]
    assert n == (34787, 249497)[level]
    rng = np.random.default_rng(level)
    x_true = _rand_c(rng, n)
    b = K @ x_true
    rp, ci = K.indptr.astype(np.int32), K.indices.astype(np.int32)
    xy = np.ascontiguousarray(xyz, np.float64)
    h = C.c_void_p()
    hx.check(hx.lib.hx_splu_analyse(C.byref(h), n, hx.nptr(rp), hx.nptr(ci), hx.nptr(xy) if with_coords else None,
                                    3 if with_coords else 0, 0, flags))
    val, bd = _dev(np.ascontiguousarray(K.data)), _dev(b)
    xd = torch.empty_like(bd)
    io, do = (C.c_int64 * 16)(), (C.c_double * 8)()
    for rep in range(2):                               # the second factorisation reuses plan and storage
        hx.check(hx.lib.hx_splu_factor(h, hx.ptr(val), hx.stream()))
        hx.check(hx.lib.hx_splu_info(h, io, do))
        assert io[12] == 0                              # no pivot had to be replaced
        assert bool(do[5]) == bool(flags)               # symmetric mode exactly when the caller vouched for the pattern
        hx.check(hx.lib.hx_splu_solve(h, hx.ptr(bd), hx.ptr(xd), 0, hx.stream()))
        x = xd.cpu().numpy()
        assert np.linalg.norm(K @ x - b) <= 1e-12 * np.linalg.norm(b)
        assert np.max(np.abs(x - x_true)) <= 1e-10 * np.max(np.abs(x_true))
    print("device LU n = %d (%s, %s mode): nnz(L) %.3e, fill %.1f, %.2e flops executed, factorisation %.1f ms = %.1f TFLOP/s"
          % (n, "inertial" if with_coords else "level-set", "symmetric" if do[5] else "general", io[3], 2 * io[3] / K.nnz, do[4],
             do[2] * 1e3, do[4] / do[2] / 1e12))
    bt = K.T @ x_true
    btd = _dev(bt)
    hx.check(hx.lib.hx_splu_solve(h, hx.ptr(btd), hx.ptr(btd), 1, hx.stream()))        # transposed, in place
    assert np.max(np.abs(btd.cpu().numpy() - x_true)) <= 1e-10 * np.max(np.abs(x_true))
    if level == 0:
        if "lu" not in _SUPERLU:                    # (17 s on one host core: once for the parametrisations of this test)
            _SUPERLU["lu"] = spla.splu(K.tocsc(), permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.1, options=dict(SymmetricMode=True))
        lu = _SUPERLU["lu"]
        g = _rand_c(rng, n)
        gd = _dev(g)
        hx.check(hx.lib.hx_splu_solve(h, hx.ptr(gd), hx.ptr(xd), 0, hx.stream()))
        ref = lu.solve(g)
        assert np.max(np.abs(xd.cpu().numpy() - ref)) <= 1e-10 * np.max(np.abs(ref))
        hx.check(hx.lib.hx_splu_solve(h, hx.ptr(gd), hx.ptr(xd), 1, hx.stream()))
        ref = lu.solve(g, "T")
        assert np.max(np.abs(xd.cpu().numpy() - ref)) <= 1e-10 * np.max(np.abs(ref))
    hx.check(hx.lib.hx_splu_destroy(h))


@pytest.mark.parametrize("case", ["n1", "n2_swap", "diag", "n64", "n65", "n129_dense", "random_unsym", "two_components",
                                  "unsymmetric_pattern", "complex_symmetric", "symmetric_pattern_only"])
def test_device_sparse_lu_edge_cases(hx, case):
    """Sizes around the 64-column pivot block, a dense matrix (one supernode, several pivot blocks, left-looking updates), values
    without symmetry, a pattern without symmetry, disconnected graphs, a pivot that only exists after a row exchange inside the
    diagonal block -- against numpy / SciPy."""
    import torch
    rng = np.random.default_rng(7)
    if case == "n1":
        K = sp.csr_matrix(np.array([[2.0 - 3.0j]]))
    elif case == "n2_swap":
        K = sp.csr_matrix(np.array([[0.0, 2.0], [3.0 + 1j, 0.0]]))
    elif case == "diag":
        K = sp.diags(_rand_c(rng, 300) + 3.0).tocsr()
    elif case in ("n64", "n65"):
        n = int(case[1:])
        K = sp.csr_matrix(sp.random(n, n, density=0.2, random_state=rng) + 1j * sp.random(n, n, density=0.2, random_state=rng)
                          + sp.eye(n) * 2.0)
    elif case == "n129_dense":
        K = sp.csr_matrix(_rand_c(rng, 329, 329) + 10.0 * np.eye(329))
    elif case == "random_unsym":
        n = 1500
        P = sp.random(n, n, density=0.003, random_state=rng, format="csr")
        P = ((P + P.T) != 0).astype(np.float64)
        K = sp.csr_matrix(P.multiply(sp.csr_matrix(_rand_c(rng, n, 1)) @ sp.csr_matrix(np.ones((1, n)))) + sp.diags(4.0 + _rand_c(rng, n)))
        K.data = K.data * (1.0 + 0.3 * rng.standard_normal(K.nnz))          # values without any symmetry
    elif case in ("complex_symmetric", "symmetric_pattern_only"):
        n = 2500                                     # flags = 1 below: symmetric mode / its refusal when only the PATTERN is symmetric
        P = sp.random(n, n, density=0.002, random_state=rng, format="csr")
        P = sp.triu(P, 1)
        Pc = P + 1j * sp.triu(sp.random(n, n, density=0.002, random_state=rng, format="csr"), 1).multiply(P != 0)
        K = sp.csr_matrix(Pc + Pc.T + sp.diags(4.0 + _rand_c(rng, n)))
        if case == "symmetric_pattern_only":
            K.data = K.data * (1.0 + 0.2 * rng.standard_normal(K.nnz))
    elif case == "two_components":
        A = sp.random(500, 500, density=0.01, random_state=rng) + sp.eye(500) * 3
        K = sp.csr_matrix(sp.block_diag([A, 2j * A.T + sp.eye(500)]))
    else:
        K = sp.csr_matrix(sp.triu(sp.random(800, 800, density=0.01, random_state=rng)) + sp.eye(800) * 2 +
                          1j * sp.random(800, 800, density=0.001, random_state=rng))
    K = sp.csr_matrix(K, dtype=np.complex128)
    K.sort_indices()
    n = K.shape[0]
    rp, ci = K.indptr.astype(np.int32), K.indices.astype(np.int32)
    h = C.c_void_p()
    sym_flag = 1 if case in ("complex_symmetric", "symmetric_pattern_only", "diag") else 0
    hx.check(hx.lib.hx_splu_analyse(C.byref(h), n, hx.nptr(rp), hx.nptr(ci), None, 0, 0, sym_flag))
    val = _dev(np.ascontiguousarray(K.data))
    hx.check(hx.lib.hx_splu_factor(h, hx.ptr(val), hx.stream()))
    io, do = (C.c_int64 * 16)(), (C.c_double * 8)()
    hx.check(hx.lib.hx_splu_info(h, io, do))
    assert bool(do[5]) == (case in ("complex_symmetric", "diag")), (case, do[5])
    b = _rand_c(rng, n)
    bd = _dev(b)
    xd = torch.empty_like(bd)
    ref = np.linalg.solve(K.toarray(), b) if n <= 400 else spla.splu(K.tocsc()).solve(b)
    reft = np.linalg.solve(K.toarray().T, b) if n <= 400 else spla.splu(K.tocsc()).solve(b, "T")
    for tr, r in ((0, ref), (1, reft)):
        hx.check(hx.lib.hx_splu_solve(h, hx.ptr(bd), hx.ptr(xd), tr, hx.stream()))
        assert np.max(np.abs(xd.cpu().numpy() - r)) <= 1e-9 * np.max(np.abs(r)), (case, tr)
    hx.check(hx.lib.hx_splu_destroy(h))


def test_device_sparse_lu_random_stress():
    """150 random matrices through hx_splu_* (tools/r6/splu_fuzz.py: sizes 1 ... 7000, grid / band / random / dense patterns,
    symmetric or not in pattern and values, coordinates or not, leaf sizes, the symmetric-mode flag, A x = b and A^T x = b): every
    matrix with diagonally dominant rows is solved to a backward error below 1e-12 without a replaced pivot."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "r6", "splu_fuzz.py"), "150", "1000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "failures []" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("n,density", [(1, 1.0), (999, 0.02), (30000, 0.0006)])
def test_complex64_spmv_and_jacobi(hx, n, density):
    """Single-precision copies used by the AMG V-cycle: ragged rows, an empty row."""
    import torch
    rng = np.random.default_rng(n)
    M = sp.random(n, n, density=density, random_state=rng, format="csr") + 1j * sp.random(n, n, density=density, random_state=rng, format="csr")
    M = sp.csr_matrix(M + sp.diags(3.0 + rng.random(n)))
    if n > 10:
        M = sp.csr_matrix(sp.vstack([M[:5], sp.csr_matrix((1, n)), M[6:]]))       # one empty row
    M.sort_indices()
    rp, ci = _dev(M.indptr.astype(np.int32)), _dev(M.indices.astype(np.int32))
    va = _dev(M.data.astype(np.complex64))
    x = _rand_c(rng, n).astype(np.complex64)
    y0 = _rand_c(rng, n).astype(np.complex64)
    xd, yd = _dev(x), _dev(y0.copy())
    hx.check(hx.lib.hx_csr_spmv_c64(n, M.nnz, hx.ptr(rp), hx.ptr(ci), hx.ptr(va), hx.ptr(xd), hx.ptr(yd), -1.0, 1.0, hx.stream()))
    ref = y0.astype(np.complex128) - M @ x.astype(np.complex128)
    assert np.max(np.abs(yd.cpu().numpy() - ref)) <= 2e-5 * max(np.max(np.abs(ref)), 1.0)
    d = M.diagonal()
    d[d == 0] = 1.0
    dinv, b = _dev((1.0 / d).astype(np.complex64)), _dev(y0)
    out = torch.empty_like(xd)
    hx.check(hx.lib.hx_csr_jacobi_c64(n, M.nnz, hx.ptr(rp), hx.ptr(ci), hx.ptr(va), hx.ptr(dinv), hx.ptr(b), hx.ptr(xd),
                                      hx.ptr(out), 0.7, hx.stream()))
    refj = x + 0.7 * (1.0 / d) * (y0 - M @ x.astype(np.complex128))
    assert np.max(np.abs(out.cpu().numpy() - refj)) <= 2e-5 * max(np.max(np.abs(refj)), 1.0)


@pytest.mark.parametrize("n,density", [(1, 1.0), (63, 0.2), (64, 0.2), (1000, 0.02), (40000, 0.0005)])
def test_sell64_products_match_csr(hx, n, density):
    """SELL-64 companion: SpMV (c128, c64) and the smoother step against SciPy; ragged rows, an empty row, sizes around
    the slice height."""
    import torch
    rng = np.random.default_rng(n)
    M = sp.random(n, n, density=density, random_state=rng, format="csr") + 1j * sp.random(n, n, density=density, random_state=rng, format="csr")
    M = sp.csr_matrix(M + sp.diags(3.0 + rng.random(n)))
    if n > 10:
        M = sp.csr_matrix(sp.vstack([M[:5], sp.csr_matrix((1, n)), M[6:]]))
        M[7, :] = _rand_c(rng, n) * (rng.random(n) < 0.3)          # one long row
        M = sp.csr_matrix(M)
    M.sort_indices()
    rp, ci, va = _dev(M.indptr.astype(np.int32)), _dev(M.indices.astype(np.int32)), _dev(M.data.astype(np.complex128))
    h = C.c_void_p()
    hx.check(hx.lib.hx_sell_create(C.byref(h), n, hx.ptr(rp), hx.ptr(ci), hx.ptr(va), 1, 1, hx.stream()))
    ns, pad = C.c_int32(), C.c_int64()
    hx.check(hx.lib.hx_sell_info(h, C.byref(ns), C.byref(pad)))
    tail = C.c_int64()
    hx.check(hx.lib.hx_sell_tail_info(h, C.byref(tail), None))
    assert ns.value == (n + 63) // 64 and pad.value + tail.value >= M.nnz and pad.value % 64 == 0
    # index compression is decided per slice of 64 rows: count the slices that must stay on 32-bit columns
    coo = M.tocoo()
    far = np.zeros((n + 63) // 64, dtype=bool)
    far[(coo.row[np.abs(coo.row - coo.col) > 32767]) // 64] = True
    n_far = int(far.sum())                                                      # > 0 only for n = 40000
    expect_bits = 16 if n_far <= ((n + 63) // 64) // 2 else 32
    assert hx.lib.hx_sell_index_bits(h) == expect_bits
    assert hx.lib.hx_sell_wide_slices(h) == (n_far if expect_bits == 16 else (n + 63) // 64)
    x, y0 = _rand_c(rng, n), _rand_c(rng, n)
    xd, yd = _dev(x), _dev(y0.copy())
    a, b = 0.7 - 0.2j, -1.3 + 0.4j
    hx.check(hx.lib.hx_sell_spmv_z(h, hx.ptr(xd), hx.ptr(yd), hx.z2(a), hx.z2(b), hx.stream()))
    ref = a * (M @ x) + b * y0
    assert np.max(np.abs(yd.cpu().numpy() - ref)) <= 1e-13 * max(np.max(np.abs(ref)), 1.0)
    d = M.diagonal()
    d[d == 0] = 1.0
    dinv, bb, xprev = _dev(1.0 / d), _dev(y0), _dev(_rand_c(rng, n))
    out = torch.empty_like(xd)
    hx.check(hx.lib.hx_sell_smooth_z(h, hx.ptr(dinv), hx.ptr(bb), hx.ptr(xd), hx.ptr(out), 0.7, hx.ptr(xprev), 0.3, hx.stream()))
    refs = x + 0.7 * (1.0 / d) * (y0 - M @ x) + 0.3 * (x - xprev.cpu().numpy())
    assert np.max(np.abs(out.cpu().numpy() - refs)) <= 1e-13 * max(np.max(np.abs(refs)), 1.0)
    x32, y32 = _dev(x.astype(np.complex64)), _dev(y0.astype(np.complex64))
    hx.check(hx.lib.hx_sell_spmv_c64(h, hx.ptr(x32), hx.ptr(y32), -1.0, 1.0, hx.stream()))
    ref32 = y0 - M @ x
    assert np.max(np.abs(y32.cpu().numpy() - ref32)) <= 3e-5 * max(np.max(np.abs(ref32)), 1.0)
    d32, b32, o32 = _dev((1.0 / d).astype(np.complex64)), _dev(y0.astype(np.complex64)), torch.empty_like(x32)
    hx.check(hx.lib.hx_sell_smooth_c64(h, hx.ptr(d32), hx.ptr(b32), hx.ptr(x32), hx.ptr(o32), 0.7, None, 0.25, hx.stream()))
    refs32 = x + 0.7 * (1.0 / d) * (y0 - M @ x) + 0.25 * x
    assert np.max(np.abs(o32.cpu().numpy() - refs32)) <= 3e-5 * max(np.max(np.abs(refs32)), 1.0)
    hx.check(hx.lib.hx_sell_destroy(h))



@pytest.mark.parametrize("want,tails", [((1, 1), 1), ((1, 2), 1), ((0, 2), 1), ((1, 2), 0)])
def test_sell64_overflow_tails_of_ragged_slices(hx, want, tails):
    """Unstructured row lengths (14.5 +- 1.8 entries, as on tetrahedra): slices keep a body of 8 / 16 entries per row and the longer
    rows' remaining entries in a packed tail that the same kernel adds (hx_sell_tail_info); products, residual and smoother
    step in every value type agree with SciPy exactly as without tails; the stored entries shrink."""
    import torch
    rng = np.random.default_rng(11)
    n = 20_000 + 37
    lens = np.clip(np.rint(rng.normal(13.5, 1.8, n)).astype(np.int64), 8, 22)
    lens[100:164] = 15                                   # one uniform slice: no tail there
    lens[300] = 90                                       # one very long row: its slice keeps its width or a long tail
    rows = np.repeat(np.arange(n), lens)
    cols = (rows + rng.integers(-400, 400, rows.size)) % n
    M = sp.csr_matrix((_rand_c(rng, rows.size), (rows, cols)), shape=(n, n))
    M = sp.csr_matrix(M + sp.diags(6.0 + rng.random(n)))
    M.sort_indices()
    rp, ci, va = _dev(M.indptr.astype(np.int32)), _dev(M.indices.astype(np.int32)), _dev(M.data.astype(np.complex128))
    hx.check(hx.lib.hx_set_option(b"sell_tail", str(tails).encode()))
    try:
        h = C.c_void_p()
        hx.check(hx.lib.hx_sell_create(C.byref(h), n, hx.ptr(rp), hx.ptr(ci), hx.ptr(va), want[0], want[1], hx.stream()))
    finally:
        hx.check(hx.lib.hx_set_option(b"sell_tail", None))
    ns, pad, tail, split = C.c_int32(), C.c_int64(), C.c_int64(), C.c_int32()
    hx.check(hx.lib.hx_sell_info(h, C.byref(ns), C.byref(pad)))
    hx.check(hx.lib.hx_sell_tail_info(h, C.byref(tail), C.byref(split)))
    row_len = np.diff(M.indptr)
    full = 64 * int(sum(row_len[i:i + 64].max() for i in range(0, n, 64)))       # every slice as wide as its longest row
    if tails:
        assert split.value > 0.8 * ns.value and 0 < tail.value < 0.1 * M.nnz
        assert pad.value + tail.value >= M.nnz and pad.value + tail.value < 0.92 * full
    else:
        assert split.value == 0 and tail.value == 0 and pad.value == full
    x, y0 = _rand_c(rng, n), _rand_c(rng, n)
    d = M.diagonal()
    if want[0]:
        xd, yd = _dev(x), _dev(y0.copy())
        a, b = 0.7 - 0.2j, -1.3 + 0.4j
        hx.check(hx.lib.hx_sell_spmv_z(h, hx.ptr(xd), hx.ptr(yd), hx.z2(a), hx.z2(b), hx.stream()))
        ref = a * (M @ x) + b * y0
        assert np.max(np.abs(yd.cpu().numpy() - ref)) <= 1e-13 * np.max(np.abs(ref))
        dinv, bb, xprev = _dev(1.0 / d), _dev(y0), _dev(_rand_c(rng, n))
        out = torch.empty_like(xd)
        hx.check(hx.lib.hx_sell_smooth_z(h, hx.ptr(dinv), hx.ptr(bb), hx.ptr(xd), hx.ptr(out), 0.7, hx.ptr(xprev), 0.3, hx.stream()))
        refs = x + 0.7 * (1.0 / d) * (y0 - M @ x) + 0.3 * (x - xprev.cpu().numpy())
        assert np.max(np.abs(out.cpu().numpy() - refs)) <= 1e-13 * np.max(np.abs(refs))
    tol = 3e-5 if want[1] == 1 else 2e-3                 # complex64 values / complex-half storage of the values
    x32, y32 = _dev(x.astype(np.complex64)), _dev(y0.astype(np.complex64))
    hx.check(hx.lib.hx_sell_spmv_c64(h, hx.ptr(x32), hx.ptr(y32), -1.0, 1.0, hx.stream()))
    ref32 = y0 - M @ x
    assert np.max(np.abs(y32.cpu().numpy() - ref32)) <= tol * np.max(np.abs(ref32))
    d32, b32, o32 = _dev((1.0 / d).astype(np.complex64)), _dev(y0.astype(np.complex64)), torch.empty_like(x32)
    hx.check(hx.lib.hx_sell_smooth_c64(h, hx.ptr(d32), hx.ptr(b32), hx.ptr(x32), hx.ptr(o32), 0.7, None, 0.25, hx.stream()))
    refs32 = x + 0.7 * (1.0 / d) * (y0 - M @ x) + 0.25 * x
    assert np.max(np.abs(o32.cpu().numpy() - refs32)) <= tol * np.max(np.abs(refs32))
    hx.check(hx.lib.hx_sell_destroy(h))


def test_sell64_random_stress():
    """120 random operators through hx_sell_* (tools/r6/sell_fuzz.py: six row-length distributions incl. rows far beyond any body, banded
    and far columns, rectangular operators, the three value types, random parameters of the body / tail split): operator product, block
    product, residual and smoother step agree with SciPy."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "r6", "sell_fuzz.py"), "120", "5000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "failures []" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_device_buffer_pool_reuses_released_blocks(hx):
    """csrc/pool.hip: the large buffers a destroyed SELL-64 operator releases are handed to the next operator of the same shape
    without a driver call (every fixed-point call of a driver builds the same objects again); the product computed in reused
    memory is right; hx_pool_trim gives everything back."""
    rng = np.random.default_rng(3)
    n = 200_000
    M = sp.diags([1.0 + 0.5j, 4.0 - 1.0j, 0.7 + 0.2j], [-1, 0, 1], shape=(n, n), format="csr", dtype=np.complex128)
    M.sort_indices()
    rp, ci, va = _dev(M.indptr.astype(np.int32)), _dev(M.indices.astype(np.int32)), _dev(M.data.astype(np.complex128))
    x = _rand_c(rng, n)
    xd, yd = _dev(x), _dev(np.zeros(n, dtype=np.complex128))

    def stats(reset=0):
        a, b, c, e = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int64(0)
        d = C.c_double(0.0)
        hx.check(hx.lib.hx_pool_stats(C.byref(a), C.byref(b), C.byref(c), C.byref(d), C.byref(e), reset))
        return a.value, b.value, c.value, e.value

    hx.check(hx.lib.hx_pool_trim())
    stats(1)
    for trip in range(2):
        h = C.c_void_p()
        hx.check(hx.lib.hx_sell_create(C.byref(h), n, hx.ptr(rp), hx.ptr(ci), hx.ptr(va), 1, 1, hx.stream()))
        hx.check(hx.lib.hx_sell_spmv_z(h, hx.ptr(xd), hx.ptr(yd), hx.z2(1.0), hx.z2(0.0), hx.stream()))
        assert np.max(np.abs(yd.cpu().numpy() - M @ x)) <= 1e-13 * np.max(np.abs(x)) * 6
        hx.check(hx.lib.hx_sell_destroy(h))
        requests, reused, driver, cached = stats()
        if trip == 0:
            assert reused == 0 and cached >= 16 * M.nnz            # (the complex128 value array alone)
            first = requests
        else:
            assert reused > 0 and requests == 2 * first and cached >= 16 * M.nnz
    hx.check(hx.lib.hx_pool_trim())
    assert stats()[3] == 0

@pytest.mark.parametrize("want", [(1, 0), (1, 1), (0, 2)])
def test_sell64_stencil_slices(hx, want):
    """Slices whose rows share one list of column deltas are stored without per-entry indices (structured numberings):
    a 3-D 7-point operator with complex variable coefficients (boundary rows lack neighbours: zeros in the shared list),
    some slices made irregular by extra entries (those keep their indices), a first/last row wrap.  Products must equal
    the CSR ones in every value format, and the detection must find exactly the regular slices."""
    import torch
    nx, ny, nz = 23, 17, 19
    n = nx * ny * nz
    rng = np.random.default_rng(5)
    idx = np.arange(n).reshape(nz, ny, nx)
    rows, cols = [np.arange(n)], [np.arange(n)]
    for axis, off in ((2, 1), (1, nx), (0, nx * ny)):
        a = np.take(idx, np.arange(idx.shape[axis] - 1), axis=axis).ravel()
        rows += [a, a + off]
        cols += [a + off, a]
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    vals = _rand_c(rng, rows.size) + 6.0 * (rows == cols)
    M = sp.csr_matrix((vals, (rows, cols)), shape=(n, n))
    irregular = np.arange(20 * 64, 23 * 64)                       # three slices get a random extra column each row
    extra = sp.csr_matrix((_rand_c(rng, irregular.size), (irregular, rng.integers(0, n, irregular.size))), shape=(n, n))
    M = sp.csr_matrix(M + extra)
    M.sort_indices()
    n_slices = (n + 63) // 64
    # expected: a slice is a stencil slice iff the union of its rows' deltas is at most width + width // 8 long
    expect = 0
    for sl in range(n_slices):
        r0, r1 = sl * 64, min(n, sl * 64 + 64)
        width = int(np.diff(M.indptr[r0:r1 + 1]).max())
        deltas = set()
        for r in range(r0, r1):
            deltas.update((M.indices[M.indptr[r]:M.indptr[r + 1]] - r).tolist())
        expect += int(len(deltas) <= min(64, width + width // 8))
    rp, ci, va = _dev(M.indptr.astype(np.int32)), _dev(M.indices.astype(np.int32)), _dev(M.data.astype(np.complex128))
    h = C.c_void_p()
    hx.check(hx.lib.hx_sell_create(C.byref(h), n, hx.ptr(rp), hx.ptr(ci), hx.ptr(va), want[0], want[1], hx.stream()))
    got = hx.lib.hx_sell_stencil_slices(h)
    assert got == expect and 0 < got < n_slices
    x, y0 = _rand_c(rng, n), _rand_c(rng, n)
    if want[0]:
        xd, yd = _dev(x), _dev(y0.copy())
        hx.check(hx.lib.hx_sell_spmv_z(h, hx.ptr(xd), hx.ptr(yd), hx.z2(0.5 + 1j), hx.z2(-2.0), hx.stream()))
        ref = (0.5 + 1j) * (M @ x) - 2.0 * y0
        assert np.max(np.abs(yd.cpu().numpy() - ref)) <= 1e-13 * np.max(np.abs(ref))
    if want[1]:
        x32, y32 = _dev(x.astype(np.complex64)), _dev(y0.astype(np.complex64))
        hx.check(hx.lib.hx_sell_spmv_c64(h, hx.ptr(x32), hx.ptr(y32), -1.0, 1.0, hx.stream()))
        ref32 = y0 - M @ x
        tol = 3e-5 if want[1] == 1 else 2e-3                  # complex-half storage rounds the entries to 11 bits
        assert np.max(np.abs(y32.cpu().numpy() - ref32)) <= tol * np.max(np.abs(ref32))
        d = M.diagonal()
        d32, o32 = _dev((1.0 / d).astype(np.complex64)), torch.empty_like(x32)
        hx.check(hx.lib.hx_sell_smooth_c64(h, hx.ptr(d32), hx.ptr(y32), hx.ptr(x32), hx.ptr(o32), 0.6, None, 0.0, hx.stream()))
        refs = x + 0.6 * (1.0 / d) * (y32.cpu().numpy() - M @ x)
        assert np.max(np.abs(o32.cpu().numpy() - refs)) <= tol * np.max(np.abs(refs))
    hx.check(hx.lib.hx_sell_destroy(h))


def test_sell64_stencil_slices_of_a_row_sharded_local_operator(hx):
    """Rows [r0, r1) of a 3-D 7-point operator over [owned | ghost] columns, as ShardContext.localize builds them: the
    ghost columns come numbered behind the owned ones whatever their global position, so a row's column list -- and with
    it the slice's delta list -- is NOT ascending (lower-plane ghosts first).  Stencil slices must still be found, and the
    products (gather form for complex128, LDS-staged windows for the complex-half copy) must equal the CSR ones."""
    import torch
    nx, ny, nz = 16, 16, 24
    n = nx * ny * nz
    rng = np.random.default_rng(11)
    idx = np.arange(n).reshape(nz, ny, nx)
    rows, cols = [np.arange(n)], [np.arange(n)]
    for axis, off in ((2, 1), (1, nx), (0, nx * ny)):
        a = np.take(idx, np.arange(idx.shape[axis] - 1), axis=axis).ravel()
        rows += [a, a + off]
        cols += [a + off, a]
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    vals = _rand_c(rng, rows.size) + 6.0 * (rows == cols)
    M = sp.csr_matrix((vals, (rows, cols)), shape=(n, n))
    M.sort_indices()
    r0, r1 = 8 * nx * ny, 16 * nx * ny
    L = M[r0:r1].tocsr()
    L.sort_indices()
    gcol = L.indices
    own = (gcol >= r0) & (gcol < r1)
    ghost = np.unique(gcol[~own])
    lcol = np.where(own, gcol - r0, (r1 - r0) + np.searchsorted(ghost, gcol)).astype(np.int32)   # global order kept: unsorted rows
    n_own, n_ext = r1 - r0, r1 - r0 + ghost.size
    Lloc = sp.csr_matrix((L.data, lcol, L.indptr), shape=(n_own, n_ext))
    rp, ci, va = _dev(L.indptr.astype(np.int32)), _dev(lcol), _dev(L.data.astype(np.complex128))
    h = C.c_void_p()
    hx.check(hx.lib.hx_sell_create(C.byref(h), n_own, hx.ptr(rp), hx.ptr(ci), hx.ptr(va), 1, 2, hx.stream()))
    assert hx.lib.hx_sell_stencil_slices(h) >= (n_own // 64) // 2
    x, y0 = _rand_c(rng, n_ext), _rand_c(rng, n_own)
    xd, yd = _dev(x), _dev(y0.copy())
    hx.check(hx.lib.hx_sell_spmv_z(h, hx.ptr(xd), hx.ptr(yd), hx.z2(1.0), hx.z2(0.0), hx.stream()))
    ref = Lloc @ x
    assert np.max(np.abs(yd.cpu().numpy() - ref)) <= 1e-13 * np.max(np.abs(ref))
    x32, y32 = _dev(x.astype(np.complex64)), _dev(y0.astype(np.complex64))
    hx.check(hx.lib.hx_sell_spmv_c64(h, hx.ptr(x32), hx.ptr(y32), -1.0, 1.0, hx.stream()))
    ref32 = y0 - ref
    assert np.max(np.abs(y32.cpu().numpy() - ref32)) <= 2e-3 * np.max(np.abs(ref32))
    d = M.diagonal()[r0:r1]
    xin = _dev(x.astype(np.complex64))
    d32, o32 = _dev((1.0 / d).astype(np.complex64)), torch.zeros(n_ext, dtype=torch.complex64, device=xin.device)
    hx.check(hx.lib.hx_sell_smooth_c64(h, hx.ptr(d32), hx.ptr(y32), hx.ptr(xin), hx.ptr(o32), 0.6, None, 0.0, hx.stream()))
    refs = x[:n_own] + 0.6 * (1.0 / d) * (y32.cpu().numpy() - ref)
    assert np.max(np.abs(o32.cpu().numpy()[:n_own] - refs)) <= 2e-3 * np.max(np.abs(refs))
    hx.check(hx.lib.hx_sell_destroy(h))


@pytest.mark.parametrize("n,scale", [(64, 1.0), (5000, 3.0e6), (5000, 2.0e-7)])
def test_sell64_complex_half_storage(hx, n, scale):
    """want_c64 = 2: values stored as scaled complex-half, arithmetic in complex64.  The products must equal those of the
    matrix whose entries were rounded to fp16 relative to the largest component (no overflow / underflow for entries
    seven orders of magnitude apart in absolute size: the handle rescales), to complex64 accuracy."""
    import torch
    rng = np.random.default_rng(n)
    M = sp.random(n, n, density=min(1.0, 12.0 / n), random_state=rng, format="csr") \
        + 1j * sp.random(n, n, density=min(1.0, 12.0 / n), random_state=rng, format="csr")
    M = sp.csr_matrix((M + sp.diags(3.0 + rng.random(n))) * scale)
    M.sort_indices()
    rp, ci, va = _dev(M.indptr.astype(np.int32)), _dev(M.indices.astype(np.int32)), _dev(M.data.astype(np.complex128))
    h = C.c_void_p()
    hx.check(hx.lib.hx_sell_create(C.byref(h), n, hx.ptr(rp), hx.ptr(ci), hx.ptr(va), 0, 2, hx.stream()))
    vmax = max(np.abs(M.data.real).max(), np.abs(M.data.imag).max())
    s = np.float32(vmax) / np.float32(16384.0)
    q = (M.data.real / s).astype(np.float16).astype(np.float64) * s + 1j * (M.data.imag / s).astype(np.float16).astype(np.float64) * s
    Mq = sp.csr_matrix((q, M.indices, M.indptr), shape=M.shape)
    assert np.isfinite(q).all() and abs(Mq - M).max() <= 1e-3 * vmax
    x, y0 = _rand_c(rng, n), _rand_c(rng, n) * scale
    x32, y32 = _dev(x.astype(np.complex64)), _dev(y0.astype(np.complex64))
    hx.check(hx.lib.hx_sell_spmv_c64(h, hx.ptr(x32), hx.ptr(y32), -1.0, 1.0, hx.stream()))
    ref = y0 - Mq @ x
    flip = 1e-3 * vmax * np.max(np.abs(x))     # one entry per row may round to the neighbouring fp16 (device multiplies by 1/s)
    assert np.max(np.abs(y32.cpu().numpy() - ref)) <= 3e-5 * np.max(np.abs(ref)) + flip
    d = M.diagonal()
    d32, b32, o32 = _dev((1.0 / d).astype(np.complex64)), _dev(y0.astype(np.complex64)), torch.empty_like(x32)
    hx.check(hx.lib.hx_sell_smooth_c64(h, hx.ptr(d32), hx.ptr(b32), hx.ptr(x32), hx.ptr(o32), 0.7, None, 0.25, hx.stream()))
    refs = x + 0.7 * (1.0 / d) * (y0 - Mq @ x) + 0.25 * x
    assert np.max(np.abs(o32.cpu().numpy() - refs)) <= 3e-5 * np.max(np.abs(refs)) + flip * np.max(np.abs(1.0 / d))
    hx.check(hx.lib.hx_sell_destroy(h))


def test_in_situ_spmv_profile_counts_launches(hx):
    """hx_profile_sell_spmv: HIP events around every hx_sell_spmv_z launch on operators of at least min_rows rows."""
    import torch
    n = 4096
    rng = np.random.default_rng(2)
    M = sp.random(n, n, density=8.0 / n, random_state=rng, format="csr") + sp.eye(n, format="csr")
    M = sp.csr_matrix(M, dtype=np.complex128)
    M.sort_indices()
    rp, ci, va = _dev(M.indptr.astype(np.int32)), _dev(M.indices.astype(np.int32)), _dev(M.data)
    h = C.c_void_p()
    hx.check(hx.lib.hx_sell_create(C.byref(h), n, hx.ptr(rp), hx.ptr(ci), hx.ptr(va), 1, 0, hx.stream()))
    x = _dev(_rand_c(rng, n))
    y = torch.empty_like(x)
    one, zero = hx.z2(1.0), hx.z2(0.0)
    cnt, ms = C.c_int64(-1), C.c_double(-1.0)
    hx.check(hx.lib.hx_profile_sell_spmv(1, n + 1))            # operator too small: nothing recorded
    hx.check(hx.lib.hx_sell_spmv_z(h, hx.ptr(x), hx.ptr(y), one, zero, hx.stream()))
    hx.check(hx.lib.hx_profile_sell_spmv_read(C.byref(cnt), C.byref(ms)))
    assert cnt.value == 0 and ms.value == 0.0
    hx.check(hx.lib.hx_profile_sell_spmv(1, n))
    for _ in range(7):
        hx.check(hx.lib.hx_sell_spmv_z(h, hx.ptr(x), hx.ptr(y), one, zero, hx.stream()))
    hx.check(hx.lib.hx_profile_sell_spmv_read(C.byref(cnt), C.byref(ms)))
    assert cnt.value == 7 and 0.0 < ms.value < 50.0
    # the smoother step (kind 1) is SAMPLED: every 8th launch carries an event pair (launches 0, 8, 16 of 20)
    h2 = C.c_void_p()                                           # (a handle that carries the complex64 copy of the values)
    hx.check(hx.lib.hx_sell_create(C.byref(h2), n, hx.ptr(rp), hx.ptr(ci), hx.ptr(va), 0, 1, hx.stream()))
    d32 = _dev((1.0 / M.diagonal()).astype(np.complex64))
    b32, x32 = _dev(_rand_c(rng, n).astype(np.complex64)), _dev(_rand_c(rng, n).astype(np.complex64))
    o32 = torch.empty_like(x32)
    for _ in range(20):
        hx.check(hx.lib.hx_sell_smooth_c64(h2, hx.ptr(d32), hx.ptr(b32), hx.ptr(x32), hx.ptr(o32), 0.6, None, 0.0, hx.stream()))
    hx.check(hx.lib.hx_profile_sell_read(1, C.byref(cnt), C.byref(ms)))
    assert cnt.value == 3 and 0.0 < ms.value < 50.0 and hx.lib.hx_profile_sell_skipped(1) == 0
    hx.check(hx.lib.hx_sell_destroy(h2))
    hx.check(hx.lib.hx_profile_sell_spmv_read(C.byref(cnt), C.byref(ms)))
    assert cnt.value == 7                                       # (the product's records are its own)
    hx.check(hx.lib.hx_profile_sell_spmv(0, 0))
    hx.check(hx.lib.hx_sell_spmv_z(h, hx.ptr(x), hx.ptr(y), one, zero, hx.stream()))
    hx.check(hx.lib.hx_profile_sell_spmv_read(C.byref(cnt), C.byref(ms)))
    assert cnt.value == 7                                       # disabled: no new records
    assert np.max(np.abs(y.cpu().numpy() - M @ x.cpu().numpy())) < 1e-12
    hx.check(hx.lib.hx_sell_destroy(h))


# ------------------------------------------------------------------------------------------------ device AMG set-up
def _dcsr(M, dtype=None):
    """scipy CSR -> helmholtz_x.amg.DevCSR on the device (values complex128 or float64)."""
    from helmholtz_x.amg import DevCSR
    M = sp.csr_matrix(M)
    M.sort_indices()
    val = M.data.astype(dtype if dtype is not None else (np.complex128 if np.iscomplexobj(M.data) else np.float64))
    return DevCSR(M.shape, _dev(M.indptr.astype(np.int32)), _dev(M.indices.astype(np.int32)), _dev(val))


def _same_csr(got, ref, tol=1e-12):
    """Pattern bit-exact (sorted rows), values to a relative tolerance."""
    ref = sp.csr_matrix(ref)
    ref.sort_indices()
    g = got.to_scipy()
    assert g.shape == ref.shape
    assert np.array_equal(g.indptr, ref.indptr) and np.array_equal(g.indices, ref.indices)
    assert np.abs(g.data - ref.data).max() <= tol * max(np.abs(ref.data).max(), 1e-300)


def test_device_amg_setup_passes_match_host(hx):
    """amg_setup.hip against the host passes of round 1 (themselves pinned to NumPy in test_host_logic.py): strength
    filter, power iteration, smoothed prolongator, transpose, sparse products, and the whole hierarchy."""
    import torch
    from helmholtz_x import amg
    from oracle import fem, meshes
    m = meshes.box(9, 7, 8, 0.4, 0.1, 0.25) if hasattr(meshes, "box") else None
    rng = np.random.default_rng(3)
    if m is None:
        m = meshes.rectangle(40, 12, 0.4, 0.1)
    x = m.x.copy()
    x[:, 1] *= 3.0                                          # anisotropy => weak couplings to lump
    S = fem.assemble_A(x, m.cells, 300.0 + 50 * rng.random(x.shape[0])).real.tocsr()       # negative semi-definite
    n = S.shape[0]
    K = (S + (40.0 + 6j) * fem.assemble_C(x, m.cells)).tocsr().astype(np.complex128)
    Kd = _dcsr(K)
    # diagonal
    diag = torch.empty(n, dtype=torch.complex128, device=Kd.val.device)
    hx.check(hx.lib.hx_csr_diagonal_z(n, hx.ptr(Kd.rowptr), hx.ptr(Kd.col), hx.ptr(Kd.val), hx.ptr(diag), hx.stream()))
    assert np.array_equal(diag.cpu().numpy(), K.diagonal())
    # strength filter
    theta = 0.25
    sd = torch.empty(n, dtype=torch.float64, device=Kd.val.device)
    Lf_d = amg._dev_filter(Kd, -1.0, theta, sd)
    Lf_h = amg._filter(K, -1.0, theta)
    assert Lf_h.nnz < K.nnz                                  # something was lumped
    _same_csr(Lf_d, Lf_h, 1e-14)
    # rho by power iteration (same start vector on both sides is not available: compare with the dense spectrum)
    v = torch.randn(n, dtype=torch.float64, device=Kd.val.device)
    w = torch.empty_like(v)
    rho = C.c_double(0.0)
    hx.check(hx.lib.hx_sa_power_rho(n, hx.ptr(Lf_d.rowptr), hx.ptr(Lf_d.col), hx.ptr(Lf_d.val), hx.ptr(v), hx.ptr(w), 60,
                                    C.byref(rho), hx.stream()))
    exact = np.max(np.abs(np.linalg.eigvals((sp.diags(1 / Lf_h.diagonal()) @ Lf_h).toarray())))
    assert abs(rho.value - exact) < 2e-2 * exact
    # aggregation (host pass in both set-ups) + smoothed prolongator
    agg, na = amg._dev_aggregate(Lf_d)
    agg_h, na_h = amg._aggregate(Lf_h)
    assert na == na_h and np.array_equal(agg, agg_h)
    isc = 1.0 / np.sqrt(np.bincount(agg, minlength=na).astype(np.float64))
    prp = torch.empty(n + 1, dtype=torch.int32, device=Kd.val.device)
    pnnz = C.c_int64(0)
    agg_d = _dev(agg)
    hx.check(hx.lib.hx_sa_prolongator_count(n, hx.ptr(Lf_d.rowptr), hx.ptr(Lf_d.col), hx.ptr(agg_d), hx.ptr(prp),
                                            C.byref(pnnz), hx.stream()))
    pci = torch.empty(pnnz.value, dtype=torch.int32, device=Kd.val.device)
    pv = torch.empty(pnnz.value, dtype=torch.float64, device=Kd.val.device)
    isc_d = _dev(isc)
    hx.check(hx.lib.hx_sa_prolongator_fill(n, hx.ptr(Lf_d.rowptr), hx.ptr(Lf_d.col), hx.ptr(Lf_d.val), hx.ptr(agg_d),
                                           hx.ptr(isc_d), 0.6, hx.ptr(prp), hx.ptr(pci), hx.ptr(pv), hx.stream()))
    P_d = amg.DevCSR((n, na), prp, pci, pv)
    P_h = amg._prolongator(Lf_h, agg, na, 0.6)
    _same_csr(P_d, P_h, 1e-14)
    # transpose and the two products of the Galerkin operator
    R_d = amg._dev_transpose(P_d, na)
    _same_csr(R_d, P_d.to_scipy().T.tocsr(), 0.0)
    T_d = amg._dev_spgemm(Kd, P_d)
    _same_csr(T_d, K @ P_h, 1e-13)
    Kc_d = amg._dev_spgemm(R_d, T_d)
    Kc_h = amg._galerkin(P_h.T.tocsr(), K, P_h)
    _same_csr(Kc_d, Kc_h, 1e-12)
    # whole hierarchy
    lv_d = amg.build_hierarchy_device(Kd, coarse_size=40)
    lv_h = amg.build_hierarchy(K, coarse_size=40)
    # rho comes from different start vectors, so transfer weights differ by O(1e-2) and a coupling next to the strength
    # threshold may flip on the coarser levels: level 0 -> 1 must agree exactly, the rest in size
    assert len(lv_d) == len(lv_h) >= 3 and lv_d[1]["n"] == lv_h[1]["n"]
    assert all(abs(a["n"] - b["n"]) <= 0.05 * b["n"] + 2 for a, b in zip(lv_d, lv_h))
    g, h = lv_d[1]["K"].to_scipy(), sp.csr_matrix(lv_h[1]["K"])
    h.sort_indices()
    assert np.array_equal(g.indptr, h.indptr) and np.array_equal(g.indices, h.indices)
    for a, b in zip(lv_d, lv_d[1:]):
        Kc = (a["R"].to_scipy() @ a["K"].to_scipy() @ a["P"].to_scipy()).tocsr()
        assert abs(b["K"].to_scipy() - Kc).max() < 1e-11 * abs(Kc).max()              # Galerkin, level by level


@pytest.mark.parametrize("m,k,n,da,db", [(1, 1, 1, 1.0, 1.0), (300, 200, 250, 0.02, 0.03), (64, 3000, 40, 0.3, 0.5),
                                         (2000, 2000, 2000, 0.004, 0.004)])
def test_device_spgemm_and_transpose_random(hx, m, k, n, da, db):
    """Sparse products on random rectangular operands: empty rows, rows denser than the hash-table floor, every
    real/complex combination; transposes with empty columns."""
    from helmholtz_x import amg
    rng = np.random.default_rng(m + 7 * n)
    A = sp.random(m, k, density=da, random_state=rng, format="csr")
    B = sp.random(k, n, density=db, random_state=rng, format="csr")
    if A.nnz == 0:
        A = sp.csr_matrix(np.ones((m, k)))
    if B.nnz == 0:
        B = sp.csr_matrix(np.ones((k, n)))
    Az = A + 1j * A.multiply(A)
    Bz = B - 0.5j * B
    for X, Y in ((A, B), (Az, B), (A, Bz), (Az, Bz)):
        ref = (sp.csr_matrix(X) @ sp.csr_matrix(Y)).tocsr()
        if ref.nnz == 0:
            continue
        _same_csr(amg._dev_spgemm(_dcsr(X), _dcsr(Y)), ref, 1e-13)
    _same_csr(amg._dev_transpose(_dcsr(A), k), A.T.tocsr(), 0.0)


def test_bloch_triple_product_on_the_device_equals_scipy(monkeypatch):
    """Blochifier.blochify (reference: bloch_operator.py:104-111, scipy CSR x CSR x CSR on the host): the device route
    NB (M BN) through hx_spgemm_* gives the same reduced operators -- pattern bit-exact, values to rounding -- for the real
    symmetric volume operators and for the complex boundary operator of a Robin sector."""
    import scipy.sparse as sp
    from helmholtz_x.acoustic_matrices import AcousticMatrices
    from helmholtz_x.bloch_operator import Blochifier
    from helmholtz_x.synthetic import annulus, micca_sound_speed
    half = np.pi / 16

    class Geometry:
        pass

    geo = Geometry()
    geo.mesh, _, geo.facet_tags = annulus(5, 4, 7, n_flames=1, sector=(-half, half))
    bcs = {11: {'Robin': -0.875 - 0.2j}, 12: 'Master', 13: 'Slave'}
    M = AcousticMatrices(geo.mesh, geo.facet_tags, bcs, micca_sound_speed(geo.mesh))
    dev = Blochifier(geo, bcs, 16, M)
    monkeypatch.setenv("HX_BLOCH_HOST", "1")
    host = Blochifier(geo, bcs, 16, M)
    for name in ("A", "B", "B_adj", "C"):
        a, b = getattr(dev, name).to_scipy(), getattr(host, name).to_scipy()
        a.sort_indices()
        b.sort_indices()
        b.eliminate_zeros()          # (scipy keeps the explicit zeros of cancelled sums; the device product stores the same slots)
        a2 = sp.csr_matrix(a)
        a2.eliminate_zeros()
        assert a2.shape == b.shape and np.array_equal(a2.indptr, b.indptr) and np.array_equal(a2.indices, b.indices), name
        assert np.max(np.abs(a2.data - b.data)) <= 1e-13 * np.max(np.abs(b.data)), name


@pytest.mark.parametrize("degree", [1, 2])
def test_shape_derivative_functional_kernel(degree):
    """hx_shape_derivative_facets (shape_derivatives.py:12-37: the boundary integral of the Hadamard shape derivative) against
    the analytic patch value and, with random complex fields on a distorted mesh, against oracle/shape.py."""
    from helmholtz_x.fem import Function, FunctionSpace, MeshTags
    from helmholtz_x.shape_derivatives import boundary_functional
    from oracle import shape
    from test_oracle_goldens import shape_patch_case
    mesh, fids, V, p, q, c, exact, cell_dofs = shape_patch_case(degree, jitter=0.04)
    tags = MeshTags(mesh, 2, fids, np.full(len(fids), 6))
    cfun = Function(FunctionSpace(mesh, "CG", 1), "c")
    cfun.x.array[:] = c
    got = boundary_functional(mesh, tags, 6, V, p, q, cfun, degree=degree)
    assert abs(got - exact) < 1e-11 * abs(exact), (got, exact)
    rng = np.random.default_rng(11)
    nd = p.size
    p2, q2 = _rand_c(rng, nd), _rand_c(rng, nd)
    c2 = 300.0 + 50 * rng.random(mesh.num_vertices) + 5j * rng.random(mesh.num_vertices)
    V2 = rng.standard_normal((mesh.num_vertices, 3))
    cfun.x.array[:] = c2
    ref = shape.boundary_functional(mesh.geometry.x, mesh.topology.cells, mesh.facets[fids], mesh.facet_cells[fids], V2, p2, q2, c2,
                                    degree=degree, cell_dofs=cell_dofs)
    got = boundary_functional(mesh, tags, 6, V2, p2, q2, cfun, degree=degree)
    assert abs(got - ref) < 1e-12 * abs(ref), (got, ref)
    # DG0 sound speed (the MICCA parameter files give c per cell): grad c = 0 inside the cell
    cd = Function(FunctionSpace(mesh, "DG", 0), "c")
    cd.x.array[:] = 300.0 + 40 * rng.random(mesh.num_cells)
    ref = shape.boundary_functional(mesh.geometry.x, mesh.topology.cells, mesh.facets[fids], mesh.facet_cells[fids], V2, p2, q2,
                                    cd.x.array, degree=degree, cell_dofs=cell_dofs, c_is_cellwise=True)
    got = boundary_functional(mesh, tags, 6, V2, p2, q2, cd, degree=degree)
    assert abs(got - ref) < 1e-12 * abs(ref)


def test_small_dense_helpers_of_the_solve_path():
    """hx_zsmall_gemv, hx_zroll, hx_zdense_inverse (round 4: the r x r Woodbury product, the translated Woodbury column and the
    coarsest-level inverse were torch calls) against NumPy; the inverse with a matrix that NEEDS its row pivoting, and a
    singular one is reported."""
    import torch
    from helmholtz_x import _hx
    from helmholtz_x.petsc import to_device
    rng = np.random.default_rng(3)
    for m, n in ((16, 16), (1, 5), (64, 64), (7, 3)):
        A, x = _rand_c(rng, m * n).reshape(m, n), _rand_c(rng, n)
        y = torch.empty(m, dtype=torch.complex128, device="cuda")
        Ad, xd = to_device(A), to_device(x)           # (named: the device copies must outlive the launch)
        _hx.check(_hx.lib.hx_zsmall_gemv(m, n, _hx.ptr(Ad), _hx.ptr(xd), _hx.ptr(y), _hx.stream()))
        assert _rel(y.cpu().numpy(), A @ x) < 1e-14
    assert _hx.lib.hx_zsmall_gemv(65, 4, None, None, None, _hx.stream()) != 0
    for n, shift in ((1000, 17), (1000, -17), (5, 12), (64, 0)):
        x = _rand_c(rng, n)
        dst = torch.empty(n, dtype=torch.complex128, device="cuda")
        xd = to_device(x)
        _hx.check(_hx.lib.hx_zroll(n, shift, _hx.ptr(xd), _hx.ptr(dst), _hx.stream()))
        assert np.array_equal(dst.cpu().numpy(), np.roll(x, shift))
    for n in (1, 2, 37, 300, 1000):
        A = _rand_c(rng, n * n).reshape(n, n) + 0.1 * np.eye(n)
        if n > 2:
            A[0, 0] = 0.0                     # a zero on the diagonal: elimination without row exchanges would stop here
            A[[1, n - 1]] = A[[n - 1, 1]]
        d = to_device(A.copy())
        _hx.check(_hx.lib.hx_zdense_inverse(n, _hx.ptr(d), _hx.stream()))
        inv = d.cpu().numpy()
        assert np.max(np.abs(inv @ A - np.eye(n))) < 1e-10 * np.linalg.cond(A) / max(n, 10) + 1e-11, n
        assert _rel(inv, np.linalg.inv(A)) < 1e-9
    Sd = to_device(np.ones((6, 6), dtype=complex))
    assert _hx.lib.hx_zdense_inverse(6, _hx.ptr(Sd), _hx.stream()) == _hx.ERR_BREAKDOWN
    # options registry: an override is visible at once and removable
    _hx.set_option("cocg_lag", 0)
    assert _hx.get_option("cocg_lag") == "0"
    _hx.set_option("cocg_lag", None)
    assert _hx.get_option("cocg_lag", "unset") in ("unset", __import__("os").environ.get("HX_COCG_LAG", "unset"))


def test_pool_orders_the_reuse_of_cached_blocks_across_streams():
    """Device-buffer pool: a block released while a kernel of a NON-DEFAULT (non-blocking) torch stream still writes
    it is handed to the next request of the same size only after that kernel has finished -- the release records events on the
    caller's streams, the reuse waits for them.  Checked with a long-running fill on a side stream followed at once by a release
    and a re-allocation through the library (SELL-64 handles are pooled blocks)."""
    import ctypes as C
    import torch
    from helmholtz_x import _hx
    ev0, w0 = C.c_int64(0), C.c_int64(0)
    _hx.check(_hx.lib.hx_pool_ordering_stats(C.byref(ev0), C.byref(w0), None))
    import scipy.sparse as sp
    from helmholtz_x.petsc import CSR
    n = 400_000
    A = sp.diags([np.full(n, 2.0 + 1j), np.full(n - 1, -1.0 + 0j), np.full(n - 1, -1.0 + 0j)], [0, 1, -1]).tocsr()
    K = CSR.from_scipy(A)
    side = torch.cuda.Stream()
    x = torch.ones(n, dtype=torch.complex128, device="cuda")
    y = torch.empty_like(x)
    one, zero = _hx.z2(1.0), _hx.z2(0.0)
    results = []
    for rep in range(3):
        h = C.c_void_p()
        with torch.cuda.stream(side):
            _hx.check(_hx.lib.hx_sell_create(C.byref(h), n, _hx.ptr(K.rowptr), _hx.ptr(K.col), _hx.ptr(K.val), 1, 0, _hx.stream()))
            for _ in range(20):
                _hx.check(_hx.lib.hx_sell_spmv_z(h, _hx.ptr(x), _hx.ptr(y), one, zero, _hx.stream()))
            results.append(y.clone())
            _hx.lib.hx_sell_destroy(h)         # its blocks go to the cache while the side stream may still be running
    torch.cuda.synchronize()
    ref = A @ np.ones(n)
    for r in results:
        assert _rel(r.cpu().numpy(), ref) < 1e-14
    ev1, w1, cap = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    _hx.check(_hx.lib.hx_pool_ordering_stats(C.byref(ev1), C.byref(w1), C.byref(cap)))
    assert ev1.value > ev0.value and w1.value > w0.value       # releases recorded events, reuses waited for them
    free_b, total_b = torch.cuda.mem_get_info()
    assert 0 < cap.value <= total_b // 4


@pytest.mark.parametrize("fmt", ["stencil", "i16", "i32"])
def test_block_products_and_block_cocg_match_the_single_vector_path(fmt):
    """Blocks of right-hand sides (csrc/block.hip, cocg_block.hip): the SELL-64 block product on the three index formats equals
    four single-vector products; hx_cocg_block (four deflated COCG recurrences side by side, block V-cycle) returns the
    solutions of four hx_cocg solves to the solver's tolerance in about the same number of iterations; a zero column and a
    partial block (3 of 4 columns) are handled; AMGSolver.solve_block falls back cleanly when the batched route is off."""
    import torch
    import bench
    from helmholtz_x import _hx
    from helmholtz_x.acoustic_matrices import AcousticMatrices
    from helmholtz_x.amg import AMGSolver
    from helmholtz_x.petsc import combine_csr, to_device
    if fmt == "i32":
        _hx.set_option("sell_index", "i32")
    if fmt != "stencil":
        _hx.set_option("sell_stencil", 0)
    try:
        case = bench.build_case((10, 128, 12), "native" if fmt == "stencil" else "rcm", 0.0 if fmt == "stencil" else 0.2)
        P = case["P"]
        M = AcousticMatrices(case["mesh"], case["facet_tags"], {11: {'Robin': P.R_outlet}}, case["c"])
        K = combine_csr(M.A.shape, M.A.csr_terms + [(P.target * c, m) for c, m in M.B.csr_terms]
                        + [(P.target ** 2 * c, m) for c, m in M.C.csr_terms])
        n = K.shape[0]
        solver = AMGSolver(K, cocg_min_rows=0, recycle_min_rows=0, sell_min_rows=0)
        st = solver.stats()
        assert st["fine_format"].startswith("sell64")
        assert (st["stencil_slices"] > 0) == (fmt == "stencil")
        assert st["fine_format"].endswith("i32" if fmt == "i32" else "i16")
        rng = np.random.default_rng(7)
        w = _hx.lib.hx_block_width()
        assert w == 4
        # block product against single-vector products
        Xp = to_device(_rand_c(rng, w, n))
        blk, yblk = torch.empty(n * w, dtype=torch.complex128, device="cuda"), torch.empty(n * w, dtype=torch.complex128, device="cuda")
        _hx.check(_hx.lib.hx_zblock_pack(n, w, _hx.ptr(Xp), n, _hx.ptr(blk), _hx.stream()))
        assert torch.equal(blk.reshape(n, w).t().contiguous(), Xp)
        al, be = _hx.z2(0.7 - 0.3j), _hx.z2(0.0)
        _hx.check(_hx.lib.hx_sell_spmm_z(solver.sell, _hx.ptr(blk), _hx.ptr(yblk), al, be, _hx.stream()))
        Yp = torch.empty_like(Xp)
        _hx.check(_hx.lib.hx_zblock_unpack(n, w, _hx.ptr(yblk), _hx.ptr(Yp), n, _hx.stream()))
        for c in range(w):
            y1 = torch.empty(n, dtype=torch.complex128, device="cuda")
            _hx.check(_hx.lib.hx_sell_spmv_z(solver.sell, _hx.ptr(Xp[c]), _hx.ptr(y1), al, be, _hx.stream()))
            assert float(torch.linalg.norm(Yp[c] - y1) / torch.linalg.norm(y1)) < 1e-14
        # solves: the first one harvests (GMRES), the second prepares the deflation pairs (hx_cocg)
        B = to_device(_rand_c(rng, 7, n))
        B[5].zero_()                                        # a zero right-hand side inside a block
        Xs = torch.zeros_like(B)
        assert not solver.block_ready()
        for c in range(2):
            solver.solve(B[c], Xs[c])
        assert solver.block_ready()
        its_single = []
        for c in range(7):
            solver.solve(B[c], Xs[c])
            its_single.append(solver.last[0])
        Xb = torch.full_like(B, 7.0)
        solver.solve_block(B, Xb)                           # 4 + 3 columns
        stb = solver.stats()
        assert stb["block_solves"] == 2 and stb["block_columns"] == 7
        for c in range(7):
            if c == 5:
                assert float(torch.abs(Xb[c]).max()) == 0.0
                continue
            r = torch.empty(n, dtype=torch.complex128, device="cuda")
            K.spmv(Xb[c], r)
            assert float(torch.linalg.norm(B[c] - r) / torch.linalg.norm(B[c])) < 1e-10
            assert float(torch.linalg.norm(Xb[c] - Xs[c]) / torch.linalg.norm(Xs[c])) < 1e-8
        nz = [c for c in range(7) if c != 5]
        assert stb["block_iters"] <= max(its_single[c] for c in nz[:4]) + max(its_single[c] for c in nz[4:]) + 4
        # batched route switched off: the same call goes vector by vector
        import os
        os.environ["HX_BLOCK_SOLVES"] = "0"
        try:
            Xc = torch.zeros_like(B)
            solver.solve_block(B[:2], Xc[:2])
            assert solver.stats()["block_solves"] == 2
            assert float(torch.linalg.norm(Xc[0] - Xs[0]) / torch.linalg.norm(Xs[0])) < 1e-8
        finally:
            del os.environ["HX_BLOCK_SOLVES"]
    finally:
        _hx.set_option("sell_index", None)
        _hx.set_option("sell_stencil", None)


def test_krylov_basis_rotation_in_place_and_destroy_releases_memory():
    """Round 4: DeviceBasis.rotate runs IN PLACE through a slab scratch (the second panel is gone) -- one slab and several slabs
    against NumPy, thick-restart shape (k kept columns + the residual column) and final shape (k == active columns); and
    E.destroy() of the solver a fixed-point call returns gives the device memory back without a garbage collection."""
    import gc
    import torch
    from helmholtz_x import slepc
    rng = np.random.default_rng(11)
    for n in (5000, 300_000):                      # (scratch slab: max(n / 8, 65536) positions -> 1 and 5 slabs)
        m = 12
        basis = slepc.DeviceBasis(n, m + 1, apply_op=None)
        V0 = _rand_c(rng, (m + 1) * n).reshape(m + 1, n)
        basis.V.copy_(_dev(V0))
        assert basis._scratch.shape[1] == min(n, max(-(-n // slepc.ROTATE_SLABS), 65536))
        for mact, k in ((m, 5), (7, 7)):           # thick restart: 12 active -> 5 kept; final Ritz vectors: 7 of 7 active
            Vin = basis.V.cpu().numpy().copy()
            Q = _rand_c(rng, mact * k).reshape(mact, k)
            basis.rotate(Q, k)
            Vout = basis.V.cpu().numpy()
            ref = Q.T @ Vin[:mact]
            assert np.max(np.abs(Vout[:k] - ref)) <= 1e-13 * np.max(np.abs(ref))
            assert np.array_equal(Vout[k], Vin[mact])                       # residual column moved behind the kept ones
        basis.close()
        assert basis.V is None and basis._panel is None
    # ---- destroy() releases: a small fixed-point call, then memory with the collector switched off
    import bench
    case = bench.build_case(bench.WORKLOADS["annulus-16k"])
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        E, matrices = bench.hot_path(case)          # warm-up: pools, pinned buffers, the case's device arrays
        E.destroy()
        del E, matrices
        gc.collect()
        torch.cuda.synchronize()
        base = torch.cuda.memory_allocated()
        gc.disable()
        try:
            E, matrices = bench.hot_path(case)
            held = torch.cuda.memory_allocated() - base
            E.destroy()
            del E, matrices
            left = torch.cuda.memory_allocated() - base
        finally:
            gc.enable()
    assert held > 20e6                              # the call did hold device memory (basis, hierarchy, operators) ...
    assert left <= 0.05 * held, (held, left)        # ... and destroy() + dropping the references gave it back, no collection needed
    from helmholtz_x import petsc
    assert all(torch.is_tensor(v) for v in petsc._SLOT_CACHE.values())     # the slot cache holds maps, never operators


def test_pool_notes_every_caller_stream_where_it_is_used():
    """a block whose last kernels ran on caller stream A must get an event on A even if the caller made a libhx launch
    on stream B before the handle was destroyed -- hx_s() notes a stream when it is USED (not only the last one at release).
    Two non-blocking caller streams: long product chain on A, one short launch on B, release, re-allocate, overwrite."""
    import ctypes as C
    import torch
    from helmholtz_x import _hx
    from helmholtz_x.petsc import CSR
    n = 400_000
    A = sp.diags([np.full(n, 2.0 + 1j), np.full(n - 1, -1.0 + 0j), np.full(n - 1, -1.0 + 0j)], [0, 1, -1]).tocsr()
    K = CSR.from_scipy(A)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    x = torch.ones(n, dtype=torch.complex128, device="cuda")
    ya, yb = torch.empty_like(x), torch.empty_like(x)
    one, zero = _hx.z2(1.0), _hx.z2(0.0)
    torch.cuda.synchronize()
    _hx.check(_hx.lib.hx_pool_trim())                      # forget earlier streams: the set starts empty
    ev0 = C.c_int64(0)
    _hx.check(_hx.lib.hx_pool_ordering_stats(C.byref(ev0), None, None))
    h = C.c_void_p()
    with torch.cuda.stream(sa):
        _hx.check(_hx.lib.hx_sell_create(C.byref(h), n, _hx.ptr(K.rowptr), _hx.ptr(K.col), _hx.ptr(K.val), 1, 0, _hx.stream()))
        for _ in range(40):
            _hx.check(_hx.lib.hx_sell_spmv_z(h, _hx.ptr(x), _hx.ptr(ya), one, zero, _hx.stream()))
    with torch.cuda.stream(sb):                            # the LAST stream the library sees before the release is B
        K.spmv(x, yb)
    _hx.check(_hx.lib.hx_pool_ordering_stats(C.byref(ev0), None, None))     # (the set-up's own scratch releases are behind us)
    _hx.lib.hx_sell_destroy(h)                             # released while stream A may still be reading the block
    ev1 = C.c_int64(0)
    _hx.check(_hx.lib.hx_pool_ordering_stats(C.byref(ev1), None, None))
    # one event per noted caller stream (A and B) and pooled block: with A forgotten there would be half as many
    h2 = C.c_void_p()
    val2 = (2.0 * K.val).contiguous()
    _hx.check(_hx.lib.hx_sell_create(C.byref(h2), n, _hx.ptr(K.rowptr), _hx.ptr(K.col), _hx.ptr(val2), 1, 0, _hx.stream()))
    torch.cuda.synchronize()
    assert _rel(ya.cpu().numpy(), A @ np.ones(n)) < 1e-14          # A's products never saw the re-used block's new values
    y2 = torch.empty_like(x)
    _hx.check(_hx.lib.hx_sell_spmv_z(h2, _hx.ptr(x), _hx.ptr(y2), one, zero, _hx.stream()))
    assert _rel(y2.cpu().numpy(), 2.0 * (A @ np.ones(n))) < 1e-14
    _hx.lib.hx_sell_destroy(h2)
    n_blocks_events = ev1.value - ev0.value
    assert n_blocks_events >= 2 and n_blocks_events % 2 == 0, n_blocks_events     # events on BOTH streams for every released block
    torch.cuda.synchronize()
    _hx.check(_hx.lib.hx_pool_trim())


def test_factored_flame_term_scatter_with_device_coefficients(hx):
    """hx_zsparse_axpys: y += coef * sum_f t[f] u_f over packed left factors, t on the device -- disjoint supports in one launch,
    overlapping supports one launch per factor (fixed order); and Mat.mult with a low-rank term against the explicit matrix."""
    import torch
    from helmholtz_x.petsc import CSR, LowRank, Mat, Vec
    rng = np.random.default_rng(5)
    n, r = 5000, 7
    for overlapping in (False, True):
        pool = rng.permutation(n)
        sup = [np.sort(pool[300 * f:300 * f + 200 + 13 * f]) for f in range(r)]
        if overlapping:
            sup[3] = np.sort(np.unique(np.concatenate([sup[3][:50], sup[2][:80], sup[5][10:40]])))
        vals = [_rand_c(rng, len(s)) for s in sup]
        ptr = np.zeros(r + 1, np.int32)
        ptr[1:] = np.cumsum([len(s) for s in sup])
        idx = np.concatenate(sup).astype(np.int32)
        t = _rand_c(rng, r)
        y0 = _rand_c(rng, n)
        coef = 0.3 - 1.7j
        ref = y0.copy()
        for f in range(r):
            ref[sup[f]] += coef * t[f] * vals[f]
        y = _dev(y0)
        # (device operands are bound to names: a temporary would be released -- and its block handed to the next allocation --
        # before the asynchronous launch has read it)
        d_ptr, d_idx, d_val, d_t = _dev(ptr), _dev(idx), _dev(np.concatenate(vals)), _dev(t)
        hx.check(hx.lib.hx_zsparse_axpys(r, hx.ptr(d_ptr), hx.ptr(d_idx), hx.ptr(d_val), hx.ptr(d_t),
                                         hx.z2(coef), hx.ptr(y), int(not overlapping), hx.stream()))
        torch.cuda.synchronize()
        assert _rel(y.cpu().numpy(), ref) < 1e-14
        del d_ptr, d_idx, d_val, d_t
        # the petsc4py-shaped product: (S + coef * sum_f u_f v_f^T) x
        S = sp.random(n, n, density=0.001, random_state=rng, format="csr") + sp.eye(n)
        S = sp.csr_matrix(S.astype(np.complex128))
        rsup = [np.sort(rng.choice(n, 4, replace=False)).astype(np.int32) for _ in range(r)]
        rval = [_rand_c(rng, 4) for _ in range(r)]
        lr = LowRank(n, [((sup[f].astype(np.int32), vals[f]), (rsup[f], rval[f])) for f in range(r)])
        assert lr.left_packed_all()[3] == int(not overlapping)
        M = Mat((n, n), [(1.0, CSR.from_scipy(S))], [(coef, lr)])
        xv = _rand_c(rng, n)
        out = Vec.zeros(n)
        M.mult(Vec(xv), out)
        expl = S @ xv
        for f in range(r):
            expl[sup[f]] += coef * vals[f] * (rval[f] @ xv[rsup[f]])
        assert _rel(out.array, expl) < 1e-13


def test_debug_bounds_turns_a_wild_index_into_an_error_code(hx):
    """HX_DEBUG_BOUNDS (on for this suite, conftest.py): an index array pointing outside the allocation of the vector it indexes
    -- what a stale or recycled temporary looks like to a kernel -- is refused with HX_ERR_ARG naming the position, for every entry
    point that follows caller-supplied indices; nothing is launched, the device stays healthy (the last call of the test runs)."""
    import torch
    assert hx.get_option("debug_bounds") == "1"
    n = 1000
    rng = np.random.default_rng(0)
    x = _dev(_rand_c(rng, n))
    y = torch.zeros_like(x)
    good = _dev(rng.permutation(n).astype(np.int32))
    wild = good.clone()
    wild[137] = 2 ** 30                              # ~16 GiB past the start of a 16 kB vector
    hx.check(hx.lib.hx_zgather(n, hx.ptr(x), hx.ptr(good), hx.ptr(y), hx.stream()))
    for call in (lambda: hx.lib.hx_zgather(n, hx.ptr(x), hx.ptr(wild), hx.ptr(y), hx.stream()),
                 lambda: hx.lib.hx_zscatter_axpy(n, hx.z2(1.0), hx.ptr(x), hx.ptr(wild), hx.ptr(y), hx.stream())):
        rc = call()
        assert rc == -1, rc                           # HX_ERR_ARG
        msg = hx.lib.hx_last_error().decode()
        assert "debug_bounds" in msg and "entry 137" in msg, msg
    ptr = _dev(np.array([0, 400, n], np.int32))
    vals, t, out = _dev(_rand_c(rng, n)), _dev(_rand_c(rng, 2)), _dev(np.zeros(2, np.complex128))
    for call in (lambda idx: hx.lib.hx_zsparse_dots(2, hx.ptr(ptr), hx.ptr(idx), hx.ptr(vals), hx.ptr(x), hx.ptr(out), hx.stream()),
                 lambda idx: hx.lib.hx_zsparse_axpys(2, hx.ptr(ptr), hx.ptr(idx), hx.ptr(vals), hx.ptr(t), hx.z2(0.5), hx.ptr(y), 1, hx.stream())):
        assert call(good) == 0
        assert call(wild) == -1 and "entry 137" in hx.lib.hx_last_error().decode()
    hx.set_option("debug_bounds", "0")               # off: the same call is not inspected (not made here: it would fault)
    try:
        hx.check(hx.lib.hx_zgather(n, hx.ptr(x), hx.ptr(good), hx.ptr(y), hx.stream()))
    finally:
        hx.set_option("debug_bounds", None)
    torch.cuda.synchronize()
    assert np.array_equal(y.cpu().numpy(), x.cpu().numpy()[good.cpu().numpy()])
