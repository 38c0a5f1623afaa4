"""Oracle: P1 finite-element assembly on simplices (intervals / triangles / tetrahedra).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows, form by form,
helmholtz_x/acoustic_matrices.py:46-125 and helmholtz_x/flame_matrices.py:61-73,139-160,199-216
of the reference; the element integrals DOLFINx/FFCx would generate for those UFL forms are
restated in closed form (exact simplex monomial integrals), except for the one non-polynomial
form (division by a P1 density), which uses the degree-2 simplex rule FFCx selects for it.

All functions take plain arrays:
  x      (n, 3)  float64 vertex coordinates (DOLFINx pads to 3 columns)
  cells  (nc, d+1) int32 vertex ids (P1 dofmap == vertex ids)
"""
from math import factorial

import numpy as np
import scipy.sparse as sp


# ----------------------------------------------------------------------------- geometry
def cell_geometry(x, cells):
    """Return (G, vol): G[c, i, :] = grad(phi_i) on cell c (nc, d+1, d), vol[c] = |K_c|."""
    cells = np.asarray(cells)
    d = cells.shape[1] - 1
    X = x[cells][:, :, :d]                       # (nc, d+1, d)
    J = np.transpose(X[:, 1:, :] - X[:, :1, :], (0, 2, 1))   # J[:, :, i] = v_i - v_0
    detJ = np.linalg.det(J)
    Jinv = np.linalg.inv(J)                      # rows of Jinv = physical gradients of phi_1..d
    G = np.empty((cells.shape[0], d + 1, d))
    G[:, 1:, :] = Jinv
    G[:, 0, :] = -Jinv.sum(axis=1)
    vol = np.abs(detJ) / factorial(d)
    return G, vol


def facet_measure(x, facets):
    """Measure of boundary facets (nf, d) -> (nf,).  Points have measure 1."""
    facets = np.asarray(facets)
    d = facets.shape[1]
    if d == 1:
        return np.ones(facets.shape[0])
    X = x[facets]
    if d == 2:
        return np.linalg.norm(X[:, 1] - X[:, 0], axis=1)
    return 0.5 * np.linalg.norm(np.cross(X[:, 1] - X[:, 0], X[:, 2] - X[:, 0]), axis=1)


def _scatter(n, rows, cols, vals):
    M = sp.coo_matrix((vals.ravel(), (rows.ravel(), cols.ravel())), shape=(n, n)).tocsr()
    M.sum_duplicates()
    M.sort_indices()
    return M


def _cell_scatter(n, cells, Ae):
    k = cells.shape[1]
    rows = np.repeat(cells[:, :, None], k, axis=2)
    cols = np.repeat(cells[:, None, :], k, axis=1)
    return _scatter(n, rows, cols, Ae)


def mass_template(d):
    """(1+delta_ij)/((d+1)(d+2)): exact P1 mass matrix on a unit-measure d-simplex."""
    return (np.ones((d + 1, d + 1)) + np.eye(d + 1)) / ((d + 1) * (d + 2))


def triple_template(d):
    """T[i,j,k] = int phi_i phi_j phi_k on a unit-measure d-simplex = d! a!b!c!/(d+3)!."""
    k = d + 1
    T = np.empty((k, k, k))
    for i in range(k):
        for j in range(k):
            for l in range(k):
                cnt = np.bincount([i, j, l], minlength=k)
                T[i, j, l] = factorial(d) * np.prod([factorial(c) for c in cnt]) / factorial(d + 3)
    return T


# ----------------------------------------------------------------------------- coefficients
def gamma_of_T(T):
    """parameters_utils.py:62-78 (r_gas = 287.1 as written there)."""
    cp = 973.60091 + 0.1333 * T
    return cp / (cp - 287.1)


def sound_speed_variable_gamma(T):
    """parameters_utils.py:80-93."""
    return np.sqrt(gamma_of_T(T) * 287.1 * T)


# ----------------------------------------------------------------------------- operators
def apply_dirichlet(M, dofs):
    """DOLFINx assemble_matrix(bcs=...) semantics: zero rows+cols, diagonal = 1."""
    if len(dofs) == 0:
        return M
    n = M.shape[0]
    keep = np.ones(n)
    keep[dofs] = 0.0
    Dk = sp.diags(keep)
    ind = sp.diags(1.0 - keep)
    # keep the full sparsity pattern (explicit zeros) as DOLFINx does
    M = (Dk @ M @ Dk + ind).tocsr()
    M.sort_indices()
    return M


def assemble_A(x, cells, c, dirichlet_dofs=(), c_is_cellwise=None):
    """A_jk = - int c^2 grad(phi_k).grad(phi_j) dx   (acoustic_matrices.py:101-105)."""
    n = x.shape[0]
    G, vol = cell_geometry(x, cells)
    c = np.asarray(c, dtype=complex)
    if c_is_cellwise is None:
        c_is_cellwise = (c.shape[0] == cells.shape[0] and c.shape[0] != n)
    if c_is_cellwise:
        w = c ** 2 * vol
    else:
        d = cells.shape[1] - 1
        cc = c[cells]
        w = vol / ((d + 1) * (d + 2)) * (cc.sum(axis=1) ** 2 + (cc ** 2).sum(axis=1))
    GG = np.einsum('cid,cjd->cij', G, G)
    Ae = -w[:, None, None] * GG
    A = _cell_scatter(n, cells, Ae.astype(complex))
    return apply_dirichlet(A, dirichlet_dofs)


def assemble_C(x, cells, dirichlet_dofs=()):
    """C_jk = int phi_k phi_j dx   (acoustic_matrices.py:121-125)."""
    n = x.shape[0]
    _, vol = cell_geometry(x, cells)
    d = cells.shape[1] - 1
    Ce = vol[:, None, None] * mass_template(d)[None]
    C = _cell_scatter(n, cells, Ce.astype(complex))
    return apply_dirichlet(C, dirichlet_dofs)


def robin_Z(R):
    """acoustic_matrices.py:70."""
    return (1 + R) / (1 - R)


def choked_inlet_R(gamma_inlet, Mach):
    """acoustic_matrices.py:81."""
    g = gamma_inlet
    t = g * Mach / (1 + (g - 1) * Mach ** 2)
    return (1 - t) / (1 + t)


def choked_outlet_R(gamma_outlet, Mach):
    """acoustic_matrices.py:93."""
    g = gamma_outlet
    return (1 - 0.5 * (g - 1) * Mach) / (1 + 0.5 * (g - 1) * Mach)


def facet_average(x, facets, f, facet_cells=None):
    """(1/|Gamma|) int_Gamma f ds for P1 (nodal) or DG0 (cell) f  (acoustic_matrices.py:76-78)."""
    meas = facet_measure(x, facets)
    f = np.asarray(f)
    if facet_cells is not None and f.shape[0] != x.shape[0]:
        vals = f[facet_cells]
    else:
        vals = f[facets].mean(axis=1)
    return (vals * meas).sum() / meas.sum()


def assemble_B(x, n, terms):
    """B_jk = sum_tags (i/Z) int_Gamma c phi_k phi_j ds   (acoustic_matrices.py:71,83,95,108-110).

    terms: list of (facets (nf, d), Z, c, facet_cells or None).  c nodal (P1) or cellwise (DG0,
    needs facet_cells).  Assembled WITHOUT Dirichlet conditions, like the reference."""
    rows, cols, vals = [], [], []
    for facets, Z, c, facet_cells in terms:
        facets = np.asarray(facets)
        nf, k = facets.shape
        meas = facet_measure(x, facets)
        c = np.asarray(c, dtype=complex)
        if facet_cells is not None and c.shape[0] != x.shape[0]:
            Be = (c[facet_cells] * meas)[:, None, None] * mass_template(k - 1)[None]
        else:
            T = triple_template(k - 1)
            Be = meas[:, None, None] * np.einsum('fm,mjk->fjk', c[facets], T)
        Be = (1j / Z) * Be
        rows.append(np.repeat(facets[:, :, None], k, axis=2).ravel())
        cols.append(np.repeat(facets[:, None, :], k, axis=1).ravel())
        vals.append(Be.ravel())
    if not rows:
        return None
    return _scatter(n, np.concatenate(rows), np.concatenate(cols), np.concatenate(vals))


# ----------------------------------------------------------------------------- fields
def gaussian(x, x_ref, sigma, n):
    """parameters_utils.py:8-33 (x: (npts,3); x_ref: length-3)."""
    x_ref = np.asarray(x_ref, dtype=float).reshape(-1)
    spatial = sum((x[:, i] - x_ref[i]) ** 2 for i in range(n))
    amplitude = 1 / (sigma ** n * (2 * np.pi) ** (n / 2))
    return amplitude * np.exp(-spatial / (2 * sigma ** 2))


def integrate_p1(x, cells, f):
    """int f dx for nodal P1 f (dolfinx_utils.py:42-43: quadrature degree 1 is exact)."""
    _, vol = cell_geometry(x, cells)
    return (vol * f[cells].mean(axis=1)).sum()


def gaussian_function(x, cells, x_r, a_r):
    """parameters_utils.py:35-43: nodal interpolation, then scale so that int w dx = 1."""
    d = cells.shape[1] - 1
    w = gaussian(x, np.asarray(x_r).reshape(-1, 3)[0], a_r, d).astype(complex)
    return w / integrate_p1(x, cells, w)


def density_step(xa, x_f, sigma, rho_d, rho_u):
    """parameters_utils.py:106-107."""
    return rho_u + (rho_d - rho_u) / 2 * (1 + np.tanh((xa - x_f) / sigma))


def step_field(x, d, x_f, up, down):
    """c_step / temperature_step (parameters_utils.py:129-153,185-208): axis 0 (1D/2D) or 2 (3D)."""
    axis = 2 if d == 3 else 0
    xf = np.asarray(x_f).reshape(-1, 3)[0][axis]
    return np.where(x[:, axis] < xf, up, down).astype(complex)


def q_multiple(x, cells, cell_tags, n_sector):
    """parameters_utils.py:228-247: DG0 field = 1/V_f on cells tagged f."""
    _, vol = cell_geometry(x, cells)
    q = np.zeros(cells.shape[0], dtype=complex)
    for f in range(n_sector):
        m = cell_tags == f
        q[m] = 1.0 / vol[m].sum()
    return q


# ----------------------------------------------------------------------------- load vectors
def threshold_pack(v, tol):
    """flame_matrices.py:66-72: zero |Re|<tol and |Im|<tol separately; (int32 idx, c128 val)."""
    v = np.array(v, dtype=complex)
    v.real[np.abs(v.real) < tol] = 0.0
    v.imag[np.abs(v.imag) < tol] = 0.0
    idx = np.flatnonzero(v).astype(np.int32)
    return idx, v[idx]


def distributed_left(x, cells, gamma, h, q_0, u_b):
    """a_j = int (gamma-1) q0/u_b phi_j h dx  (flame_matrices.py:199); gamma nodal array or scalar."""
    n = x.shape[0]
    d = cells.shape[1] - 1
    _, vol = cell_geometry(x, cells)
    hc = np.asarray(h, dtype=complex)[cells]
    if np.isscalar(gamma):
        loc = (gamma - 1) * vol[:, None] * np.einsum('jl,cl->cj', mass_template(d), hc)
    else:
        gm1 = (np.asarray(gamma, dtype=complex) - 1)[cells]
        loc = vol[:, None] * np.einsum('jml,cm,cl->cj', triple_template(d), gm1, hc)
    a = np.zeros(n, dtype=complex)
    np.add.at(a, cells.ravel(), (q_0 / u_b * loc).ravel())
    return a


_QUAD2 = {
    1: (np.array([[0.5 + 0.5 / np.sqrt(3), 0.5 - 0.5 / np.sqrt(3)],
                  [0.5 - 0.5 / np.sqrt(3), 0.5 + 0.5 / np.sqrt(3)]]), np.array([0.5, 0.5])),
    2: (np.array([[2 / 3, 1 / 6, 1 / 6], [1 / 6, 2 / 3, 1 / 6], [1 / 6, 1 / 6, 2 / 3]]),
        np.array([1 / 3, 1 / 3, 1 / 3])),
    3: (np.array([[0.5854101966249685, 0.1381966011250105, 0.1381966011250105, 0.1381966011250105],
                  [0.1381966011250105, 0.5854101966249685, 0.1381966011250105, 0.1381966011250105],
                  [0.1381966011250105, 0.1381966011250105, 0.5854101966249685, 0.1381966011250105],
                  [0.1381966011250105, 0.1381966011250105, 0.1381966011250105, 0.5854101966249685]]),
        np.array([0.25, 0.25, 0.25, 0.25])),
}


def reference_direction(d):
    """flame_matrices.py:35-40: e_x in 1D/2D, e_z in 3D."""
    n_r = np.zeros(d)
    n_r[2 if d == 3 else 0] = 1.0
    return n_r


def distributed_right(x, cells, rho, w):
    """b_k = int (n_r.grad phi_k)/rho * w dx  (flame_matrices.py:200).

    UFL estimates degree 0+1+1 = 2 for this (non-polynomial) integrand, so FFCx integrates it with
    the degree-2 simplex rule (2-pt Gauss / 3-pt / 4-pt), restated here in barycentric form."""
    n = x.shape[0]
    d = cells.shape[1] - 1
    G, vol = cell_geometry(x, cells)
    pts, wts = _QUAD2[d]
    rq = np.asarray(rho, dtype=complex)[cells] @ pts.T      # (nc, nq)
    wq = np.asarray(w, dtype=complex)[cells] @ pts.T
    s = ((wq / rq) * wts[None, :]).sum(axis=1) * vol
    gn = G @ reference_direction(d)                          # (nc, d+1)
    b = np.zeros(n, dtype=complex)
    np.add.at(b, cells.ravel(), (gn * s[:, None]).ravel())
    return b


def pointwise_left(x, cells, cell_tags, flame, h_cell, gamma, q_0, u_b):
    """a_j = (gamma-1) q0/u_b int_{Omega_f} h phi_j dx, h DG0  (flame_matrices.py:141)."""
    n = x.shape[0]
    d = cells.shape[1] - 1
    _, vol = cell_geometry(x, cells)
    m = np.flatnonzero(cell_tags == flame)
    loc = (gamma - 1) * q_0 / u_b * np.asarray(h_cell, dtype=complex)[m] * vol[m] / (d + 1)
    a = np.zeros(n, dtype=complex)
    np.add.at(a, cells[m].ravel(), np.repeat(loc, d + 1))
    return a


def locate_cell(x, cells, point, tol=1e-10):
    """Index of a cell containing ``point`` (flame_matrices.py:144-145), lowest index on ties."""
    d = cells.shape[1] - 1
    p = np.asarray(point, dtype=float).reshape(-1)[:d]
    X = x[cells][:, :, :d]
    J = np.transpose(X[:, 1:, :] - X[:, :1, :], (0, 2, 1))
    lam = np.linalg.solve(J, (p[None, :] - X[:, 0, :])[:, :, None])[:, :, 0]
    bary = np.concatenate([1 - lam.sum(axis=1, keepdims=True), lam], axis=1)
    hit = np.flatnonzero((bary >= -tol).all(axis=1))
    if hit.size == 0:
        raise ValueError("point not inside the mesh")
    return int(hit[0])


def pointwise_right(x, cells, point, rho_u):
    """b_k = (grad phi_k . n_r)(x_r)/rho_u on the d+1 dofs of the cell holding x_r
    (flame_matrices.py:144-156)."""
    d = cells.shape[1] - 1
    c = locate_cell(x, cells, point)
    G, _ = cell_geometry(x, cells[c:c + 1])
    vals = (G[0] @ reference_direction(d)) / rho_u
    return cells[c].astype(np.int32), vals.astype(complex)


def flame_outer(n, left, right, problem_type='direct'):
    """Explicit D_ij = a (x) b as the reference materialises it (flame_matrices.py:75-94,
    162-189, 218-244); adjoint = plain transpose (no conjugate)."""
    (li, lv), (ri, rv) = left, right
    if problem_type == 'adjoint':
        (li, lv), (ri, rv) = (ri, rv), (li, lv)
    rows = np.repeat(li, len(ri))
    cols = np.tile(ri, len(li))
    vals = np.outer(lv, rv).ravel()
    M = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()
    M.sum_duplicates()
    return M
