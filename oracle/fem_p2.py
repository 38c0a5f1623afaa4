"""Oracle: P2 (quadratic Lagrange) assembly on tetrahedra, with P1 coefficient fields.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates what DOLFINx/FFCx generate for the reference's forms when
``degree = 2`` (ShapeSensitivities/RijkeFFD/main_shape.py:68-83): the trial/test space is P2, every coefficient
(temperature -> sound speed, density, w, h) stays P1:

  A_jk = - int c^2 grad(phi_k).grad(phi_j) dx          (acoustic_matrices.py:101; degree 2 + 1 + 1 = 4)
  C_jk =   int phi_k phi_j dx                           (acoustic_matrices.py:121; degree 4)
  B_jk = sum_tags (i/Z) int_Gamma c phi_k phi_j ds      (acoustic_matrices.py:70,108; degree 5 on triangles)
  a_j  =   int (gamma-1) q0/u_b phi_j h dx              (flame_matrices.py:199; degree 3)
  b_k  =   int (n_r.grad phi_k) w / rho dx              (flame_matrices.py:200; NOT polynomial)

All polynomial integrands are integrated exactly by a collapsed Gauss-Legendre rule (4 points per direction: exact to
degree 5 on the simplex); the non-polynomial one with the same rule.  FFCx would use basix's degree-3 Xiao-Gimbutas
rule there; the rule matters at the 1e-10 level only.

PINNED to the reference (tests/test_oracle_goldens.py::test_rijkeffd_degree2_eigenpair_and_36_shape_derivatives_reference_goldens):
on the mesh of the reference's own degree-2 run (recovered from Results/ShapeDerivatives/p_dir.h5) this module reproduces
the stored 16-digit eigenvalue 1156.7431442242196+0.8682828502864666j to 1.6e-10, the logged trajectory and the vertex samples
of the stored P2 modes to 1e-8.

DOF numbering: the mesh vertices first (P1 numbering), then one DOF per edge, edges sorted lexicographically by their
(low, high) vertex pair.  DOLFINx numbers differently; eigenvalues do not depend on it.
"""
import numpy as np
import scipy.sparse as sp

from . import fem

TET_EDGES = np.array([(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)])
TRI_EDGES = np.array([(0, 1), (0, 2), (1, 2)])


# ----------------------------------------------------------------------------- dof map
def p2_dofmap(n_vertices, cells):
    """(cell_dofs (nc, 10), edges (ne, 2)): local order = 4 vertices, then the 6 edges in TET_EDGES order."""
    cells = np.asarray(cells, dtype=np.int64)
    pairs = np.sort(cells[:, TET_EDGES], axis=2).reshape(-1, 2)              # (nc*6, 2)
    key = pairs[:, 0] * n_vertices + pairs[:, 1]
    uniq, inv = np.unique(key, return_inverse=True)
    edges = np.stack([uniq // n_vertices, uniq % n_vertices], axis=1)
    cell_dofs = np.concatenate([cells, n_vertices + inv.reshape(-1, 6)], axis=1)
    return cell_dofs.astype(np.int32), edges.astype(np.int32)


def facet_dofs(n_vertices, edges, facets):
    """(nf, 6): 3 vertices + the 3 edges in TRI_EDGES order, as global P2 dofs."""
    facets = np.asarray(facets, dtype=np.int64)
    ekey = edges[:, 0].astype(np.int64) * n_vertices + edges[:, 1]
    pairs = np.sort(facets[:, TRI_EDGES], axis=2)
    key = pairs[..., 0] * n_vertices + pairs[..., 1]
    pos = np.searchsorted(ekey, key)
    assert np.array_equal(ekey[pos], key), "boundary facet edge is not a mesh edge"
    return np.concatenate([facets, n_vertices + pos], axis=1).astype(np.int32)


def dof_coordinates(x, edges):
    return np.concatenate([x, 0.5 * (x[edges[:, 0]] + x[edges[:, 1]])], axis=0)


# ----------------------------------------------------------------------------- quadrature + basis
def collapsed_gauss(dim, n=4):
    """Barycentric points (nq, dim+1) and weights (sum = 1) on the reference simplex: tensor Gauss-Legendre in collapsed
    (Duffy) coordinates, exact for total degree <= 2n - 1 - (dim - 1)."""
    g, w = np.polynomial.legendre.leggauss(n)
    g, w = 0.5 * (g + 1), 0.5 * w
    pts, wts = [], []
    if dim == 2:
        for a, wa in zip(g, w):
            for b, wb in zip(g, w):
                x1, x2 = a, b * (1 - a)
                pts.append([1 - x1 - x2, x1, x2])
                wts.append(wa * wb * (1 - a))
    else:
        for a, wa in zip(g, w):
            for b, wb in zip(g, w):
                for c, wc in zip(g, w):
                    x1, x2, x3 = a, b * (1 - a), c * (1 - a) * (1 - b)
                    pts.append([1 - x1 - x2 - x3, x1, x2, x3])
                    wts.append(wa * wb * wc * (1 - a) ** 2 * (1 - b))
    pts, wts = np.array(pts), np.array(wts)
    return pts, wts / wts.sum()


def p2_basis(lam, edge_table):
    """phi (nq, nb) and dphi/dlambda (nq, nb, nv) at barycentric points lam (nq, nv)."""
    nq, nv = lam.shape
    nb = nv + len(edge_table)
    phi = np.zeros((nq, nb))
    dphi = np.zeros((nq, nb, nv))
    for a in range(nv):
        phi[:, a] = lam[:, a] * (2 * lam[:, a] - 1)
        dphi[:, a, a] = 4 * lam[:, a] - 1
    for e, (a, b) in enumerate(edge_table):
        phi[:, nv + e] = 4 * lam[:, a] * lam[:, b]
        dphi[:, nv + e, a] = 4 * lam[:, b]
        dphi[:, nv + e, b] = 4 * lam[:, a]
    return phi, dphi


def _scatter(n, dofs, Ae):
    k = dofs.shape[1]
    rows = np.repeat(dofs[:, :, None], k, axis=2)
    cols = np.repeat(dofs[:, None, :], k, axis=1)
    M = sp.coo_matrix((Ae.ravel(), (rows.ravel(), cols.ravel())), shape=(n, n)).tocsr()
    M.sum_duplicates()
    M.sort_indices()
    return M


# ----------------------------------------------------------------------------- operators
def assemble_A_C(x, cells, cell_dofs, n_dof, c):
    """(A, C) on the P2 space; c = P1 nodal sound speed (complex allowed: c**2 is the complex square)."""
    G, vol = fem.cell_geometry(x, cells)                   # G[c, m, :] = grad(lambda_m)
    lam, w = collapsed_gauss(3)
    phi, dphi = p2_basis(lam, TET_EDGES)
    cq = np.asarray(c, dtype=complex)[cells] @ lam.T        # (nc, nq)
    gphi = np.einsum('qjm,cmd->cqjd', dphi, G)             # physical gradients (nc, nq, 10, 3)
    Ae = -np.einsum('q,cq,cqjd,cqkd->cjk', w, cq ** 2, gphi, gphi) * vol[:, None, None]
    Ce = np.einsum('q,qj,qk->jk', w, phi, phi)[None] * vol[:, None, None]
    return _scatter(n_dof, cell_dofs, Ae), _scatter(n_dof, cell_dofs, Ce.astype(complex))


def assemble_B(x, n_dof, terms):
    """terms: list of (facets (nf, 3) vertex ids, fdofs (nf, 6), Z, c nodal P1)."""
    B = sp.csr_matrix((n_dof, n_dof), dtype=complex)
    lam, w = collapsed_gauss(2)
    phi, _ = p2_basis(lam, TRI_EDGES)
    for facets, fdofs, Z, c in terms:
        area = fem.facet_measure(x, facets)
        cq = np.asarray(c, dtype=complex)[facets] @ lam.T   # (nf, nq)
        Be = (1j / Z) * np.einsum('q,fq,qj,qk->fjk', w, cq, phi, phi) * area[:, None, None]
        B = B + _scatter(n_dof, fdofs, Be)
    B.sort_indices()
    return B.tocsr()


def distributed_left(x, cells, cell_dofs, n_dof, gamma, h, q_0, u_b):
    """a_j = int (gamma-1) q0/u_b phi_j h dx, gamma scalar or P1 nodal, h P1 nodal."""
    _, vol = fem.cell_geometry(x, cells)
    lam, w = collapsed_gauss(3)
    phi, _ = p2_basis(lam, TET_EDGES)
    hq = np.asarray(h, dtype=complex)[cells] @ lam.T
    gq = (np.asarray(gamma, dtype=complex)[cells] @ lam.T - 1) if np.ndim(gamma) else (gamma - 1) * np.ones_like(hq)
    loc = np.einsum('q,cq,cq,qj->cj', w, gq, hq, phi) * vol[:, None] * (q_0 / u_b)
    a = np.zeros(n_dof, dtype=complex)
    np.add.at(a, cell_dofs.ravel(), loc.ravel())
    return a


def distributed_right(x, cells, cell_dofs, n_dof, rho, w_field):
    """b_k = int (n_r.grad phi_k) w / rho dx (n_r = e_z in 3-D)."""
    G, vol = fem.cell_geometry(x, cells)
    lam, w = collapsed_gauss(3)
    _, dphi = p2_basis(lam, TET_EDGES)
    rq = np.asarray(rho, dtype=complex)[cells] @ lam.T
    wq = np.asarray(w_field, dtype=complex)[cells] @ lam.T
    gz = np.einsum('qjm,cm->cqj', dphi, G[:, :, 2])         # d phi_j / dz at the quadrature points
    loc = np.einsum('q,cq,cqj->cj', w, wq / rq, gz) * vol[:, None]
    b = np.zeros(n_dof, dtype=complex)
    np.add.at(b, cell_dofs.ravel(), loc.ravel())
    return b


# ----------------------------------------------------------------------------- pointwise flames (flame_matrices.py:139-160, degree 2)
def pointwise_left(x, cells, cell_dofs, n_dof, cell_tags, flame, h_cell, gamma, q_0, u_b):
    """a_j = (gamma-1) q0/u_b int_{Omega_f} h phi_j dx with h DG0 and phi_j quadratic (flame_matrices.py:141): on a tetrahedron
    int phi_vertex = -|K|/20, int phi_edge = |K|/5 -- integrated here with the rule of the other forms."""
    _, vol = fem.cell_geometry(x, cells)
    lam, w = collapsed_gauss(3)
    phi, _ = p2_basis(lam, TET_EDGES)
    ints = np.einsum('q,qj->j', w, phi)                        # reference integrals of the 10 basis functions (sum to 1)
    m = np.flatnonzero(np.asarray(cell_tags) == flame)
    loc = (gamma - 1) * q_0 / u_b * np.asarray(h_cell, dtype=complex)[m, None] * vol[m, None] * ints[None, :]
    a = np.zeros(n_dof, dtype=complex)
    np.add.at(a, cell_dofs[m].ravel(), loc.ravel())
    return a


def pointwise_right(x, cells, cell_dofs, point, rho_u, cell=None):
    """b_k = (grad phi_k . e_z)(x_r) / rho_u on the 10 dofs of the cell holding x_r (flame_matrices.py:144-156, degree 2):
    (dofs (10,), values (10,)).  ``cell``: the owner cell (default: lowest index containing the point)."""
    c = fem.locate_cell(x, cells, point) if cell is None else int(cell)
    G, _ = fem.cell_geometry(x, cells[c:c + 1])                # G[0, m, :] = grad lambda_m
    X = x[cells[c]]
    J = (X[1:] - X[0]).T
    l123 = np.linalg.solve(J, np.asarray(point, dtype=float).reshape(-1)[:3] - X[0])
    lam = np.concatenate([[1 - l123.sum()], l123])[None, :]
    _, dphi = p2_basis(lam, TET_EDGES)                         # d phi_j / d lambda_m at the point: (1, 10, 4)
    grad = dphi[0] @ G[0]                                      # (10, 3)
    return cell_dofs[c].astype(np.int32), (grad[:, 2] / rho_u).astype(complex)
