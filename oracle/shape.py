"""CPU restatement of the shape-derivative workflow (test infrastructure only -- see oracle/__init__.py).

Reference: helmholtz_x/shape_derivatives.py:12-37 -- ``G_neu = div(p_adj_conj * c**2 * grad(p_dir))`` integrated against
``inner(V_ffd, normal)`` over the tagged boundary -- with the displacement fields of ``ffd_displacement_vector`` (:39-77), the
lattice of shape_derivatives_utils.py:92-189 (``FFDCylindrical``) and the table normalisation of :38-90.

PINNED to the reference's own data (tests/test_oracle_goldens.py): the 36 normalised shape derivatives of
ShapeSensitivities/RijkeFFD/ShapeDerivatives/normalized.txt are reproduced on the mesh the reference's run wrote into its
result files (tests/golden/rijkeffd_shape/), from this oracle's degree-2 eigenvectors.  The two gmsh calls of the reference
are replaced by what they return for that case: ``mesh.getNodes`` = the mesh's node coordinates, ``model.getNormal`` on the
lateral surface of ``occ.addCylinder`` = the radial unit vector.  Closed-form patch tests pin the functional besides.

Plain NumPy loops over facets and quadrature points, P1 and P2 (vertex + edge dofs, helmholtz_x.fem.p2_dofmap layout)."""
from math import comb

import numpy as np

TET_EDGES = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]


def _strang_fix_cowper_deg6():
    pts = []
    for a, b, w in ((0.063089014491502, 0.873821971016996, 0.050844906370207),
                    (0.249286745170910, 0.501426509658179, 0.116786275726379)):
        pts += [(b, a, a, w), (a, b, a, w), (a, a, b, w)]
    a, b, c, w = 0.053145049844817, 0.310352451033784, 0.636502499121399, 0.082851075618374
    pts += [(a, b, c, w), (b, c, a, w), (c, a, b, w), (a, c, b, w), (c, b, a, w), (b, a, c, w)]
    return np.array(pts)


def _eval(degree, u, lam, gl):
    """value, gradient (3,), Laplacian of a P1 / P2 field on one tetrahedron at barycentric point lam."""
    if degree == 1:
        return np.dot(lam, u[:4]), gl.T @ u[:4], 0.0
    val = sum(u[i] * lam[i] * (2 * lam[i] - 1) for i in range(4))
    grad = sum(u[i] * (4 * lam[i] - 1) * gl[i] for i in range(4))
    lap = sum(u[i] * 4 * np.dot(gl[i], gl[i]) for i in range(4))
    for e, (i, j) in enumerate(TET_EDGES):
        val = val + u[4 + e] * 4 * lam[i] * lam[j]
        grad = grad + u[4 + e] * 4 * (lam[i] * gl[j] + lam[j] * gl[i])
        lap = lap + u[4 + e] * 8 * np.dot(gl[i], gl[j])
    return val, grad, lap


def boundary_functional_weights(x, cells, facets, facet_cells, p, q, c, degree=1, cell_dofs=None, c_is_cellwise=False):
    """The functional is linear in the displacement field: per listed facet f the outward unit normal n_f (nf, 3) and
    complex weights W (nf, 3) with   int_f (V . n) G ds = sum_k (V[facets[f, k]] . n_f) W[f, k],
    G = (c^2 grad q + 2 q c grad c) . grad p + q c^2 lap p   evaluated from the adjacent cell's basis functions."""
    quad = _strang_fix_cowper_deg6()
    facets = np.asarray(facets)
    W = np.zeros((facets.shape[0], 3), dtype=complex)
    N = np.zeros((facets.shape[0], 3))
    for k, (f, cell) in enumerate(zip(facets, np.asarray(facet_cells))):
        v = cells[cell]
        X = x[v]
        J = X[1:] - X[0]
        G = np.linalg.inv(J)                       # columns: grad lam_1..3
        gl = np.vstack([-G.sum(axis=1), G.T])
        loc = [int(np.flatnonzero(v == fv)[0]) for fv in f]
        opp = [i for i in range(4) if i not in loc][0]
        nrm = np.cross(X[loc[1]] - X[loc[0]], X[loc[2]] - X[loc[0]])
        area = 0.5 * np.linalg.norm(nrm)
        nrm = nrm / np.linalg.norm(nrm)
        if np.dot(nrm, X[loc[0]] - X[opp]) < 0:
            nrm = -nrm
        dofs = v if degree == 1 else cell_dofs[cell]
        pu, qu = p[dofs], q[dofs]
        cu = np.full(4, c[cell]) if c_is_cellwise else c[v]
        for b0, b1, b2, w in quad:
            lam = np.zeros(4)
            lam[loc[0]], lam[loc[1]], lam[loc[2]] = b0, b1, b2
            pv, gp, lp = _eval(degree, pu, lam, gl)
            qv, gq, _ = _eval(degree, qu, lam, gl)
            cv, gc, _ = _eval(1, cu, lam, gl)
            Gn = np.dot(cv * cv * gq + 2 * qv * cv * gc, gp) + qv * cv * cv * lp
            W[k] += (w * area * Gn) * np.array([b0, b1, b2])
        N[k] = nrm
    return N, W


def boundary_functional(x, cells, facets, facet_cells, V, p, q, c, degree=1, cell_dofs=None, c_is_cellwise=False,
                        weights=None):
    """sum over the listed facets of  int_f (V . n) [(c^2 grad q + 2 q c grad c) . grad p + q c^2 lap p] ds.
    ``weights``: a precomputed (normals, W) pair of boundary_functional_weights (many fields V, same p, q, c)."""
    N, W = weights if weights is not None else boundary_functional_weights(x, cells, facets, facet_cells, p, q, c, degree,
                                                                          cell_dofs, c_is_cellwise)
    Vn = np.einsum('fkd,fd->fk', np.asarray(V)[np.asarray(facets)], N)
    return complex(np.sum(Vn * W))


# ----------------------------------------------------------------------------- FFD lattice (cylinder)
class CylindricalLattice:
    """shape_derivatives_utils.py:92-189 (FFDCylindrical) from node coordinates: the lattice spans
    [min rho, max rho] x [min phi, min phi + 2 pi] x [min z, max z] of the nodes, phi = atan2(y, x) (dolfinx_utils.py:15-20)."""

    def __init__(self, x, l, m, n):
        self.l, self.m, self.n = l, m, n
        rho, phi, z = np.hypot(x[:, 0], x[:, 1]), np.arctan2(x[:, 1], x[:, 0]), x[:, 2]
        self.P0 = np.array([rho.min(), phi.min(), z.min()])
        self.dr, self.dphi, self.dz = rho.max() - rho.min(), 2 * np.pi, z.max() - z.min()

    def stu(self, x):
        rho, phi, z = np.hypot(x[:, 0], x[:, 1]), np.arctan2(x[:, 1], x[:, 0]), x[:, 2]
        return (rho - self.P0[0]) / self.dr, (phi - self.P0[1]) / self.dphi, (z - self.P0[2]) / self.dz


def _bernstein(N, k, t):
    return comb(N, k) * np.power(1 - t, N - k) * np.power(t, k)


def ffd_displacement_cylinder(x, surface_vertices, lattice, i, j, k):
    """shape_derivatives.py:39-77 for the lateral surface of a z-axis cylinder: at every vertex of the tagged surface the
    trivariate Bernstein weight of control point (i, j, k) (:60-62) times the CAD surface normal -- for the lateral face of
    gmsh's ``occ.addCylinder`` the outward radial unit vector -- and zero at every other vertex: (n_vertices, 3)."""
    s, t, u = lattice.stu(x)
    value = _bernstein(lattice.l - 1, i, s) * _bernstein(lattice.m - 1, j, t) * _bernstein(lattice.n - 1, k, u)
    rho = np.hypot(x[:, 0], x[:, 1])
    V = np.zeros_like(x)
    sv = np.asarray(surface_vertices)
    V[sv, 0] = value[sv] * x[sv, 0] / rho[sv]
    V[sv, 1] = value[sv] * x[sv, 1] / rho[sv]
    return V


def nonaxisymmetric_normalize(table):
    """shape_derivatives_utils.py:38-90 on a (n_zeta, n_phi) complex array: real parts over the largest |real part|,
    imaginary parts over the largest |imaginary part| of the table."""
    table = np.asarray(table)
    return table.real / np.abs(table.real).max() + 1j * table.imag / np.abs(table.imag).max()


def normalize_adjoint(omega, p_dir, p_adj, B, C, a, b, dftf):
    """eigenvectors.py:125-177: p_adj / meas with meas = vector_matrix_vector(p_adj, dL/domega, p_dir), where the bracket is
    petsc4py's ``p_adj.dot(y)`` = VecDot(p_adj, y) = y^H p_adj (petsc4py_utils.py:42-88), y = (B + 2 omega C - FTF'(omega) a b^T) p_dir."""
    y = 2 * omega * (C @ p_dir) - dftf(omega) * a * (b @ p_dir)
    if B is not None:
        y = y + B @ p_dir
    return p_adj / np.vdot(y, p_adj)
