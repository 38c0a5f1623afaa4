"""CPU restatement of the shape-derivative boundary functional (test infrastructure only -- see oracle/__init__.py).

Reference: helmholtz_x/shape_derivatives.py:12-37 -- ``G_neu = div(p_adj_conj * c**2 * grad(p_dir))`` integrated against
``inner(V_ffd, normal)`` over the tagged boundary.  PARITY UNPINNED against the reference: it commits no value of this
functional that can be recomputed here (the FFD displacement fields need gmsh's CAD model, and the degree-2 run behind
Results/ShapeDerivatives was made on a mesh that is not in the repository).  The restatement is pinned by closed-form patch
tests instead (tests/test_oracle_goldens.py): fields for which the integral is known analytically.

Plain NumPy loops over facets and quadrature points, P1 and P2 (vertex + edge dofs, helmholtz_x.fem.p2_dofmap layout)."""
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


def boundary_functional(x, cells, facets, facet_cells, V, p, q, c, degree=1, cell_dofs=None, c_is_cellwise=False):
    """sum over the listed facets of  int_f (V . n) [(c^2 grad q + 2 q c grad c) . grad p + q c^2 lap p] ds."""
    quad = _strang_fix_cowper_deg6()
    total = 0.0 + 0.0j
    for f, cell in zip(np.asarray(facets), np.asarray(facet_cells)):
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
        Vn = V[v[loc]] @ nrm
        acc = 0.0 + 0.0j
        for b0, b1, b2, w in quad:
            lam = np.zeros(4)
            lam[loc[0]], lam[loc[1]], lam[loc[2]] = b0, b1, b2
            pv, gp, lp = _eval(degree, pu, lam, gl)
            qv, gq, _ = _eval(degree, qu, lam, gl)
            cv, gc, _ = _eval(1, cu, lam, gl)
            Gn = np.dot(cv * cv * gq + 2 * qv * cv * gc, gp) + qv * cv * cv * lp
            acc += w * (b0 * Vn[0] + b1 * Vn[1] + b2 * Vn[2]) * Gn
        total += area * acc
    return total
