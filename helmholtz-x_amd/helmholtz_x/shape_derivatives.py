"""Hadamard-form shape derivative of an eigenvalue: the reference's ``shapeDerivativesFFD`` and ``ffd_displacement_vector``
(helmholtz_x/shape_derivatives.py:12-77) with the boundary functional on the GPU.

    d omega / dV  =  int_Gamma (V . n) div( conj(p_adj) c^2 grad(p_dir) ) ds

with the adjoint eigenvector normalised by ``normalize_adjoint`` (eigenvectors.py:125-177), exactly as the reference's lines
17-20 do.  The reference builds the displacement fields V from a free-form-deformation lattice and two calls into gmsh's CAD
kernel: ``mesh.getNodes`` (the surface's node coordinates -- the mesh has them) and ``model.getNormal`` (the CAD surface
normal at those nodes).  gmsh is not part of this path, so the normal is an argument: ``normals='radial'`` is what
``getNormal`` returns on the lateral face of ``occ.addCylinder`` (the reference's RijkeFFD case, main_shape.py:32),
``normals='vertex'`` (default) uses area-weighted facet normals of the mesh, and an (n_vertices, 3) array or a callable
x -> normals is taken as given.  Degree 1 and degree 2.  Pinned by the 36 normalised shape derivatives the reference commits
(ShapeSensitivities/RijkeFFD/ShapeDerivatives/normalized.txt; tests/test_gpu_parity.py).
"""
import ctypes as C

import numpy as np
import torch

from . import _hx
from .fem import Function, FunctionSpace
from .petsc import to_device

# Dunavant's 12-point rule on the triangle, exact to degree 6 (barycentric points, weights summing to 1)
_D6 = []
for _a, _b, _w in ((0.063089014491502, 0.873821971016996, 0.050844906370207),
                   (0.249286745170910, 0.501426509658179, 0.116786275726379)):
    _D6 += [(_b, _a, _a, _w), (_a, _b, _a, _w), (_a, _a, _b, _w)]
_a, _b, _c, _w = 0.053145049844817, 0.310352451033784, 0.636502499121399, 0.082851075618374
_D6 += [(_a, _b, _c, _w), (_b, _c, _a, _w), (_c, _a, _b, _w), (_a, _c, _b, _w), (_c, _b, _a, _w), (_b, _a, _c, _w)]
DUNAVANT6 = np.ascontiguousarray(np.array(_D6, dtype=np.float64))


class _FacetFunctional:
    """Device-resident operands of the functional for one tagged boundary and one pair (p, q, c): every further displacement
    field costs one upload of V and one launch of hx_shape_derivative_facets."""

    def __init__(self, mesh, facet_tags, tag, p_dir, p_adj_conj, c, degree):
        if mesh.topology.dim != 3:
            raise NotImplementedError("the shape-derivative functional is provided on tetrahedral meshes (the reference's RijkeFFD case)")
        if degree not in (1, 2):
            raise NotImplementedError("degree 1 or 2")
        self.mesh, self.degree = mesh, int(degree)
        self.asm = mesh.assembler()
        self.ids = np.asarray(facet_tags.find(tag), dtype=np.int64)
        if self.ids.size == 0:
            return
        self.facets = to_device(np.ascontiguousarray(mesh.facets[self.ids], dtype=np.int32))
        self.fcell = to_device(np.ascontiguousarray(mesh.facet_cells[self.ids], dtype=np.int32))

        def dofs(f):
            a = f.x.array if isinstance(f, Function) else np.asarray(f)
            return to_device(np.ascontiguousarray(a, dtype=np.complex128))

        self.cell_dofs = to_device(FunctionSpace(mesh, "CG", 2).cell_dofs) if degree == 2 else None
        self.cf = dofs(c)
        self.cellwise = isinstance(c, Function) and c.function_space.kind == "DG0"
        self.quad = to_device(DUNAVANT6)
        self.pd, self.qd = dofs(p_dir), dofs(p_adj_conj)
        self.out = torch.empty(self.ids.size, dtype=torch.complex128, device=self.asm.x.device)

    def __call__(self, V):
        if self.ids.size == 0:
            return 0.0 + 0.0j
        Vd = to_device(np.ascontiguousarray(np.asarray(V, dtype=np.float64).reshape(self.mesh.num_vertices, 3)))
        _hx.check(_hx.lib.hx_shape_derivative_facets(self.degree, self.ids.size, _hx.ptr(self.asm.x), _hx.ptr(self.asm.cells),
                                                     _hx.ptr(self.cell_dofs), _hx.ptr(self.facets), _hx.ptr(self.fcell),
                                                     _hx.ptr(self.pd), _hx.ptr(self.qd), _hx.ptr(self.cf), int(self.cellwise),
                                                     _hx.ptr(Vd), DUNAVANT6.shape[0], _hx.ptr(self.quad), _hx.ptr(self.out),
                                                     _hx.stream()))
        vals = self.out.cpu().numpy()                # O(n^(2/3)) numbers; summed in facet order
        return complex(np.sum(vals.real) + 1j * np.sum(vals.imag))


def boundary_functional(mesh, facet_tags, tag, V, p_dir, p_adj_conj, c, degree=1):
    """int_{Gamma_tag} (V . n) div(q c^2 grad p) ds  with p = p_dir, q = p_adj_conj (Functions of the degree-``degree``
    space, or dof arrays), c a P1 or DG0 Function, V an (n_vertices, 3) array of nodal displacement vectors.
    Device kernel hx_shape_derivative_facets + an ordered sum."""
    return _FacetFunctional(mesh, facet_tags, tag, p_dir, p_adj_conj, c, degree)(V)


def shape_derivative(geometry, physical_facet_tag, V, omega_dir, p_dir, p_adj, c, acousticMatrices, FlameMatrix, degree=1):
    """One entry of the reference's ``shapeDerivativesFFD`` loop (shape_derivatives.py:12-37) for a given displacement field
    V: normalise the adjoint (:17), conjugate it (:18), integrate (V . n) div(conj(p_adj) c^2 grad p_dir) over the tagged
    boundary (:20, :33-35)."""
    from .eigenvectors import normalize_adjoint
    p_adj_norm = normalize_adjoint(omega_dir, p_dir, p_adj, acousticMatrices, FlameMatrix)
    q = np.conj(p_adj_norm.x.array)
    return boundary_functional(geometry.mesh, geometry.facet_tags, physical_facet_tag, V, p_dir, q, c, degree=degree)


def vertex_normals(mesh, facet_tags, tag):
    """Area-weighted unit normals at the vertices of the tagged boundary (outward), zero elsewhere: (n_vertices, 3)."""
    x = mesh.geometry.x
    ids = np.asarray(facet_tags.find(tag), dtype=np.int64)
    fv = mesh.facets[ids]
    cells = mesh.topology.cells[mesh.facet_cells[ids]]
    nrm = np.cross(x[fv[:, 1]] - x[fv[:, 0]], x[fv[:, 2]] - x[fv[:, 0]])
    inside = x[cells].mean(axis=1)
    flip = np.einsum("ij,ij->i", nrm, x[fv[:, 0]] - inside) < 0
    nrm[flip] *= -1.0
    out = np.zeros_like(x)
    for k in range(3):
        np.add.at(out, fv[:, k], nrm)
    ln = np.linalg.norm(out, axis=1)
    out[ln > 0] /= ln[ln > 0][:, None]
    return out


def ffd_bernstein_field(mesh, facet_tags, tag, box_min, box_max, lmn, ijk):
    """Displacement field of FFD control point (i, j, k) of an l x m x n lattice spanning [box_min, box_max]: the trivariate
    Bernstein weight of shape_derivatives.py:60-62 at every boundary vertex times the outward normal there."""
    from math import comb
    x = mesh.geometry.x
    stu = (x - np.asarray(box_min)) / (np.asarray(box_max) - np.asarray(box_min))
    w = np.ones(x.shape[0])
    for d in range(3):
        n_, k_ = lmn[d] - 1, ijk[d]
        w *= comb(n_, k_) * np.power(1 - stu[:, d], n_ - k_) * np.power(stu[:, d], k_)
    return w[:, None] * vertex_normals(mesh, facet_tags, tag)


def _surface_normals(mesh, facet_tags, tag, normals):
    if normals is None or (isinstance(normals, str) and normals == "vertex"):
        return vertex_normals(mesh, facet_tags, tag)
    x = mesh.geometry.x
    if isinstance(normals, str):
        if normals != "radial":
            raise ValueError("normals: 'vertex', 'radial', an (n_vertices, 3) array or a callable")
        rho = np.hypot(x[:, 0], x[:, 1])
        out = np.zeros_like(x)
        on = rho > 0
        out[on, 0], out[on, 1] = x[on, 0] / rho[on], x[on, 1] / rho[on]
        return out
    return np.asarray(normals(x) if callable(normals) else normals, dtype=np.float64).reshape(x.shape[0], 3)


def ffd_displacement_vector(geometry, FFDLattice, surface_physical_tag, i, j, k,
                            includeBoundary=True, returnParametricCoord=True, tol=1e-6, deg=1, normals=None):
    """Displacement field of control point (i, j, k) (shape_derivatives.py:39-77): at every vertex of the tagged surface
    (rim vertices included, as ``includeBoundary=True`` does) the trivariate Bernstein weight of :60-62 at the vertex's lattice
    coordinates times the surface normal there; zero at every other vertex.  Returns the (n_vertices, 3) nodal array of the
    P1 vector field (the reference returns a DOLFINx vector Function holding these values)."""
    from math import comb
    if deg != 1:
        raise NotImplementedError("P1 displacement fields (the reference's only use: shape_derivatives.py:31)")
    mesh, facet_tags = geometry.mesh, geometry.facet_tags
    x = mesh.geometry.x
    ids = np.asarray(facet_tags.find(surface_physical_tag), dtype=np.int64)
    sv = np.unique(mesh.facets[ids])
    L = FFDLattice
    s, t, u = L.calcSTU(x[sv])
    value = comb(L.l - 1, i) * np.power(1 - s, L.l - 1 - i) * np.power(s, i) * \
        comb(L.m - 1, j) * np.power(1 - t, L.m - 1 - j) * np.power(t, j) * \
        comb(L.n - 1, k) * np.power(1 - u, L.n - 1 - k) * np.power(u, k)
    V = np.zeros_like(x)
    V[sv] = value[:, None] * _surface_normals(mesh, facet_tags, surface_physical_tag, normals)[sv]
    return V


def shapeDerivativesFFD(geometry, lattice, physical_facet_tag, omega_dir, p_dir, p_adj, c, acousticMatrices, FlameMatrix,
                        normals=None):
    """{zeta: {phi: d omega / d P_(l-1, phi, zeta)}}: the reference's loop over the outer ring of control points
    (shape_derivatives.py:12-37).  As there, ``p_adj`` is normalised and conjugated IN PLACE (:15-16 alias it)."""
    from .eigenvectors import normalize_adjoint
    from .petsc4py_utils import conjugate_function
    p_adj_norm = normalize_adjoint(omega_dir, p_dir, p_adj, acousticMatrices, FlameMatrix)
    p_adj_conj = conjugate_function(p_adj_norm)
    degree = 2 if p_dir.function_space.kind == "P2" else 1
    functional = _FacetFunctional(geometry.mesh, geometry.facet_tags, physical_facet_tag, p_dir, p_adj_conj, c, degree)
    derivatives = {}
    i = lattice.l - 1
    for zeta in range(lattice.n):
        derivatives[zeta] = {}
        for phi in range(lattice.m):
            V_ffd = ffd_displacement_vector(geometry, lattice, physical_facet_tag, i, phi, zeta, deg=1, normals=normals)
            derivatives[zeta][phi] = functional(V_ffd)
    return derivatives
