"""Hadamard-form shape derivative of an eigenvalue: the boundary functional of the reference's ``shapeDerivativesFFD``
(helmholtz_x/shape_derivatives.py:12-37) on the GPU.

    d omega / dV  =  int_Gamma (V . n) div( conj(p_adj) c^2 grad(p_dir) ) ds

with the adjoint eigenvector normalised by ``normalize_adjoint`` (eigenvectors.py:125-177), exactly as the reference's lines
17-20 do.  The reference builds the displacement fields V from a free-form-deformation lattice with gmsh's CAD model
(``ffd_displacement_vector``, :39-77: parametric coordinates and surface normals of the gmsh surface) -- mesh generation is
out of scope here, so the displacement field is an ARGUMENT: any nodal vector field on the
boundary (``ffd_bernstein_field`` below restates the lattice weight of :60-62 for fields given by a Bernstein control
point, with the mesh's own facet normals standing in for gmsh's).  Degree 1 and degree 2 (the reference's RijkeFFD runs).
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


def boundary_functional(mesh, facet_tags, tag, V, p_dir, p_adj_conj, c, degree=1):
    """int_{Gamma_tag} (V . n) div(q c^2 grad p) ds  with p = p_dir, q = p_adj_conj (Functions of the degree-``degree``
    space, or dof arrays), c a P1 or DG0 Function, V an (n_vertices, 3) array of nodal displacement vectors.
    Device kernel hx_shape_derivative_facets + an ordered sum."""
    if mesh.topology.dim != 3:
        raise NotImplementedError("the shape-derivative functional is provided on tetrahedral meshes (the reference's RijkeFFD case)")
    asm = mesh.assembler()
    ids = np.asarray(facet_tags.find(tag), dtype=np.int64)
    if ids.size == 0:
        return 0.0 + 0.0j
    facets = to_device(np.ascontiguousarray(mesh.facets[ids], dtype=np.int32))
    fcell = to_device(np.ascontiguousarray(mesh.facet_cells[ids], dtype=np.int32))

    def dofs(f):
        a = f.x.array if isinstance(f, Function) else np.asarray(f)
        return to_device(np.ascontiguousarray(a, dtype=np.complex128))

    cell_dofs = None
    if degree == 2:
        cell_dofs = to_device(FunctionSpace(mesh, "CG", 2).cell_dofs)
    elif degree != 1:
        raise NotImplementedError("degree 1 or 2")
    cf = dofs(c)
    cellwise = isinstance(c, Function) and c.function_space.kind == "DG0"
    Vd = to_device(np.ascontiguousarray(np.asarray(V, dtype=np.float64).reshape(mesh.num_vertices, 3)))
    quad = to_device(DUNAVANT6)
    out = torch.empty(ids.size, dtype=torch.complex128, device=asm.x.device)
    pd, qd = dofs(p_dir), dofs(p_adj_conj)       # (named: the device copies must outlive the launch)
    _hx.check(_hx.lib.hx_shape_derivative_facets(int(degree), ids.size, _hx.ptr(asm.x), _hx.ptr(asm.cells), _hx.ptr(cell_dofs),
                                                 _hx.ptr(facets), _hx.ptr(fcell), _hx.ptr(pd), _hx.ptr(qd),
                                                 _hx.ptr(cf), int(cellwise), _hx.ptr(Vd), DUNAVANT6.shape[0], _hx.ptr(quad),
                                                 _hx.ptr(out), _hx.stream()))
    vals = out.cpu().numpy()                     # O(n^(2/3)) numbers; summed in facet order
    return complex(np.sum(vals.real) + 1j * np.sum(vals.imag))


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
