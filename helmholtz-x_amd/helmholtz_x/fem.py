"""Minimal stand-ins for the DOLFINx objects the hot path receives and the
device-side P1 assembly driver.

DOLFINx is not a dependency of this package: the reference uses it for meshes, P1 function spaces and
form assembly (acoustic_matrices.py:25,101-123; flame_matrices.py:24,63,141-156,199-200).  Here
  * ``Mesh`` / ``MeshTags`` / ``FunctionSpace`` / ``Function`` expose the attributes the reference
    code touches (``mesh.topology.dim``, ``mesh.geometry.x/.dim/.dofmap``, ``tags.indices/.values/.find``,
    ``f.x.array``, ``f.name``, ``f.copy()``, ``f.function_space``);
  * ``P1Assembler`` owns the device copies of the mesh, the vertex->cell adjacency and the CSR pattern,
    and launches the libhx assembly kernels (one thread per matrix row, deterministic).
"""
import ctypes as C
from itertools import combinations

import numpy as np
import torch

from . import _hx
from .petsc import CSR, Mat, to_device


# ----------------------------------------------------------------------------- host-side objects
class _Topology:
    def __init__(self, dim, cells):
        self.dim = dim
        self.cells = cells

    def cell_name(self):
        return {1: "interval", 2: "triangle", 3: "tetrahedron"}[self.dim]

    def create_connectivity(self, d0, d1):
        pass


class _Geometry:
    def __init__(self, x, dim, dofmap):
        self.x = x
        self.dim = dim
        self.dofmap = dofmap


class _Comm:
    rank, size = 0, 1

    def allreduce(self, v, op=None):
        return v


class Mesh:
    """Simplicial mesh: x (n,3) float64, cells (nc, d+1) int32; P1 dofmap == vertex ids.

    ``facets`` (nf, d) lists the tagged / exterior facets by vertex ids; a facet *index* (as found in
    ``MeshTags.indices`` of dimension d-1) is a row of that table; ``facet_cells`` is the adjacent cell."""

    def __init__(self, x, cells, facets=None, facet_cells=None):
        x = np.ascontiguousarray(x, dtype=np.float64)
        if x.shape[1] != 3:
            x = np.concatenate([x, np.zeros((x.shape[0], 3 - x.shape[1]))], axis=1)
        cells = np.ascontiguousarray(cells, dtype=np.int32)
        dim = cells.shape[1] - 1
        self.topology = _Topology(dim, cells)
        self.geometry = _Geometry(x, dim, cells)
        self.comm = _Comm()
        if facets is None:
            facets, facet_cells = exterior_facets(cells)
        self.facets = np.ascontiguousarray(facets, dtype=np.int32)
        self.facet_cells = np.ascontiguousarray(facet_cells, dtype=np.int32)
        self._assembler = None

    @property
    def num_vertices(self):
        return self.geometry.x.shape[0]

    @property
    def num_cells(self):
        return self.topology.cells.shape[0]

    def assembler(self):
        if self._assembler is None:
            self._assembler = P1Assembler(self)
        return self._assembler


class MeshTags:
    """dolfinx.mesh.meshtags stand-in: sorted entity ``indices`` with ``values``."""

    def __init__(self, mesh, dim, indices, values):
        indices = np.asarray(indices, dtype=np.int32)
        values = np.asarray(values, dtype=np.int32)
        order = np.argsort(indices, kind="stable")
        self.mesh, self.dim = mesh, dim
        self.indices, self.values = indices[order], values[order]

    def find(self, tag):
        return self.indices[self.values == tag]


def exterior_facets(cells):
    """(facets (nf, d) with ascending vertex ids, adjacent cell (nf,)) of a simplicial mesh."""
    cells = np.asarray(cells)
    k = cells.shape[1]
    if k == 2:
        fl = cells.reshape(-1, 1)
        owner = np.repeat(np.arange(cells.shape[0]), 2)
    else:
        combos = list(combinations(range(k), k - 1))
        fl = np.sort(np.concatenate([cells[:, c] for c in combos], axis=0), axis=1)
        owner = np.tile(np.arange(cells.shape[0]), len(combos))
    uniq, first, counts = np.unique(fl, axis=0, return_index=True, return_counts=True)
    m = counts == 1
    return uniq[m].astype(np.int32), owner[first[m]].astype(np.int32)


class FunctionSpace:
    def __init__(self, mesh, family="Lagrange", degree=1):
        fam = family.upper()
        if fam in ("LAGRANGE", "CG", "P") and degree == 1:
            self.kind, self.size = "P1", mesh.num_vertices
        elif fam in ("DG", "DISCONTINUOUS LAGRANGE") and degree == 0:
            self.kind, self.size = "DG0", mesh.num_cells
        else:
            raise NotImplementedError("only P1 and DG0 spaces are on the hot path (got %s %d); degree 2 is a "
                                      "'next' row of SURVEY.md section 8f" % (family, degree))
        self.mesh = mesh
        self.dofmap = mesh.topology.cells if self.kind == "P1" else np.arange(mesh.num_cells, dtype=np.int32)[:, None]

    def tabulate_dof_coordinates(self):
        if self.kind == "P1":
            return self.mesh.geometry.x
        return self.mesh.geometry.x[self.mesh.topology.cells].mean(axis=1)


def functionspace(mesh, element):
    return FunctionSpace(mesh, element[0], element[1])


class _X:
    """``Function.x``: the dof array (host numpy complex128, like a complex DOLFINx build)."""

    def __init__(self, array):
        self.array = array

    def scatter_forward(self):
        pass

    @property
    def petsc_vec(self):
        from .petsc import Vec
        return Vec(self.array)


class Function:
    def __init__(self, V, name="f"):
        self.function_space = V
        self.x = _X(np.zeros(V.size, dtype=np.complex128))
        self.name = name

    def copy(self):
        g = Function(self.function_space, self.name)
        g.x.array[:] = self.x.array
        return g

    def interpolate(self, fn):
        pts = self.function_space.tabulate_dof_coordinates()
        self.x.array[:] = fn(pts.T)


# ----------------------------------------------------------------------------- device assembly
class P1Assembler:
    """Device mesh + adjacency + CSR pattern; wraps the hx_assemble_p1_* kernels."""

    def __init__(self, mesh):
        dev = _hx.require_gpu()
        self.mesh = mesh
        self.dim = mesh.topology.dim
        self.n = mesh.num_vertices
        self.nc = mesh.num_cells
        self.x = to_device(mesh.geometry.x)
        self.cells = to_device(mesh.topology.cells)
        self.v2c_ptr, self.v2c_idx = self._adjacency(self.cells, self.nc, self.dim + 1)
        self.rowptr, self.col = self._pattern(self.cells, self.nc, self.dim + 1, self.v2c_ptr, self.v2c_idx)
        self.pattern_id = id(self)
        self._h_pattern = None
        self.device = dev

    def _adjacency(self, ents, n_ent, k):
        ptr = torch.empty(self.n + 1, dtype=torch.int32, device=self.x.device)
        idx = torch.empty(max(n_ent * k, 1), dtype=torch.int32, device=self.x.device)
        _hx.check(_hx.lib.hx_build_v2e(self.n, n_ent, k, _hx.ptr(ents), _hx.ptr(ptr), _hx.ptr(idx), _hx.stream()))
        return ptr, idx

    def _pattern(self, ents, n_ent, k, ptr, idx):
        rowptr = torch.empty(self.n + 1, dtype=torch.int32, device=self.x.device)
        scratch = torch.empty(max(n_ent * k * k, 1), dtype=torch.int32, device=self.x.device)
        nnz = C.c_int64(0)
        _hx.check(_hx.lib.hx_p1_pattern_count(self.n, n_ent, k, _hx.ptr(ents), _hx.ptr(ptr), _hx.ptr(idx),
                                              _hx.ptr(rowptr), _hx.ptr(scratch), C.byref(nnz), _hx.stream()))
        col = torch.empty(max(nnz.value, 1), dtype=torch.int32, device=self.x.device)[:nnz.value]
        _hx.check(_hx.lib.hx_p1_pattern_fill(self.n, k, _hx.ptr(ptr), _hx.ptr(rowptr), _hx.ptr(scratch),
                                             _hx.ptr(col), _hx.stream()))
        return rowptr, col

    def _csr(self, val):
        out = CSR((self.n, self.n), self.rowptr, self.col, val, pattern_id=self.pattern_id)
        out.symmetric = True            # Galerkin P1 operators are (complex) symmetric: transpose() is the identity
        return out

    @staticmethod
    def _field(f):
        """Function | ndarray | scalar -> (device complex tensor, is_cellwise) for a coefficient."""
        if isinstance(f, Function):
            return to_device(f.x.array.astype(np.complex128)), f.function_space.kind == "DG0"
        raise TypeError("coefficient must be a Function")

    # ---- operators
    def stiffness_mass(self, c=None, want_A=True, want_C=True):
        """(A, C) as Mats: A = -int c^2 grad.grad, C = mass (acoustic_matrices.py:101,121)."""
        A_val = torch.empty(self.col.numel(), dtype=torch.complex128, device=self.x.device) if want_A else None
        C_val = torch.empty(self.col.numel(), dtype=torch.complex128, device=self.x.device) if want_C else None
        cf, cellwise = (self._field(c) if c is not None else (None, False))
        _hx.check(_hx.lib.hx_assemble_p1_stiff_mass(self.dim, self.n, _hx.ptr(self.x), _hx.ptr(self.cells),
                                                    _hx.ptr(self.v2c_ptr), _hx.ptr(self.v2c_idx),
                                                    _hx.ptr(self.rowptr), _hx.ptr(self.col), _hx.ptr(cf),
                                                    int(cellwise), _hx.ptr(A_val), _hx.ptr(C_val), _hx.stream()))
        return (Mat.from_csr(self._csr(A_val)) if want_A else None,
                Mat.from_csr(self._csr(C_val)) if want_C else None)

    def apply_dirichlet(self, mat, dofs, diag=1.0):
        flags = np.zeros(self.n, dtype=np.uint8)
        flags[np.asarray(dofs, dtype=np.int64)] = 1
        d_flags = to_device(flags)
        csr = mat.csr_part()
        _hx.check(_hx.lib.hx_csr_apply_dirichlet(self.n, _hx.ptr(csr.rowptr), _hx.ptr(csr.col), _hx.ptr(csr.val),
                                                 _hx.ptr(d_flags), float(diag), _hx.stream()))

    def robin(self, facet_ids, coef_per_facet, c):
        """B = sum_f coef_f int_F c phi_k phi_j ds over the listed facets (acoustic_matrices.py:71,108-110).
        Returns a Mat on B's own (boundary-only) pattern."""
        facet_ids = np.asarray(facet_ids, dtype=np.int64)
        facets = to_device(self.mesh.facets[facet_ids])
        nf, k = facets.shape
        fcell = to_device(self.mesh.facet_cells[facet_ids])
        coef = to_device(np.asarray(coef_per_facet, dtype=np.complex128))
        ptr, idx = self._adjacency(facets, nf, k)
        rowptr, col = self._pattern(facets, nf, k, ptr, idx)
        cf, cellwise = self._field(c)
        val = torch.empty(col.numel(), dtype=torch.complex128, device=self.x.device)
        _hx.check(_hx.lib.hx_assemble_p1_robin(self.dim, self.n, _hx.ptr(self.x), _hx.ptr(facets), _hx.ptr(ptr),
                                               _hx.ptr(idx), _hx.ptr(rowptr), _hx.ptr(col), _hx.ptr(cf),
                                               int(cellwise), _hx.ptr(fcell), _hx.ptr(coef), _hx.ptr(val),
                                               _hx.stream()))
        return Mat.from_csr(CSR((self.n, self.n), rowptr, col, val))

    # ---- load vectors (device tensors of length n)
    def load(self, kind, fa, fb=None, scale=1.0, cell_tag=None, tag=0):
        """cell_tag: optional int32 device tensor (n_cells); only cells with cell_tag == tag contribute."""
        fa_t = fa if isinstance(fa, torch.Tensor) else to_device(np.asarray(fa, dtype=np.complex128))
        fb_t = None if fb is None else (fb if isinstance(fb, torch.Tensor) else to_device(np.asarray(fb, np.complex128)))
        out = torch.empty(self.n, dtype=torch.complex128, device=self.x.device)
        _hx.check(_hx.lib.hx_assemble_p1_load(kind, self.dim, self.n, _hx.ptr(self.x), _hx.ptr(self.cells),
                                              _hx.ptr(self.v2c_ptr), _hx.ptr(self.v2c_idx), _hx.ptr(fa_t),
                                              _hx.ptr(fb_t), _hx.z2(scale), _hx.ptr(cell_tag), int(tag), _hx.ptr(out),
                                              _hx.stream()))
        return out

    def cell_tag_array(self, tags):
        """Dense int32 device array of cell tags (-1 = untagged) from a MeshTags, cached per MeshTags object."""
        cache = self.__dict__.setdefault("_tag_cache", {})
        if id(tags) not in cache:
            dense = np.full(self.nc, -1, dtype=np.int32)
            dense[tags.indices] = tags.values
            cache[id(tags)] = (tags, to_device(dense))
        return cache[id(tags)][1]

    def rebuild_maps(self):
        """Recompute adjacency and CSR pattern from the resident mesh arrays (what a fresh assembly pays)."""
        self.v2c_ptr, self.v2c_idx = self._adjacency(self.cells, self.nc, self.dim + 1)
        self.rowptr, self.col = self._pattern(self.cells, self.nc, self.dim + 1, self.v2c_ptr, self.v2c_idx)
        self.pattern_id = object()
        self.pattern_id = id(self.rowptr)

    def lumped(self):
        """m_j = int phi_j dx, so that int f dx = m . f for nodal P1 f."""
        if not hasattr(self, "_lumped"):
            self._lumped = self.load(0, np.ones(self.nc, dtype=np.complex128))
        return self._lumped

    def integrate_p1(self, f_array):
        """int f dx for a nodal array (dolfinx_utils.py:42-43 ``normalize``)."""
        from .petsc import panel_dot
        f = to_device(np.asarray(f_array, np.complex128))
        return complex(panel_dot(self.lumped().reshape(1, -1), 1, f)[0])   # lumped weights are real: m^H f = m.f

    def cell_volumes(self):
        """|K_c| (device kernel, cached; returned as a host array for the per-flame volume sums)."""
        if not hasattr(self, "_vol"):
            vol = torch.empty(self.nc, dtype=torch.float64, device=self.x.device)
            _hx.check(_hx.lib.hx_p1_cell_volumes(self.dim, self.nc, _hx.ptr(self.x), _hx.ptr(self.cells), _hx.ptr(vol),
                                                 _hx.stream()))
            self._vol = vol.cpu().numpy()
        return self._vol

    def locate_points(self, points, tol=1e-10):
        """Lowest-index cell containing each point (n_pts x 3) -> int32 host array (raises if a point is outside)."""
        pts = np.ascontiguousarray(np.asarray(points, dtype=np.float64).reshape(-1, 3))
        d_pts = to_device(pts)
        out = torch.full((pts.shape[0],), 2 ** 31 - 1, dtype=torch.int32, device=self.x.device)
        _hx.check(_hx.lib.hx_p1_locate_points(self.dim, self.nc, _hx.ptr(self.x), _hx.ptr(self.cells), pts.shape[0],
                                              _hx.ptr(d_pts), float(tol), _hx.ptr(out), _hx.stream()))
        cells = out.cpu().numpy()
        if (cells == 2 ** 31 - 1).any():
            raise ValueError("reference point(s) %s outside the mesh" % (pts[cells == 2 ** 31 - 1],))
        return cells
