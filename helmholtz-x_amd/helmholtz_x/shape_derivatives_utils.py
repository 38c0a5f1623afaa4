"""Free-form-deformation lattice and normalisation helpers of the shape-derivative workflow
(reference: helmholtz_x/shape_derivatives_utils.py:8-187).

The reference reads node coordinates from a live ``gmsh.model`` (``mesh.getNodes``); gmsh is a mesh GENERATOR and out of
scope here, but nothing the lattice needs is CAD data: ``FFDCylindrical.__init__`` (:110-158) takes minima / maxima of the
nodes' cylindrical coordinates and ``calcSTU`` (:171-189) is arithmetic on them.  So the lattice is built from the mesh's own
node coordinates -- a ``Mesh`` (or an (n, 3) coordinate array) stands where the reference passes ``gmsh.model``.  The mesh
deformation helpers (``deformCylindricalFFD`` etc., :232-300) write a new gmsh model and stay out of scope."""
import numpy as np

from .dolfinx_utils import cart2cyl, cyl2cart


def _node_coordinates(source):
    x = source.geometry.x if hasattr(source, "geometry") else np.asarray(source, dtype=np.float64)
    return np.ascontiguousarray(x, dtype=np.float64).reshape(-1, 3)


def _split_parts(shape_derivatives):
    re = {kv: {ku: np.real(v) for ku, v in row.items()} for kv, row in shape_derivatives.items()}
    im = {kv: {ku: np.imag(v) for ku, v in row.items()} for kv, row in shape_derivatives.items()}
    return re, im


def derivatives_normalize(shape_derivatives):
    """{key: complex} -> real parts divided by the largest |real part|, imaginary parts by the largest |imaginary part|
    (:8-36)."""
    max_re = max(abs(np.real(v)) for v in shape_derivatives.values())
    max_im = max(abs(np.imag(v)) for v in shape_derivatives.values())
    return {k: np.real(v) / max_re + 1j * np.imag(v) / max_im for k, v in shape_derivatives.items()}


def nonaxisymmetric_derivatives_normalize(shape_derivatives):
    """{zeta: {phi: complex}} -> the same normalisation over the whole table (:38-90).  As in the reference the maxima are
    searched over the COLUMNS that hold some row's largest entry (every row's arg-max column, then all rows of those
    columns) -- which contains the global maximum, so this is max |Re| and max |Im| over all entries."""
    re, im = _split_parts(shape_derivatives)
    cols_re = {max(row, key=lambda k: abs(row[k])) for row in re.values()}
    cols_im = {max(row, key=lambda k: abs(row[k])) for row in im.values()}
    max_re = max(abs(re[kv][ku]) for kv in re for ku in cols_re)
    max_im = max(abs(im[kv][ku]) for kv in im for ku in cols_im)
    return {kv: {ku: re[kv][ku] / max_re + 1j * im[kv][ku] / max_im for ku in row} for kv, row in shape_derivatives.items()}


class FFDCylindrical:
    """Cylindrical l x m x n lattice (radius, angle, height) around the nodes of a mesh (:92-189).

    ``source``: a Mesh or an (n, 3) array of node coordinates (the reference: ``gmsh.model`` + entity selection -- pass the
    coordinates of the nodes the selection would return; the RijkeFFD run selects all of them, main_shape.py:100).  The
    lattice spans [min rho, max rho] x [min phi, min phi + 2 pi] x [min z, max z] of those nodes, phi = atan2(y, x)."""

    def __init__(self, source, l, m, n, dim=3, tag=-1, includeBoundary=False, parametric=True):
        self.l, self.m, self.n = int(l), int(m), int(n)
        x = _node_coordinates(source)
        rhos, phis, zetas = cart2cyl(x[:, 0], x[:, 1], x[:, 2])
        self.dr = rhos.max() - rhos.min()
        self.dphi = 2 * np.pi
        self.dz = zetas.max() - zetas.min()
        i, j, k = np.meshgrid(np.arange(l), np.arange(m), np.arange(n), indexing="ij")
        self.Pr = rhos.min() + self.dr * i / (l - 1)
        self.Pphi = phis.min() + self.dphi * j / (m - 1)
        self.Pz = zetas.min() + self.dz * k / (n - 1)
        self.Px, self.Py, _ = cyl2cart(self.Pr, self.Pphi, self.Pz)
        self.P0 = np.array([self.Pr[0, 0, 0], self.Pphi[0, 0, 0], self.Pz[0, 0, 0]])
        self.dr_base, self.dphi_base, self.dz_base = self.dr, self.dphi, self.dz

    def calcSTU(self, coords):
        """Parametric lattice coordinates of points given as a flat [x0, y0, z0, x1, ...] array (gmsh's layout) or (n, 3)."""
        c = np.asarray(coords, dtype=np.float64).reshape(-1, 3)
        rhos, phis, zetas = cart2cyl(c[:, 0], c[:, 1], c[:, 2])
        return (rhos - self.P0[0]) / self.dr, (phis - self.P0[1]) / self.dphi, (zetas - self.P0[2]) / self.dz

    def control_points(self):
        """Cartesian coordinates of the l*m*n control points (what ``write_ffd_points`` would write to a .vtu file)."""
        return np.stack(cyl2cart(self.Pr.ravel(), self.Pphi.ravel(), self.Pz.ravel()), axis=1)
