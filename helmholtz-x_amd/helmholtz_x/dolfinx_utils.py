"""Built-in meshes with tags and small field helpers (reference: helmholtz_x/dolfinx_utils.py:8-207).

Tag conventions are the reference's: facet tag 1 left / 2 right / 3 bottom / 4 top; flame cells (all
vertices inside [x_f - a_f, x_f + a_f]) carry cell tag 0 (``tag`` in 1-D)."""
import numpy as np

from .fem import Function, Mesh, MeshTags


def cyl2cart(rho, phi, zeta):
    return rho * np.cos(phi), rho * np.sin(phi), zeta


def cart2cyl(x, y, z):
    return np.sqrt(x ** 2 + y ** 2), np.arctan2(y, x), z


def _flame_cells(mesh, x_f, a_f, tag):
    xs = mesh.geometry.x[:, 0]
    inside = (x_f - a_f - 1e-16 <= xs) & (xs <= x_f + a_f + 1e-16)
    marked = np.flatnonzero(inside[mesh.topology.cells].all(axis=1))
    return MeshTags(mesh, mesh.topology.dim, marked, np.full(marked.size, tag, dtype=np.int32))


def _facet_tags(mesh, locators):
    pts = mesh.geometry.x[mesh.facets]                     # (nf, d, 3)
    idx, val = [], []
    for marker, fn in locators:
        hit = np.flatnonzero(fn(pts).all(axis=1))
        idx.append(hit)
        val.append(np.full(hit.size, marker))
    return MeshTags(mesh, mesh.topology.dim - 1, np.hstack(idx), np.hstack(val))


def OneDimensionalSetup(n_elem, x_f=0.25, a_f=0.025, x_end=1.0, tag=0):
    x = np.zeros((n_elem + 1, 3))
    x[:, 0] = np.linspace(0.0, 1.0, n_elem + 1)
    ids = np.arange(n_elem, dtype=np.int32)
    mesh = Mesh(x, np.column_stack([ids, ids + 1]))
    subdomains = _flame_cells(mesh, x_f, a_f, tag)
    facet_tags = _facet_tags(mesh, [(1, lambda p: np.isclose(p[..., 0], 0)), (2, lambda p: np.isclose(p[..., 0], 1))])
    return mesh, subdomains, facet_tags


def RectangleSetup(nx, ny, L, h, x_f=0.25, a_f=0.025):
    gx, gy = np.linspace(0.0, L, nx + 1), np.linspace(0.0, h, ny + 1)
    x = np.zeros(((nx + 1) * (ny + 1), 3))
    x[:, 0] = np.tile(gx, ny + 1)
    x[:, 1] = np.repeat(gy, nx + 1)
    jj, ii = np.divmod(np.arange(nx * ny), nx)
    ll = jj * (nx + 1) + ii                                 # lower-left vertex of every quad
    lr, ul, ur = ll + 1, ll + nx + 1, ll + nx + 2
    cells = np.vstack([np.column_stack([ll, lr, ur]), np.column_stack([ll, ul, ur])])   # "right" diagonal
    mesh = Mesh(x, cells)
    subdomains = _flame_cells(mesh, x_f, a_f, 0)
    facet_tags = _facet_tags(mesh, [(1, lambda p: np.isclose(p[..., 0], 0)), (2, lambda p: np.isclose(p[..., 0], L)),
                                    (3, lambda p: np.isclose(p[..., 1], 0)), (4, lambda p: np.isclose(p[..., 1], h))])
    return mesh, subdomains, facet_tags


def SquareSetup(n_elem, x_f=0.25, a_f=0.025):
    return RectangleSetup(n_elem, n_elem, 1.0, 1.0, x_f, a_f)


def normalize(func):
    """Scale a P1 function so that it integrates to 1 over the domain."""
    asm = func.function_space.mesh.assembler()
    func.x.array[:] /= asm.integrate_p1(func.x.array)
    return func


def absolute(func):
    out = Function(func.function_space)
    mag = np.abs(func.x.array)
    out.x.array[:] = mag / mag.max()
    return out


def phase(func, deg=True):
    out = Function(func.function_space, "P_angle")
    out.x.array[:] = np.angle(func.x.array, deg=deg)
    return out


def unroll_dofmap(dofs, bs):
    dofs = np.asarray(dofs, dtype=np.int32)
    return (dofs[:, None] * bs + np.arange(bs, dtype=np.int32)[None, :]).ravel()


def distribute_vector_as_chunks(vector):
    """Single process per GPU here: the factored flame operator needs no redistribution."""
    return vector


def broadcast_vector(vector):
    return vector
