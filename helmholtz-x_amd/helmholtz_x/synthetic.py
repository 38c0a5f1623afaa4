"""Synthetic benchmark meshes: structured hex grids split into 6 Kuhn tetrahedra.

The real MICCA combustor mesh of the reference ships only as compressed HDF5, so the headline workload is a
synthetic annulus with the same radii / heights / burner layout / material data
(numerical_examples/AnnularCombustor/Micca/fullAnnulus/params.py:5-70): r in [0.14, 0.21],
z in [-0.09, 0.2], welded in theta, outlet z = 0.2 (facet tag 11), 16 cylindrical flame volumes of
radius 0.0175 on r_f = 0.175 for z in [0, 0.02] (cell tags 0..15), reference points at z_r = -0.02.

Node numbering is theta-major: idx = (i_theta*(nr+1) + i_r)*(nz+1) + i_z, so contiguous DOF ranges are
theta-slabs -- the partition used for multi-GPU row sharding.
"""
import numpy as np

from .fem import Function, FunctionSpace, Mesh, MeshTags

_KUHN = np.array([(0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0)])


def _kuhn_tets(v):
    """v: (nhex, 2, 2, 2) vertex ids indexed [h, a, b, c] along the three grid axes -> (nhex*6, 4) tets.

    Tet p of a hex walks 000 -> +e_p0 -> +e_p1 -> 111; all hexes share the 000-111 diagonal so neighbouring
    hexes agree on their common face diagonals (conforming mesh)."""
    nh = v.shape[0]
    out = np.empty((nh, 6, 4), dtype=np.int32)
    for p, perm in enumerate(_KUHN):
        cur = np.zeros(3, dtype=int)
        out[:, p, 0] = v[:, 0, 0, 0]
        for s in range(2):
            cur[perm[s]] = 1
            out[:, p, s + 1] = v[:, cur[0], cur[1], cur[2]]
        out[:, p, 3] = v[:, 1, 1, 1]
    return out.reshape(-1, 4)


def annulus(nr, ntheta, nz, r_in=0.14, r_out=0.21, z0=-0.09, z1=0.2, n_flames=16, r_f=0.175, flame_radius=0.0175,
            flame_z=(0.0, 0.02), sector=None):
    """Returns (mesh, subdomains, facet_tags).  ``sector=(theta0, theta1)`` builds an un-welded sector with
    facet tags 12 (theta0 side, 'Master') and 13 (theta1 side, 'Slave') instead of the full ring."""
    full = sector is None
    nth_nodes = ntheta if full else ntheta + 1
    th = (np.arange(nth_nodes) * (2 * np.pi / ntheta) if full
          else np.linspace(sector[0], sector[1], ntheta + 1))
    r = np.linspace(r_in, r_out, nr + 1)
    z = np.linspace(z0, z1, nz + 1)
    T, R, Z = np.meshgrid(th, r, z, indexing="ij")
    x = np.stack([R * np.cos(T), R * np.sin(T), Z], axis=-1).reshape(-1, 3)

    def nid(it, ir, iz):
        return ((it % nth_nodes) * (nr + 1) + ir) * (nz + 1) + iz

    it, ir, iz = np.meshgrid(np.arange(ntheta), np.arange(nr), np.arange(nz), indexing="ij")
    it, ir, iz = it.ravel(), ir.ravel(), iz.ravel()
    v = np.empty((it.size, 2, 2, 2), dtype=np.int64)
    for a in range(2):
        for b in range(2):
            for c in range(2):
                v[:, a, b, c] = nid(it + a, ir + b, iz + c)
    cells = _kuhn_tets(v)

    # outlet facets (top, c = 1): triangles {001,101,111} in tet perm (2,0,1) and {001,011,111} in perm (2,1,0)
    top = np.flatnonzero(iz == nz - 1)
    p201 = int(np.flatnonzero((_KUHN == (2, 0, 1)).all(axis=1))[0])
    p210 = int(np.flatnonzero((_KUHN == (2, 1, 0)).all(axis=1))[0])
    f1 = np.stack([v[top, 0, 0, 1], v[top, 1, 0, 1], v[top, 1, 1, 1]], axis=1)
    f2 = np.stack([v[top, 0, 0, 1], v[top, 0, 1, 1], v[top, 1, 1, 1]], axis=1)
    facets = [f1, f2]
    fcells = [top * 6 + p201, top * 6 + p210]
    ftags = [np.full(top.size, 11), np.full(top.size, 11)]
    if not full:
        # theta faces: a = 0 of the first slab (Master, 12) / a = 1 of the last slab (Slave, 13)
        for side, tag, sel in ((0, 12, it == 0), (1, 13, it == ntheta - 1)):
            h = np.flatnonzero(sel)
            pa = int(np.flatnonzero((_KUHN == (1, 2, 0)).all(axis=1))[0]) if side == 0 else \
                int(np.flatnonzero((_KUHN == (0, 1, 2)).all(axis=1))[0])
            pb = int(np.flatnonzero((_KUHN == (2, 1, 0)).all(axis=1))[0]) if side == 0 else \
                int(np.flatnonzero((_KUHN == (0, 2, 1)).all(axis=1))[0])
            a = side
            facets += [np.stack([v[h, a, 0, 0], v[h, a, 1, 0], v[h, a, 1, 1]], axis=1),
                       np.stack([v[h, a, 0, 0], v[h, a, 0, 1], v[h, a, 1, 1]], axis=1)]
            fcells += [h * 6 + pa, h * 6 + pb]
            ftags += [np.full(h.size, tag), np.full(h.size, tag)]
    facets = np.sort(np.concatenate(facets), axis=1)
    mesh = Mesh(x, cells, facets=facets, facet_cells=np.concatenate(fcells))
    facet_tags = MeshTags(mesh, 2, np.arange(facets.shape[0]), np.concatenate(ftags))

    # flame volumes: cell centroid inside the burner cylinder
    cent = x[cells].mean(axis=1)
    tags = np.full(cells.shape[0], -1, dtype=np.int32)
    band = (cent[:, 2] >= flame_z[0]) & (cent[:, 2] <= flame_z[1])
    idx = np.flatnonzero(band)
    for f in range(n_flames):
        ang = f * 2 * np.pi / 16
        cx, cy = r_f * np.cos(ang), r_f * np.sin(ang)
        hit = idx[(cent[idx, 0] - cx) ** 2 + (cent[idx, 1] - cy) ** 2 <= flame_radius ** 2]
        tags[hit] = f
    marked = np.flatnonzero(tags >= 0)
    subdomains = MeshTags(mesh, 3, marked, tags[marked])
    mesh._centroids = cent
    return mesh, subdomains, facet_tags


def annulus_part(nr, ntheta, nz, ranges, rank, group=None, r_in=0.14, r_out=0.21, z0=-0.09, z1=0.2, n_flames=16, r_f=0.175,
                 flame_radius=0.0175, flame_z=(0.0, 0.02)):
    """This rank's part of the welded annulus WITHOUT the global mesh: exactly what
    ``distributed.distribute_mesh(annulus(nr, ntheta, nz)[0], ranges, rank, subdomains, facet_tags)`` returns -- the cells touching
    an owned vertex (ascending global cell id), vertices [owned | ghost] in ascending global id, the outlet facets with an owned
    vertex, flame tags, ``mesh.part`` with the same maps -- generated from the grid formulas for the rank's theta-planes and the
    one plane either side of them.  ``ranges``: the ranks' row ranges, cut between theta-planes (``distributed.partition_rows``
    with align = (nr + 1)(nz + 1)).  The reference's counterpart: XDMFFile.read_mesh on N ranks hands every rank its cells and
    one ghost layer (io_utils.py:149-155); the synthetic workload is analytic, so a rank needs no reader and no N-fold host copy.
    Returns (local_mesh, local_subdomains, local_facet_tags)."""
    from .distributed import DistComm
    plane = (nr + 1) * (nz + 1)
    r0, r1 = ranges[rank]
    n_global = ntheta * plane
    if r0 % plane or r1 % plane or not (0 <= r0 < r1 <= n_global):
        raise ValueError("the row range of a rank must be cut between theta-planes (multiples of %d)" % plane)
    p0, p1 = r0 // plane, r1 // plane
    hex_planes = np.arange(p0 - 1, p1) % ntheta if p1 - p0 < ntheta else np.arange(ntheta)      # hexes with a vertex in [p0, p1)
    hex_planes = np.unique(hex_planes)
    it, ir, iz = np.meshgrid(hex_planes, np.arange(nr), np.arange(nz), indexing="ij")
    it, ir, iz = it.ravel(), ir.ravel(), iz.ravel()
    hex_id = (it * nr + ir) * nz + iz                          # global hex index, ascending (hex_planes ascending)

    def nid(jt, jr, jz):
        return ((jt % ntheta) * (nr + 1) + jr) * (nz + 1) + jz

    v = np.empty((it.size, 2, 2, 2), dtype=np.int64)
    for a in range(2):
        for b in range(2):
            for c in range(2):
                v[:, a, b, c] = nid(it + a, ir + b, iz + c)
    gcells = _kuhn_tets(v).astype(np.int64)                    # every tet of a hex holds its 000 and 111 corners: both planes
    keep = (hex_id[:, None] * 6 + np.arange(6)[None, :]).ravel()
    verts = np.unique(gcells)
    owned = np.arange(r0, r1, dtype=np.int64)
    ghost = verts[(verts < r0) | (verts >= r1)]
    l2g = np.concatenate([owned, ghost])
    order = np.argsort(l2g, kind="stable")
    lcells = order[np.searchsorted(l2g[order], gcells.ravel())].reshape(gcells.shape)
    # coordinates of the local vertices from the grid formulas (the same arithmetic as annulus(): bitwise the same numbers)
    th = np.arange(ntheta) * (2 * np.pi / ntheta)
    rr = np.linspace(r_in, r_out, nr + 1)
    zz = np.linspace(z0, z1, nz + 1)
    jt, rem = np.divmod(l2g, plane)
    jr, jz = np.divmod(rem, nz + 1)
    x = np.stack([rr[jr] * np.cos(th[jt]), rr[jr] * np.sin(th[jt]), zz[jz]], axis=-1)
    # outlet facets of the kept hexes (both triangles of a top face hold vertices of both planes): global ids as in annulus()
    n_top = ntheta * nr                                        # top hexes of the whole ring, in hex order
    top = np.flatnonzero(iz == nz - 1)
    top_rank = (it[top] * nr + ir[top])                        # position of the hex among the ring's top hexes
    f1 = np.stack([v[top, 0, 0, 1], v[top, 1, 0, 1], v[top, 1, 1, 1]], axis=1)
    f2 = np.stack([v[top, 0, 0, 1], v[top, 0, 1, 1], v[top, 1, 1, 1]], axis=1)
    p201 = int(np.flatnonzero((_KUHN == (2, 0, 1)).all(axis=1))[0])
    p210 = int(np.flatnonzero((_KUHN == (2, 1, 0)).all(axis=1))[0])
    gf = np.sort(np.concatenate([f1, f2]), axis=1)
    gf_id = np.concatenate([top_rank, n_top + top_rank])
    gf_cell = np.concatenate([hex_id[top] * 6 + p201, hex_id[top] * 6 + p210])
    fo = np.argsort(gf_id, kind="stable")
    gf, gf_id, gf_cell = gf[fo], gf_id[fo], gf_cell[fo]
    g2l = np.full(n_global, -1, dtype=np.int64) if n_global <= 4 * l2g.size else None
    if g2l is not None:
        g2l[l2g] = np.arange(l2g.size)
        lf = np.sort(g2l[gf], axis=1)
    else:
        lf = np.sort(order[np.searchsorted(l2g[order], gf.ravel())].reshape(gf.shape), axis=1)
    fcell = np.searchsorted(keep, gf_cell)
    local = Mesh(x, lcells.astype(np.int32), facets=lf.astype(np.int32), facet_cells=fcell.astype(np.int32))
    low = gcells.min(axis=1)
    flow = gf.min(axis=1) if gf.size else np.zeros(0, np.int64)
    local.part = dict(r0=r0, r1=r1, n_own=int(owned.size), n_global=n_global, local_to_global=l2g, cell_ids=keep,
                      cell_owned=(low >= r0) & (low < r1), facet_ids=gf_id, ranges=list(ranges), rank=rank,
                      n_cells_global=ntheta * nr * nz * 6, facet_owned=(flow >= r0) & (flow < r1))
    local.comm = DistComm(group)
    facet_tags = MeshTags(local, 2, np.arange(lf.shape[0]), np.full(lf.shape[0], 11))
    cent = x[lcells].mean(axis=1)
    tags = np.full(lcells.shape[0], -1, dtype=np.int32)
    band = (cent[:, 2] >= flame_z[0]) & (cent[:, 2] <= flame_z[1])
    idx = np.flatnonzero(band)
    for f in range(n_flames):
        ang = f * 2 * np.pi / 16
        cx, cy = r_f * np.cos(ang), r_f * np.sin(ang)
        hit = idx[(cent[idx, 0] - cx) ** 2 + (cent[idx, 1] - cy) ** 2 <= flame_radius ** 2]
        tags[hit] = f
    marked = np.flatnonzero(tags >= 0)
    subdomains = MeshTags(local, 3, marked, tags[marked])
    local._centroids = cent
    return local, subdomains, facet_tags


def micca_sound_speed(mesh, gamma=1.4, r=287.0, T_amb=300.0, T_a=1521.0, T_b=1200.0, l_cc=0.2):
    """DG0 axial sound-speed field of Micca/fullAnnulus/params.py:53-70 (cell-midpoint rule)."""
    c = Function(FunctionSpace(mesh, "DG", 0), "soundspeed")
    zc = getattr(mesh, "_centroids", None)
    zc = (mesh.geometry.x[mesh.topology.cells].mean(axis=1) if zc is None else zc)[:, 2]
    mid = np.sqrt(gamma * r * ((T_b - T_a) * (zc / l_cc) ** 2 + T_a))
    c.x.array[:] = np.where(zc < 0, np.sqrt(gamma * r * T_amb),
                            np.where((zc > 0) & (zc < l_cc), mid, np.sqrt(gamma * r * T_b)))
    return c


def box(nx, ny, nz, lengths=(1.0, 1.0, 1.0)):
    """[0,Lx]x[0,Ly]x[0,Lz] hex grid -> 6 tets per hex; facet tags 1..6 = x0,x1,y0,y1,z0,z1 (exterior facets found
    combinatorially, so only use for moderate sizes)."""
    gx, gy, gz = (np.linspace(0, L, n + 1) for L, n in zip(lengths, (nx, ny, nz)))
    X, Y, Z = np.meshgrid(gx, gy, gz, indexing="ij")
    x = np.stack([X, Y, Z], axis=-1).reshape(-1, 3)
    i, j, k = (a.ravel() for a in np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij"))
    v = np.empty((i.size, 2, 2, 2), dtype=np.int64)
    for a in range(2):
        for b in range(2):
            for c in range(2):
                v[:, a, b, c] = ((i + a) * (ny + 1) + (j + b)) * (nz + 1) + (k + c)
    mesh = Mesh(x, _kuhn_tets(v))
    pts = mesh.geometry.x[mesh.facets]
    idx, val = [], []
    for tag, (ax, pos) in enumerate(((0, 0.0), (0, lengths[0]), (1, 0.0), (1, lengths[1]), (2, 0.0), (2, lengths[2])), 1):
        hit = np.flatnonzero(np.isclose(pts[:, :, ax], pos).all(axis=1))
        idx.append(hit)
        val.append(np.full(hit.size, tag))
    facet_tags = MeshTags(mesh, 2, np.hstack(idx), np.hstack(val))
    return mesh, MeshTags(mesh, 3, np.zeros(0, np.int32), np.zeros(0, np.int32)), facet_tags


def refine_tets(mesh, cell_tags=None, facet_tags=None):
    """Uniform (red) refinement of a tetrahedral mesh: every edge gets a midpoint vertex, every cell eight children (the four
    corner tetrahedra + four from the inner octahedron, cut along its SHORTEST diagonal: Bey's rule keeps the shape regular),
    every tagged facet four children with its tag.  Conforming by construction (midpoints are shared through the edge table).
    The way to carry a real, unstructured geometry -- the reference's MICCA mesh has 34 787 vertices -- to the sizes the
    device path is built for (x 7.3 vertices per level), with none of the generator's regularity.  Old vertices keep their
    ids, midpoints follow in the order of the sorted edge list; children of cell c are cells 8c .. 8c+7, children of tagged
    facet f are 4f .. 4f+3.  Returns (mesh, cell MeshTags or None, facet MeshTags or None).  Host integer work."""
    from .fem import Mesh, MeshTags
    cells = np.asarray(mesh.topology.cells, dtype=np.int64)
    if cells.shape[1] != 4:
        raise NotImplementedError("refine_tets: tetrahedra only")
    x = mesh.geometry.x
    nv = x.shape[0]
    pairs = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]
    lo = np.stack([np.minimum(cells[:, a], cells[:, b]) for a, b in pairs], axis=1)
    hi = np.stack([np.maximum(cells[:, a], cells[:, b]) for a, b in pairs], axis=1)
    keys = lo * nv + hi
    uniq, inv = np.unique(keys.ravel(), return_inverse=True)
    mid = nv + inv.reshape(keys.shape)                       # midpoint vertex of each cell edge
    x_new = np.concatenate([x, 0.5 * (x[uniq // nv] + x[uniq % nv])], axis=0)
    del lo, hi, keys, inv
    v0, v1, v2, v3 = (cells[:, i] for i in range(4))
    m01, m02, m03, m12, m13, m23 = (mid[:, i] for i in range(6))
    corner = [np.stack(t, axis=1) for t in ((v0, m01, m02, m03), (m01, v1, m12, m13), (m02, m12, v2, m23),
                                            (m03, m13, m23, v3))]
    # inner octahedron: three possible diagonals (m01-m23, m02-m13, m03-m12); four tetrahedra around the chosen one
    def dist2(a, b):
        d = x_new[a] - x_new[b]
        return np.einsum("ij,ij->i", d, d)
    d = np.stack([dist2(m01, m23), dist2(m02, m13), dist2(m03, m12)], axis=1)
    pick = np.argmin(d, axis=1)
    # ring of the four other vertices around each diagonal, in cyclic order
    rings = [(m02, m03, m13, m12), (m01, m03, m23, m12), (m01, m02, m23, m13)]
    ends = [(m01, m23), (m02, m13), (m03, m12)]
    inner = []
    for j in range(4):
        cols = []
        for k in range(4):               # vertex k of the j-th inner child, per diagonal choice
            opts = []
            for dgn in range(3):
                a, b = ends[dgn]
                ring = rings[dgn]
                opts.append((a, b, ring[j], ring[(j + 1) % 4])[k])
            cols.append(np.choose(pick, opts))
        inner.append(np.stack(cols, axis=1))
    children = np.stack(corner + inner, axis=1).reshape(-1, 4)        # cell c -> rows 8c .. 8c+7
    # children take the orientation of their parent (swap the last two vertices where the signed volume disagrees)
    def signed(cc):
        X = x_new[cc]
        return np.linalg.det(X[:, 1:] - X[:, :1])
    flip = np.sign(signed(children)) != np.repeat(np.sign(signed(cells)), 8)
    children[flip] = children[flip][:, [0, 1, 3, 2]]
    # tagged facets: four children each (three corner triangles + the middle one)
    f = np.asarray(mesh.facets, dtype=np.int64)
    def midpoint(a, b):
        k = np.minimum(a, b) * nv + np.maximum(a, b)
        pos = np.searchsorted(uniq, k)
        if not np.array_equal(uniq[pos], k):
            raise ValueError("a tagged facet has an edge no cell has")
        return nv + pos
    a, b, c = f[:, 0], f[:, 1], f[:, 2]
    mab, mac, mbc = midpoint(a, b), midpoint(a, c), midpoint(b, c)
    fch = np.stack([np.stack(t, axis=1) for t in ((a, mab, mac), (mab, b, mbc), (mac, mbc, c), (mab, mac, mbc))],
                   axis=1).reshape(-1, 3)
    fch = np.sort(fch, axis=1)
    # adjacent cell of each child facet: a child of the parent's adjacent cell (search its eight children's faces)
    parent_cell = np.repeat(np.asarray(mesh.facet_cells, dtype=np.int64), 4)
    cand = children.reshape(-1, 8, 4)[parent_cell]                     # (nf4, 8, 4)
    fc = np.full(fch.shape[0], -1, dtype=np.int64)
    for j in range(8):
        tet = cand[:, j, :]
        for skip in range(4):
            face = np.sort(np.delete(tet, skip, axis=1), axis=1)
            hit = np.all(face == fch, axis=1) & (fc < 0)
            fc[hit] = 8 * parent_cell[hit] + j
    if np.any(fc < 0):
        raise ValueError("child facet without an adjacent child cell")
    new = Mesh(x_new, children.astype(np.int32), facets=fch.astype(np.int32), facet_cells=fc.astype(np.int32))
    ct = ft = None
    if cell_tags is not None:
        ct = MeshTags(new, cell_tags.dim, (8 * cell_tags.indices.astype(np.int64)[:, None] + np.arange(8)).ravel(),
                      np.repeat(cell_tags.values, 8))
    if facet_tags is not None:
        ft = MeshTags(new, facet_tags.dim, (4 * facet_tags.indices.astype(np.int64)[:, None] + np.arange(4)).ravel(),
                      np.repeat(facet_tags.values, 4))
    return new, ct, ft
