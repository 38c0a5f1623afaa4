"""Oracle: minimal mesh inputs (interval, right-diagonal rectangle, ASCII gmsh 4.1 reader).

TEST INFRASTRUCTURE ONLY.  Restates the built-in setups of helmholtz_x/dolfinx_utils.py:73-185
(OneDimensionalSetup / RectangleSetup: tag conventions, flame-cell marking) and the
gmsh -> (cells + "gmsh:physical" tags) conversion of helmholtz_x/io_utils.py:71-139 so the oracle
can be pinned to the reference's logs without touching product code.
"""
from collections import namedtuple
from itertools import combinations

import numpy as np

OMesh = namedtuple("OMesh", "x cells cell_tags facets facet_tags facet_cells")


def boundary_facets(cells):
    """All exterior facets of a simplicial mesh: (facets (nf,d) sorted ids, owning cell (nf,))."""
    cells = np.asarray(cells)
    k = cells.shape[1]
    d = k - 1
    if d == 1:
        fl = cells.reshape(-1, 1)
        owner = np.repeat(np.arange(cells.shape[0]), 2)
    else:
        combos = list(combinations(range(k), d))
        fl = np.sort(np.concatenate([cells[:, c] for c in combos], axis=0), axis=1)
        owner = np.tile(np.arange(cells.shape[0]), len(combos))
    uniq, first, counts = np.unique(fl, axis=0, return_index=True, return_counts=True)
    mask = counts == 1
    return uniq[mask].astype(np.int32), owner[first[mask]].astype(np.int32)


def _tag_facets(x, facets, locators):
    tags = np.full(facets.shape[0], -1, dtype=np.int32)
    for marker, fn in locators:
        on = np.all(fn(x[facets]), axis=1)
        tags[on] = marker
    return tags


def interval(n_elem, x_f=0.25, a_f=0.025, tag=0):
    """dolfinx_utils.py:73-106: unit interval, facet tag 1 at x=0, 2 at x=1, flame cells tag ``tag``."""
    x = np.zeros((n_elem + 1, 3))
    x[:, 0] = np.linspace(0.0, 1.0, n_elem + 1)
    cells = np.stack([np.arange(n_elem), np.arange(1, n_elem + 1)], axis=1).astype(np.int32)
    eps = 1e-16
    inside = np.logical_and(x_f - a_f - eps <= x[:, 0], x[:, 0] <= x_f + a_f + eps)
    cell_tags = np.where(np.all(inside[cells], axis=1), tag, -1).astype(np.int32)
    facets, owner = boundary_facets(cells)
    ftags = _tag_facets(x, facets, [(1, lambda p: np.isclose(p[..., 0], 0)),
                                    (2, lambda p: np.isclose(p[..., 0], 1))])
    return OMesh(x, cells, cell_tags, facets, ftags, owner)


def rectangle(nx, ny, L, h, x_f=0.25, a_f=0.025):
    """dolfinx_utils.py:149-185: [0,L]x[0,h], right-diagonal triangles; tags 1 left 2 right 3 bottom 4 top."""
    xs = np.linspace(0.0, L, nx + 1)
    ys = np.linspace(0.0, h, ny + 1)
    X, Y = np.meshgrid(xs, ys, indexing='xy')            # row-major: iy slow, ix fast
    x = np.zeros(((nx + 1) * (ny + 1), 3))
    x[:, 0] = X.ravel()
    x[:, 1] = Y.ravel()
    ix, iy = np.meshgrid(np.arange(nx), np.arange(ny), indexing='xy')
    v0 = (iy * (nx + 1) + ix).ravel()
    v1 = v0 + 1
    v2 = v0 + (nx + 1)
    v3 = v2 + 1
    cells = np.concatenate([np.stack([v0, v1, v3], axis=1), np.stack([v0, v2, v3], axis=1)]).astype(np.int32)
    eps = 1e-16
    inside = np.logical_and(x_f - a_f - eps <= x[:, 0], x[:, 0] <= x_f + a_f + eps)
    cell_tags = np.where(np.all(inside[cells], axis=1), 0, -1).astype(np.int32)
    facets, owner = boundary_facets(cells)
    ftags = _tag_facets(x, facets, [(1, lambda p: np.isclose(p[..., 0], 0)),
                                    (2, lambda p: np.isclose(p[..., 0], L)),
                                    (3, lambda p: np.isclose(p[..., 1], 0)),
                                    (4, lambda p: np.isclose(p[..., 1], h))])
    return OMesh(x, cells, cell_tags, facets, ftags, owner)


_GMSH_NODES = {1: 2, 2: 3, 4: 4, 15: 1}


def read_gmsh41(path, dim):
    """ASCII gmsh 4.1 -> OMesh keeping, like io_utils.py:71-139, the top-dimensional simplices and
    the (dim-1)-simplices with their first physical tag."""
    with open(path) as fh:
        tok = fh.read().split('\n')
    pos = {name: i for i, name in enumerate(tok) if name.startswith('$')}

    # --- $Entities: entity (dim, tag) -> first physical tag
    phys = {}
    i = pos['$Entities'] + 1
    counts = [int(v) for v in tok[i].split()]
    i += 1
    for edim, cnt in enumerate(counts):
        for _ in range(cnt):
            p = tok[i].split()
            i += 1
            off = 4 if edim == 0 else 7
            nphys = int(p[off])
            if nphys:
                phys[(edim, int(p[0]))] = int(p[off + 1])

    # --- $Nodes
    i = pos['$Nodes'] + 1
    nblocks, nnodes = [int(v) for v in tok[i].split()[:2]]
    i += 1
    tags = np.empty(nnodes, dtype=np.int64)
    xyz = np.empty((nnodes, 3))
    k = 0
    for _ in range(nblocks):
        nb = int(tok[i].split()[3])
        i += 1
        if nb == 0:
            continue
        tags[k:k + nb] = [int(t) for t in tok[i:i + nb]]
        i += nb
        xyz[k:k + nb] = [[float(v) for v in t.split()] for t in tok[i:i + nb]]
        i += nb
        k += nb
    lookup = np.full(tags.max() + 1, -1, dtype=np.int64)
    lookup[tags] = np.arange(nnodes)

    # --- $Elements
    i = pos['$Elements'] + 1
    nblocks = int(tok[i].split()[0])
    i += 1
    by_dim = {d: ([], []) for d in range(4)}
    for _ in range(nblocks):
        edim, etag, etype, nb = [int(v) for v in tok[i].split()]
        i += 1
        nn = _GMSH_NODES.get(etype)
        if nn is not None and (edim, etag) in phys:
            conn = np.array([[int(v) for v in t.split()[1:1 + nn]] for t in tok[i:i + nb]], dtype=np.int64)
            by_dim[edim][0].append(lookup[conn])
            by_dim[edim][1].append(np.full(nb, phys[(edim, etag)], dtype=np.int32))
        i += nb
    cells = np.concatenate(by_dim[dim][0])
    cell_tags = np.concatenate(by_dim[dim][1])
    facets = np.concatenate(by_dim[dim - 1][0])
    facet_tags = np.concatenate(by_dim[dim - 1][1])

    # compress to the vertices used by the top-dimensional cells
    used = np.unique(cells)
    renum = np.full(nnodes, -1, dtype=np.int64)
    renum[used] = np.arange(used.size)
    x = xyz[used]
    cells = renum[cells].astype(np.int32)
    facets = np.sort(renum[facets], axis=1).astype(np.int32)

    # owning cell of every tagged facet
    bf, owner = _all_facets_with_owner(cells)
    key = {tuple(f): o for f, o in zip(bf.tolist(), owner.tolist())}
    facet_cells = np.array([key[tuple(f)] for f in facets.tolist()], dtype=np.int32)
    return OMesh(x, cells, cell_tags, facets, facet_tags, facet_cells)


def _all_facets_with_owner(cells):
    k = cells.shape[1]
    combos = list(combinations(range(k), k - 1))
    fl = np.sort(np.concatenate([cells[:, c] for c in combos], axis=0), axis=1)
    owner = np.tile(np.arange(cells.shape[0]), len(combos))
    uniq, first = np.unique(fl, axis=0, return_index=True)
    return uniq, owner[first]
