"""Mesh ingestion and small result writers (reference: helmholtz_x/io_utils.py:10-217).

The reference converts gmsh ``.msh`` files to XDMF/HDF5 with meshio and reads those through DOLFINx; neither
library is a dependency here.  ``XDMFReader(name)`` therefore reads the ASCII gmsh-4.1 file ``name.msh``
directly and keeps, like io_utils.py:71-139, the top-dimensional simplices and the (dim-1)-simplices that
carry a physical tag.  ``dict_writer`` / ``dict_loader`` keep the reference's on-disk format."""
import ast
import json
from itertools import combinations

import numpy as np

from .fem import Mesh, MeshTags
from .solver_utils import info

_NODES_PER_TYPE = {15: 1, 1: 2, 2: 3, 4: 4}     # gmsh element type -> vertex count (point, line, tri, tet)


def dict_writer(filename, dictionary, extension=".txt"):
    with open(filename + extension, 'w') as file:
        file.write(json.dumps(str(dictionary)))
    print(filename + extension, " is saved.")


def dict_loader(filename, extension=".txt"):
    with open(filename + extension) as f:
        data = ast.literal_eval(json.load(f))
    print(filename + extension, " is loaded.")
    return data


def _sections(path):
    with open(path) as fh:
        lines = fh.read().split("\n")
    out, name, start = {}, None, 0
    for i, ln in enumerate(lines):
        if ln.startswith("$End"):
            out[name] = lines[start:i]
            name = None
        elif ln.startswith("$"):
            name, start = ln[1:].strip(), i + 1
    return out


def read_gmsh(path, dim=None):
    """ASCII gmsh 4.1 -> (Mesh, cell MeshTags, facet MeshTags)."""
    sec = _sections(path)
    if not sec["MeshFormat"][0].startswith("4.1 0"):
        raise NotImplementedError("only ASCII gmsh 4.1 files are supported")
    # entity -> first physical tag
    ent = sec["Entities"]
    counts = [int(v) for v in ent[0].split()]
    phys, row = {}, 1
    for edim, cnt in enumerate(counts):
        for _ in range(cnt):
            p = ent[row].split()
            row += 1
            off = 4 if edim == 0 else 7
            if int(p[off]):
                phys[(edim, int(p[0]))] = int(p[off + 1])
    # nodes
    nd = sec["Nodes"]
    nblocks, nnodes = (int(v) for v in nd[0].split()[:2])
    tags, xyz, row = [], [], 1
    for _ in range(nblocks):
        nb = int(nd[row].split()[3])
        row += 1
        tags.extend(int(t) for t in nd[row:row + nb])
        xyz.extend([float(v) for v in t.split()] for t in nd[row + nb:row + 2 * nb])
        row += 2 * nb
    tags = np.asarray(tags, dtype=np.int64)
    xyz = np.asarray(xyz, dtype=np.float64).reshape(-1, 3)
    lookup = np.full(tags.max() + 1, -1, dtype=np.int64)
    lookup[tags] = np.arange(tags.size)
    # elements of tagged entities, grouped by dimension
    el = sec["Elements"]
    nblocks = int(el[0].split()[0])
    conn = {d: [] for d in range(4)}
    ptag = {d: [] for d in range(4)}
    row = 1
    for _ in range(nblocks):
        edim, etag, etype, nb = (int(v) for v in el[row].split())
        row += 1
        k = _NODES_PER_TYPE.get(etype)
        if k is not None and (edim, etag) in phys:
            block = np.array([[int(v) for v in t.split()[1:1 + k]] for t in el[row:row + nb]], dtype=np.int64)
            conn[edim].append(lookup[block])
            ptag[edim].append(np.full(nb, phys[(edim, etag)], dtype=np.int32))
        row += nb
    if dim is None:
        dim = max(d for d in conn if conn[d])
    cells = np.concatenate(conn[dim])
    cell_values = np.concatenate(ptag[dim])
    used = np.unique(cells)
    renum = np.full(xyz.shape[0], -1, dtype=np.int64)
    renum[used] = np.arange(used.size)
    cells = renum[cells].astype(np.int32)
    facets = np.sort(renum[np.concatenate(conn[dim - 1])], axis=1).astype(np.int32)
    facet_values = np.concatenate(ptag[dim - 1])
    # adjacent cell of each tagged facet
    combos = list(combinations(range(dim + 1), dim))
    allf = np.sort(np.concatenate([cells[:, c] for c in combos], axis=0), axis=1)
    owner = np.tile(np.arange(cells.shape[0]), len(combos))
    base = int(used.size) + 1
    key_all = np.zeros(allf.shape[0], dtype=np.int64)
    key_f = np.zeros(facets.shape[0], dtype=np.int64)
    for c in range(dim):
        key_all = key_all * base + allf[:, c]
        key_f = key_f * base + facets[:, c]
    order = np.argsort(key_all, kind="stable")
    pos = np.searchsorted(key_all[order], key_f)
    facet_cells = owner[order[pos]].astype(np.int32)
    mesh = Mesh(xyz[used], cells, facets=facets, facet_cells=facet_cells)
    cell_tags = MeshTags(mesh, dim, np.arange(cells.shape[0]), cell_values)
    facet_tags = MeshTags(mesh, dim - 1, np.arange(facets.shape[0]), facet_values)
    return mesh, cell_tags, facet_tags


def _xdmf_items(path):
    """(h5 file, dataset) of the DataItems of a meshio-written .xdmf, in document order: geometry, topology, tags."""
    import os
    import re
    text = open(path).read()
    here = os.path.dirname(path)
    return [(os.path.join(here, f), d) for f, d in re.findall(r">\s*([^<>:\s]+\.h5):/([^<\s]+)\s*<", text)]


def read_xdmf(name):
    """``name.xdmf`` + ``name_tags.xdmf`` (HDF5 heavy data, as written by the reference's write_xdmf_mesh,
    io_utils.py:96-139) -> (Mesh, cell MeshTags, facet MeshTags)."""
    from ._h5lite import H5File
    files = {}

    def load(item):
        f, d = item
        if f not in files:
            files[f] = H5File(f)
        return files[f].read(d)

    geo, topo, tags = _xdmf_items(name + ".xdmf")
    x = np.asarray(load(geo), dtype=np.float64)
    if x.shape[1] == 2:
        x = np.concatenate([x, np.zeros((x.shape[0], 1))], axis=1)
    cells = np.asarray(load(topo), dtype=np.int64)
    cell_values = np.asarray(load(tags), dtype=np.int32)
    _, ftopo, ftags = _xdmf_items(name + "_tags.xdmf")
    fv = np.asarray(load(ftopo), dtype=np.int64)
    facet_values = np.asarray(load(ftags), dtype=np.int32)
    dim = cells.shape[1] - 1
    used = np.unique(cells)
    renum = np.full(x.shape[0], -1, dtype=np.int64)
    renum[used] = np.arange(used.size)
    cells = renum[cells].astype(np.int32)
    facets = np.sort(renum[fv], axis=1).astype(np.int32)
    combos = list(combinations(range(dim + 1), dim))
    allf = np.sort(np.concatenate([cells[:, c] for c in combos], axis=0), axis=1)
    owner = np.tile(np.arange(cells.shape[0]), len(combos))
    base = int(used.size) + 1
    key_all = np.zeros(allf.shape[0], dtype=np.int64)
    key_f = np.zeros(facets.shape[0], dtype=np.int64)
    for c in range(dim):
        key_all = key_all * base + allf[:, c]
        key_f = key_f * base + facets[:, c]
    order = np.argsort(key_all, kind="stable")
    facet_cells = owner[order[np.searchsorted(key_all[order], key_f)]].astype(np.int32)
    mesh = Mesh(x[used], cells, facets=facets, facet_cells=facet_cells)
    return (mesh, MeshTags(mesh, dim, np.arange(cells.shape[0]), cell_values),
            MeshTags(mesh, dim - 1, np.arange(facets.shape[0]), facet_values))


def xdmf_writer(name, mesh, function):
    """Write a P1 / DG0 function for visualisation: ``name.xdmf`` with the heavy data INLINE (Format="XML"), real and
    imaginary parts as two attributes named like the reference's output (``real_f`` / ``imag_f``, io_utils.py:141-158 ->
    dolfinx XDMFFile.write_function).  Post-processing is outside the hot path; no HDF5 is written."""
    import os
    x = mesh.geometry.x
    cells = mesh.topology.cells
    dim = mesh.topology.dim
    ttype = {1: "Polyline", 2: "Triangle", 3: "Tetrahedron"}[dim]
    vals = np.asarray(function.x.array)
    center = "Node" if vals.shape[0] == x.shape[0] else "Cell"
    assert vals.shape[0] in (x.shape[0], cells.shape[0]), "function does not live on the mesh vertices or cells"

    def block(a, fmt):
        return "\n".join(" ".join(fmt % v for v in row) for row in np.atleast_2d(a))

    parts = ['<?xml version="1.0"?>', '<Xdmf Version="3.0"><Domain><Grid Name="mesh" GridType="Uniform">',
             '<Topology TopologyType="%s" NumberOfElements="%d" NodesPerElement="%d">' % (ttype, cells.shape[0], dim + 1),
             '<DataItem Dimensions="%d %d" NumberType="Int" Format="XML">' % cells.shape, block(cells, "%d"),
             '</DataItem></Topology>', '<Geometry GeometryType="XYZ">',
             '<DataItem Dimensions="%d 3" NumberType="Float" Precision="8" Format="XML">' % x.shape[0],
             block(x, "%.17g"), '</DataItem></Geometry>']
    for label, comp in (("real_f", vals.real), ("imag_f", vals.imag)):
        parts += ['<Attribute Name="%s" AttributeType="Scalar" Center="%s">' % (label, center),
                  '<DataItem Dimensions="%d 1" NumberType="Float" Precision="8" Format="XML">' % vals.shape[0],
                  block(comp.reshape(-1, 1), "%.17g"), '</DataItem></Attribute>']
    parts.append('</Grid></Domain></Xdmf>')
    os.makedirs(os.path.dirname(os.path.abspath(name)), exist_ok=True)
    with open(name + ".xdmf", "w") as fh:
        fh.write("\n".join(parts) + "\n")
    print(name + ".xdmf is saved.")


class XDMFReader:
    """Same accessors as the reference's reader (io_utils.py:160-217).  Reads ``name.xdmf`` / ``name_tags.xdmf`` with
    their HDF5 heavy data when present (the reference's own format), else the ASCII gmsh file ``name.msh``."""

    def __init__(self, name, dim=None):
        import os
        self.name = name
        if os.path.exists(name + ".xdmf") and os.path.exists(name + "_tags.xdmf"):
            self._mesh, self._cell_tags, self._facet_tags = read_xdmf(name)
        else:
            self._mesh, self._cell_tags, self._facet_tags = read_gmsh(name + ".msh", dim)
        self._renumber_for_locality()
        info("\nXDMF Mesh - Cell data loaded.")
        info("XDMF Mesh - Facet data loaded.")

    @property
    def mesh(self):
        return self._mesh

    @property
    def subdomains(self):
        return self._cell_tags

    @property
    def facet_tags(self):
        return self._facet_tags

    @property
    def dimension(self):
        return self._mesh.topology.dim

    def _renumber_for_locality(self):
        """Meshes of ``HX_RENUMBER_MIN`` vertices or more (default 200 000; 0: never) are renumbered by reverse Cuthill-McKee
        as they are read -- what the reference gets from DOLFINx, which reorders the dofs of every mesh it reads: a file
        numbers vertices entity by entity, and the device kernels pay a factor for a numbering without locality
        (fem.bandwidth_permutation).  Cell and facet tags keep their entity order; ``mesh.input_vertex`` maps back to the
        numbering of the file.  Smaller meshes (all fixtures of the reference) keep the file numbering, which its index-based
        conventions (FixSign on entry 0, Bloch pairing by index) refer to."""
        import os
        from .fem import MeshTags, vertex_bandwidth
        limit = int(os.environ.get("HX_RENUMBER_MIN", "200000"))
        mesh = self._mesh
        if limit <= 0 or mesh.num_vertices < limit:
            return
        before = vertex_bandwidth(mesh.topology.cells)
        new = mesh.locality_renumbered("rcm")
        after = vertex_bandwidth(new.topology.cells)
        if after >= before:                       # already well ordered (a structured generator): keep the file numbering
            return
        self._mesh = new
        self._cell_tags = MeshTags(new, self._cell_tags.dim, self._cell_tags.indices, self._cell_tags.values)
        self._facet_tags = MeshTags(new, self._facet_tags.dim, self._facet_tags.indices, self._facet_tags.values)
        info("- vertices renumbered for locality (reverse Cuthill-McKee): half bandwidth %d -> %d" % (before, after))

    def getAll(self):
        return self.mesh, self.subdomains, self.facet_tags

    def getInfo(self):
        print("Number of cells:  {:,}".format(self._mesh.num_cells))
        print("Number of cores: ", 1, "\n")
