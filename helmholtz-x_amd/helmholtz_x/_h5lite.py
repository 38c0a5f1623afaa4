"""Minimal read-only HDF5 reader for the mesh files the reference ships (meshio/h5py output) and for the eigenvector
files its examples commit (DOLFINx XDMFFile output): superblock version 0, version-1 object headers, symbol-table
groups (nested), contiguous or chunked (B-tree v1) layouts, optional deflate filter, little-endian integer / IEEE
float datasets.  h5py is not a dependency of this package; mesh ingestion is a one-off host
step.  Only what those files use is implemented; anything else raises NotImplementedError.
"""
import struct
import zlib

import numpy as np

_UNDEF = 0xFFFFFFFFFFFFFFFF


class H5File:
    def __init__(self, path):
        with open(path, "rb") as fh:
            self.b = fh.read()
        b = self.b
        if b[:8] != b"\x89HDF\r\n\x1a\n":
            raise ValueError("%s is not an HDF5 file" % path)
        if b[8] != 0 or b[13] != 8 or b[14] != 8:
            raise NotImplementedError("only superblock version 0 with 8-byte offsets/lengths is supported")
        # root group symbol table entry starts after: 8 sig + 8 versions/sizes + 4 (k's) + 4 flags + 4*8 addresses
        ent = 8 + 8 + 4 + 4 + 32
        _, ohdr, cache, _ = struct.unpack_from("<QQII", b, ent)
        if cache != 1:
            raise NotImplementedError("root group without cached B-tree/heap addresses")
        btree, heap = struct.unpack_from("<QQ", b, ent + 24)
        self.datasets = {}                        # "name" or "group/sub/name" -> object header address
        self._walk("", btree, heap, 0)

    def _walk(self, prefix, btree, heap, depth):
        """Symbol-table group: members that carry a symbol-table message (0x0011) are groups and are descended into
        (DOLFINx writes /Mesh/<name>/geometry, /Function/real_f/0), everything else is recorded as a dataset."""
        if depth > 8:
            raise ValueError("group nesting too deep")
        b = self.b
        heap_data = self._heap_data(heap)
        for name_off, obj in self._group_entries(btree):
            end = b.index(b"\x00", heap_data + name_off)
            name = prefix + b[heap_data + name_off:end].decode()
            sub = next(((body) for mtype, body, _ in self._messages(obj) if mtype == 0x0011), None)
            if sub is not None:
                bt, hp = struct.unpack_from("<QQ", b, sub)
                self._walk(name + "/", bt, hp, depth + 1)
            else:
                self.datasets[name] = obj

    # ---- group structures
    def _heap_data(self, addr):
        if self.b[addr:addr + 4] != b"HEAP":
            raise ValueError("bad local heap")
        return struct.unpack_from("<Q", self.b, addr + 24)[0]

    def _group_entries(self, addr):
        b = self.b
        if b[addr:addr + 4] == b"SNOD":
            n = struct.unpack_from("<H", b, addr + 6)[0]
            for i in range(n):
                name_off, obj = struct.unpack_from("<QQ", b, addr + 8 + 40 * i)
                yield name_off, obj
            return
        if b[addr:addr + 4] != b"TREE" or b[addr + 4] != 0:
            raise ValueError("bad group B-tree node")
        used = struct.unpack_from("<H", b, addr + 6)[0]
        pos = addr + 24 + 8                        # skip key 0
        for _ in range(used):
            child = struct.unpack_from("<Q", b, pos)[0]
            pos += 16                              # child + next key
            yield from self._group_entries(child)

    # ---- object headers
    def _messages(self, addr):
        b = self.b
        ver, _, nmsg, _, size = struct.unpack_from("<BBHII", b, addr)
        if ver != 1:
            raise NotImplementedError("object header version %d" % ver)
        blocks = [(addr + 16, size)]
        out = []
        while blocks and len(out) < nmsg:
            pos, length = blocks.pop(0)
            end = pos + length
            while pos + 8 <= end and len(out) < nmsg:
                mtype, msize, _ = struct.unpack_from("<HHB", b, pos)
                body = pos + 8
                if mtype == 0x0010:                # continuation
                    off, ln = struct.unpack_from("<QQ", b, body)
                    blocks.append((off, ln))
                out.append((mtype, body, msize))
                pos = body + msize
        return out

    def read(self, name):
        b = self.b
        shape = dtype = layout = None
        deflate = False
        for mtype, body, msize in self._messages(self.datasets[name]):
            if mtype == 0x0001:                    # dataspace
                ver, rank, flags = struct.unpack_from("<BBB", b, body)
                off = body + (8 if ver == 1 else 4)
                shape = struct.unpack_from("<%dQ" % rank, b, off)
            elif mtype == 0x0003:                  # datatype
                cls = b[body] & 0x0F
                size = struct.unpack_from("<I", b, body + 4)[0]
                bits0 = b[body + 1]
                if bits0 & 1:
                    raise NotImplementedError("big-endian data")
                if cls == 0:
                    dtype = np.dtype("<%s%d" % ("i" if bits0 & 8 else "u", size))
                elif cls == 1:
                    dtype = np.dtype("<f%d" % size)
                else:
                    raise NotImplementedError("datatype class %d" % cls)
            elif mtype == 0x0008:                  # layout
                ver, cls = struct.unpack_from("<BB", b, body)
                if ver != 3:
                    raise NotImplementedError("layout version %d" % ver)
                if cls == 1:
                    layout = ("contiguous",) + struct.unpack_from("<QQ", b, body + 2)
                elif cls == 2:
                    nd = b[body + 2]
                    bt = struct.unpack_from("<Q", b, body + 3)[0]
                    dims = struct.unpack_from("<%dI" % nd, b, body + 11)
                    layout = ("chunked", bt, dims[:-1])
                else:
                    raise NotImplementedError("compact layout")
            elif mtype == 0x000B:                  # filter pipeline
                ver, nf = struct.unpack_from("<BB", b, body)
                pos = body + (8 if ver == 1 else 2)
                for _ in range(nf):
                    fid, nlen, _, ncd = struct.unpack_from("<HHHH", b, pos)
                    pos += 8
                    if ver == 1 or fid >= 256:
                        pos += (nlen + 7) // 8 * 8 if ver == 1 else nlen
                    pos += 4 * ncd + (4 if (ver == 1 and ncd % 2) else 0)
                    if fid == 1:
                        deflate = True
                    else:
                        raise NotImplementedError("HDF5 filter id %d" % fid)
        if shape is None or dtype is None or layout is None:
            raise ValueError("incomplete dataset header for %s" % name)
        if layout[0] == "contiguous":
            _, addr, nbytes = layout
            return np.frombuffer(b, dtype=dtype, count=int(np.prod(shape)), offset=addr).reshape(shape).copy()
        out = np.zeros(shape, dtype=dtype)
        _, btree, cdims = layout
        for offs, addr, nbytes in self._chunks(btree, len(shape)):
            raw = b[addr:addr + nbytes]
            if deflate:
                raw = zlib.decompress(raw)
            chunk = np.frombuffer(raw, dtype=dtype, count=int(np.prod(cdims))).reshape(cdims)
            sl_out = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, shape))
            sl_in = tuple(slice(0, s.stop - s.start) for s in sl_out)
            out[sl_out] = chunk[sl_in]
        return out

    def _chunks(self, addr, rank):
        b = self.b
        if b[addr:addr + 4] != b"TREE" or b[addr + 4] != 1:
            raise ValueError("bad chunk B-tree node")
        level = b[addr + 5]
        used = struct.unpack_from("<H", b, addr + 6)[0]
        key = 8 + 8 * (rank + 1)
        pos = addr + 24
        for _ in range(used):
            nbytes, _mask = struct.unpack_from("<II", b, pos)
            offs = struct.unpack_from("<%dQ" % rank, b, pos + 8)
            child = struct.unpack_from("<Q", b, pos + key)[0]
            pos += key + 8
            if level == 0:
                yield offs, child, nbytes
            else:
                yield from self._chunks(child, rank)
