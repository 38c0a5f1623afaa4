"""petsc4py-shaped ``Vec`` / ``Mat`` wrappers over device memory and libhx kernels.

Only the surface the reference's hot path touches is provided (call sites:
eigensolvers.py:45,173-176,211,240,309-315; petsc4py_utils.py:4-111; bloch_operator.py:42,68,106-110;
eigenvectors.py:20-46).  A ``Mat`` is a *lazy* linear combination

    M = sum_i coef_i * CSR_i  +  sum_j coef_j * (U_j V_j^T)

so ``A - D`` or ``A + omega*B - D`` (eigensolvers.py:174-176,240) never materialises the rank-r flame
block ``a (x) b`` that the reference stores explicitly (flame_matrices.py:91-92,176,233).
PyTorch is used for device allocation / copies only; all arithmetic is libhx (HIP).
"""
import ctypes as C

import numpy as np
import scipy.sparse as sp
import torch

from . import _hx


def _dev():
    return _hx.require_gpu()


def to_device(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(_dev())


class Vec:
    """Device complex128 vector with the petsc4py Vec methods the reference calls."""

    shard = None          # sharded.ShardContext when the vector lives on a distributed mesh ([owned | ghost] entries)

    def __init__(self, data):
        if isinstance(data, torch.Tensor):
            self.t = data
        else:
            self.t = to_device(np.asarray(data, dtype=np.complex128))

    @classmethod
    def zeros(cls, n):
        return cls(torch.zeros(int(n), dtype=torch.complex128, device=_dev()))

    # --- petsc4py surface
    def copy(self):
        v = Vec(self.t.clone())
        v.shard = self.shard
        return v

    def duplicate(self):
        v = Vec.zeros(self.t.numel())
        v.shard = self.shard
        return v

    def getSize(self):
        return self.t.numel()

    def scale(self, z):
        _hx.check(_hx.lib.hx_zscal(self.t.numel(), _hx.z2(z), _hx.ptr(self.t), _hx.stream()))

    def conjugate(self):
        self.t = torch.conj_physical(self.t)

    def axpy(self, alpha, x):
        _hx.check(_hx.lib.hx_zaxpy(self.t.numel(), _hx.z2(alpha), _hx.ptr(x.t), _hx.ptr(self.t), _hx.stream()))

    def _reduced(self, value):
        t = torch.tensor([complex(value)], dtype=torch.complex128, device=self.t.device)
        self.shard.all_reduce_(t)
        return complex(t[0].item())

    def dot(self, other):
        """PETSc VecDot(x, y) = y^H x  (petsc4py_utils.py:57-61 relies on this).  Row-sharded vectors: owned entries, summed
        over the ranks (what VecDot does on a distributed Vec)."""
        if self.shard is not None:
            n = self.shard.n_own
            return self._reduced(panel_dot(other.t[:n].reshape(1, -1), 1, self.t[:n])[0])
        return panel_dot(other.t.reshape(1, -1), 1, self.t)[0]

    def norm(self):
        if self.shard is not None:
            return float(np.sqrt(self._reduced(panel_dot(None, 0, self.t[:self.shard.n_own])[-1].real).real))
        return float(np.sqrt(panel_dot(None, 0, self.t)[-1].real))

    def sum(self):
        return complex(self.t.sum().item())

    @property
    def array(self):
        return self.t.cpu().numpy()

    def getArray(self):
        return self.array

    def setArray(self, a):
        self.t.copy_(torch.from_numpy(np.asarray(a, dtype=np.complex128)))

    def getComm(self):
        return _SerialComm()

    def __getitem__(self, i):
        return complex(self.t[i].item())

    def assemble(self):
        pass

    def destroy(self):
        self.t = None


class Options:
    """``PETSc.Options``-shaped view of ONE process-wide options database (petsc4py: ``PETSc.Options()``; the reference's solver
    set-up ends in ``E.setFromOptions()`` / ``Q.setFromOptions()``, eigensolvers.py:60,109, so that ``-eps_nev 6 -eps_tol 1e-10
    -st_pc_type lu ...`` on the command line of a driver, or ``PETSC_OPTIONS`` in the environment, override what the code set).
    The database is seeded once from ``PETSC_OPTIONS`` and ``sys.argv`` (what ``petsc4py.init(sys.argv)`` does); names are stored
    without the leading dash.  ``slepc.EPS/PEP.setFromOptions`` maps the ``eps_* / pep_* / st_*`` names it knows onto the setters
    of the device solvers and ``hx_<name>`` onto the library's own registry (hx_set_option)."""

    _db = None

    def __init__(self, prefix=None):
        self.prefix = (prefix or "").lstrip("-")
        if Options._db is None:
            Options._db = {}
            import os
            import shlex
            import sys
            Options._parse(shlex.split(os.environ.get("PETSC_OPTIONS", "")))
            Options._parse(sys.argv[1:])

    @staticmethod
    def _is_name(tok):
        """An option name starts with a dash followed by a letter (``-1e-3`` or ``-0.5`` are values)."""
        return len(tok) > 1 and tok[0] == "-" and (tok[1].isalpha() or (tok[1] == "-" and len(tok) > 2 and tok[2].isalpha()))

    @classmethod
    def _parse(cls, tokens):
        i = 0
        while i < len(tokens):
            tok = tokens[i]
            if cls._is_name(tok):
                val = ""
                if i + 1 < len(tokens) and not cls._is_name(tokens[i + 1]):
                    val = tokens[i + 1]
                    i += 1
                cls._db[tok.lstrip("-")] = val
            i += 1

    def _key(self, name):
        return self.prefix + str(name).lstrip("-")

    # --- petsc4py surface
    def setValue(self, name, value):
        Options._db[self._key(name)] = "" if value is None else (("1" if value else "0") if isinstance(value, bool) else str(value))

    __setitem__ = setValue

    def delValue(self, name):
        Options._db.pop(self._key(name), None)

    __delitem__ = delValue

    def hasName(self, name):
        return self._key(name) in Options._db

    __contains__ = hasName

    def getString(self, name, default=None):
        return Options._db.get(self._key(name), default)

    def __getitem__(self, name):
        return Options._db[self._key(name)]

    def getInt(self, name, default=None):
        v = self.getString(name)
        return default if v in (None, "") else int(float(v))

    def getReal(self, name, default=None):
        v = self.getString(name)
        return default if v in (None, "") else float(v)

    def getScalar(self, name, default=None):
        v = self.getString(name)
        return default if v in (None, "") else complex(v.replace("i", "j").replace(" ", ""))

    def getBool(self, name, default=None):
        v = self.getString(name)
        if v is None:
            return default
        return v.lower() in ("", "1", "true", "yes", "on")

    def getAll(self):
        n = len(self.prefix)
        return {k[n:]: v for k, v in Options._db.items() if k.startswith(self.prefix)}

    def clear(self):
        for k in [k for k in Options._db if k.startswith(self.prefix)]:
            del Options._db[k]
        return self


class _SerialComm:
    def getRank(self):
        return 0

    def getSize(self):
        return 1


_ws_cache = {}
_PATTERN_IDS = __import__("itertools").count(1)


def new_pattern_id():
    """Identity of a sparsity pattern: a process-wide counter, never reused (an ``id()`` would be once its object is freed, so
    caches keyed by it had to keep whole operators -- values included -- alive)."""
    return next(_PATTERN_IDS)


_PATTERN_COORDS = __import__("collections").OrderedDict()     # pattern id -> host (n, dim) coordinates of the unknowns


def register_coords(pattern_id, coords):
    """Where the unknowns of a sparsity pattern sit in space (a reference to the caller's host array, not a copy): the direct
    solver's nested dissection cuts along these (slepc.LUSolver); patterns without coordinates are dissected on their graph."""
    _PATTERN_COORDS[pattern_id] = coords
    while len(_PATTERN_COORDS) > 16:
        _PATTERN_COORDS.popitem(last=False)


def coords_of(pattern_id):
    return _PATTERN_COORDS.get(pattern_id)


def panel_dot(V, k, w):
    """[V_j^H w for j<k] + [w^H w] as a host numpy array (one D2H sync).  V: (m, n) tensor, rows = vectors."""
    n = w.numel()
    key = (w.device.index, k)
    if key not in _ws_cache:
        _ws_cache[key] = (torch.empty(_hx.lib.hx_panel_dot_ws(k), dtype=torch.complex128, device=w.device),
                          torch.empty(k + 1, dtype=torch.complex128, device=w.device))
    ws, out = _ws_cache[key]
    ldv = V.stride(0) if (V is not None and k > 0) else n
    _hx.check(_hx.lib.hx_zpanel_dot(n, k, _hx.ptr(V) if k > 0 else None, ldv, _hx.ptr(w), _hx.ptr(out),
                                    _hx.ptr(ws), _hx.stream()))
    return out.cpu().numpy()


class CSR:
    """Device CSR (int32 rowptr/col, complex128 val) + a host copy of the integer pattern."""

    def __init__(self, shape, rowptr, col, val, h_rowptr=None, h_col=None, pattern_id=None):
        self.shape = (int(shape[0]), int(shape[1]))
        self.rowptr, self.col, self.val = rowptr, col, val
        self._h_rowptr, self._h_col = h_rowptr, h_col
        self.pattern_id = pattern_id if pattern_id is not None else new_pattern_id()
        self.value_id = new_pattern_id()                # identity of this set of VALUES (caches of objects built from them)
        self.nnz = int(col.numel())

    @classmethod
    def from_scipy(cls, M):
        M = sp.csr_matrix(M, dtype=np.complex128)
        M.sort_indices()
        rp, ci = M.indptr.astype(np.int32), M.indices.astype(np.int32)
        return cls(M.shape, to_device(rp), to_device(ci), to_device(M.data), rp, ci)

    def host_pattern(self):
        if self._h_rowptr is None:
            self._h_rowptr = self.rowptr.cpu().numpy()
            self._h_col = self.col.cpu().numpy()
        return self._h_rowptr, self._h_col

    def to_scipy(self):
        rp, ci = self.host_pattern()
        return sp.csr_matrix((self.val.cpu().numpy(), ci.copy(), rp.copy()), shape=self.shape)

    def with_values(self, val):
        out = CSR(self.shape, self.rowptr, self.col, val, self._h_rowptr, self._h_col, self.pattern_id)
        out.symmetric = getattr(self, "symmetric", False)
        out.local = getattr(self, "local", False)      # this rank's rows over [owned | ghost] columns (row-sharded run)
        return out

    def spmv(self, x, y, alpha=1.0, beta=0.0, lanes=0):
        if self.nnz == 0:                        # an operator without entries: y = beta y
            if beta == 0.0:
                y.zero_()
            elif beta != 1.0:
                _hx.check(_hx.lib.hx_zscal(y.numel(), _hx.z2(beta), _hx.ptr(y), _hx.stream()))
            return
        _hx.check(_hx.lib.hx_csr_spmv_z(self.shape[0], self.nnz, _hx.ptr(self.rowptr), _hx.ptr(self.col),
                                        _hx.ptr(self.val), _hx.ptr(x), _hx.ptr(y), _hx.z2(alpha), _hx.z2(beta),
                                        lanes, _hx.stream()))


class LowRank:
    """sum_f u_f v_f^T with sparse u_f, v_f given as (int32 idx, complex128 val) host pairs."""

    def __init__(self, n, pairs, local=False):
        """``local``: indices are this rank's OWNED local rows (row-sharded run; n = owned rows)."""
        self.local = bool(local)
        self.n = int(n)
        self.pairs = [((np.asarray(ui, np.int32), np.asarray(uv, np.complex128)),
                       (np.asarray(vi, np.int32), np.asarray(vv, np.complex128))) for (ui, uv), (vi, vv) in pairs]

    @property
    def rank(self):
        return len(self.pairs)

    def transposed(self):
        return LowRank(self.n, [(v, u) for u, v in self.pairs], self.local)

    def conjugated(self):
        return LowRank(self.n, [((ui, uv.conj()), (vi, vv.conj())) for (ui, uv), (vi, vv) in self.pairs], self.local)

    def to_scipy(self):
        M = sp.csr_matrix((self.n, self.n), dtype=np.complex128)
        for (ui, uv), (vi, vv) in self.pairs:
            rows = np.repeat(ui, len(vi))
            cols = np.tile(vi, len(ui))
            M = M + sp.coo_matrix((np.outer(uv, vv).ravel(), (rows, cols)), shape=(self.n, self.n)).tocsr()
        return M

    # device views (built on demand)
    def right_packed(self):
        if not hasattr(self, "_rp"):
            ptr = np.zeros(self.rank + 1, np.int32)
            ptr[1:] = np.cumsum([len(v[0]) for _, v in self.pairs])
            idx = np.concatenate([v[0] for _, v in self.pairs]) if self.rank else np.zeros(0, np.int32)
            val = np.concatenate([v[1] for _, v in self.pairs]) if self.rank else np.zeros(0, np.complex128)
            self._rp = (to_device(ptr), to_device(idx), to_device(val))
        return self._rp

    def left_packed(self):
        """[(device int32 indices, device complex128 values)] of the left factors u_f: their supports are a few hundred entries."""
        if not hasattr(self, "_lp"):
            self._lp = [(to_device(np.asarray(ui, dtype=np.int32)), to_device(np.asarray(uv, dtype=np.complex128)))
                        for (ui, uv), _ in self.pairs]
        return self._lp

    def left_packed_all(self):
        """(ptr, idx, val, disjoint) of ALL left factors, packed like ``right_packed``: one scatter launch per product
        (hx_zsparse_axpys).  ``disjoint``: no row index occurs in two factors."""
        if not hasattr(self, "_lpa"):
            lens = [len(u[0]) for u, _ in self.pairs]
            ptr = np.zeros(self.rank + 1, np.int32)
            ptr[1:] = np.cumsum(lens)
            idx = np.concatenate([np.asarray(u[0], dtype=np.int32) for u, _ in self.pairs]) if self.rank else np.zeros(0, np.int32)
            val = np.concatenate([np.asarray(u[1], dtype=np.complex128) for u, _ in self.pairs]) if self.rank else np.zeros(0, np.complex128)
            disjoint = int(np.unique(idx).size == idx.size)
            if idx.size == 0:
                idx, val = np.zeros(1, np.int32), np.zeros(1, np.complex128)
            self._lpa = (to_device(ptr), to_device(idx), to_device(val), disjoint)
        return self._lpa

    def left_dense(self):
        """(rank, n) device tensor whose rows are the dense u_f: right-hand sides of the Woodbury columns.  Built on every call
        and owned by the caller (16 flames x 5 M rows are 1.2 GiB: until round 4 the tensor was cached here for the lifetime
        of the flame matrix)."""
        U = torch.zeros((self.rank, self.n), dtype=torch.complex128, device=_dev())
        for f, (ui, uv) in enumerate(self.left_packed()):
            if ui.numel():
                U[f, ui.long()] = uv
        return U


class Mat:
    """Lazy linear combination of CSR terms and low-rank terms (see module docstring)."""

    __array_ufunc__ = None        # numpy scalars defer to Mat.__rmul__ (omega * B, eigensolvers.py:176)

    def __init__(self, shape=(0, 0), csr_terms=(), lr_terms=()):
        self.shape = (int(shape[0]), int(shape[1]))
        self.csr_terms = [(complex(c), m) for c, m in csr_terms]
        self.lr_terms = [(complex(c), m) for c, m in lr_terms]
        self._mat = None
        self._coo = None             # (rows, cols, vals) blocks collected by setValues until assemblyEnd

    # ---- incremental assembly in the petsc4py style (flame_matrices.py:165-189, 219-244 build the explicit D this way:
    #      create -> setSizes -> setType('aij') -> setUp -> setValues(rows, cols, block, ADD_VALUES) -> assemblyBegin/End)
    def create(self, comm=None):
        return self

    def setSizes(self, size, bsize=None):
        def glob(s):
            return int(s[1] if isinstance(s, (tuple, list)) else s)
        size = size if isinstance(size, (tuple, list)) else (size, size)
        self.shape = (glob(size[0]), glob(size[1]))

    def setType(self, mat_type):
        pass

    def setUp(self):
        pass

    def setPreallocationNNZ(self, nnz):
        pass

    def setOption(self, option, flag):
        pass

    def setValues(self, rows, cols, values, addv=None):
        """Dense block ``values`` (len(rows) x len(cols), row-major) added into the matrix (ADD_VALUES semantics)."""
        rows = np.atleast_1d(np.asarray(rows, dtype=np.int64))
        cols = np.atleast_1d(np.asarray(cols, dtype=np.int64))
        vals = np.asarray(values, dtype=np.complex128).reshape(len(rows), len(cols))
        if self._coo is None:
            self._coo = ([], [], [])
        self._coo[0].append(np.repeat(rows, len(cols)))
        self._coo[1].append(np.tile(cols, len(rows)))
        self._coo[2].append(vals.ravel())

    # ---- construction helpers (petsc4py API)
    @classmethod
    def from_csr(cls, csr):
        return cls(csr.shape, [(1.0, csr)])

    def createAIJ(self, size, csr=None, comm=None):
        if np.isscalar(size):
            size = (size, size)
        ai, aj, av = csr
        M = sp.csr_matrix((np.asarray(av, np.complex128), np.asarray(aj), np.asarray(ai)), shape=tuple(size))
        m = Mat.from_csr(CSR.from_scipy(M))
        self.__dict__.update(m.__dict__)
        return self

    def getSize(self):
        return self.shape

    def createVecs(self):
        return Vec.zeros(self.shape[1]), Vec.zeros(self.shape[0])

    getVecs = createVecs

    def assemble(self):
        self.assemblyEnd()
        return self

    def assemblyBegin(self):
        pass

    def assemblyEnd(self):
        if self._coo is not None:                  # duplicates are summed (ADD_VALUES), then the CSR goes to the device
            r, c, v = (np.concatenate(a) if a else np.zeros(0) for a in self._coo)
            M = sp.coo_matrix((v, (r, c)), shape=self.shape).tocsr()
            M.sum_duplicates()
            self.csr_terms.append((1.0 + 0j, CSR.from_scipy(M)))
            self._coo = None

    def destroy(self):
        self.csr_terms, self.lr_terms, self._mat, self._coo = [], [], None, None

    def copy(self):
        return Mat(self.shape, list(self.csr_terms), list(self.lr_terms))

    # ---- algebra (new lazy objects; nothing is computed here)
    def _scaled(self, z):
        z = complex(z)
        return Mat(self.shape, [(c * z, m) for c, m in self.csr_terms], [(c * z, m) for c, m in self.lr_terms])

    def __mul__(self, z):
        if isinstance(z, Vec):
            y = Vec.zeros(self.shape[0])
            self.mult(z, y)
            return y
        return self._scaled(z)

    __rmul__ = __mul__

    def __neg__(self):
        return self._scaled(-1.0)

    def __add__(self, other):
        if other is None:
            return self.copy()
        return Mat(self.shape, self.csr_terms + other.csr_terms, self.lr_terms + other.lr_terms)

    __radd__ = __add__

    def __sub__(self, other):
        return self + (-other)

    def scale(self, z):
        m = self._scaled(z)
        self.csr_terms, self.lr_terms, self._mat = m.csr_terms, m.lr_terms, None

    def transpose(self):
        """In place, like petsc4py (acoustic_matrices.py:113).  The CSR transpose is integer/structural work done
        once on the host; values are re-uploaded.  Terms flagged symmetric (the P1 operators A, B, C) are kept."""
        self.csr_terms = [(c, m if getattr(m, "symmetric", False) else CSR.from_scipy(m.to_scipy().T.tocsr()))
                          for c, m in self.csr_terms]
        self.lr_terms = [(c, m.transposed()) for c, m in self.lr_terms]
        self._mat = None
        return self

    def conjugate(self):
        self.csr_terms = [(np.conj(c), m.with_values(torch.conj_physical(m.val))) for c, m in self.csr_terms]
        self.lr_terms = [(np.conj(c), m.conjugated()) for c, m in self.lr_terms]
        self._mat = None
        return self

    # ---- evaluation
    def csr_part(self):
        """Materialise sum_i coef_i CSR_i as ONE device CSR (cached)."""
        if self._mat is None:
            self._mat = combine_csr(self.shape, self.csr_terms)
        return self._mat

    def lowrank_part(self):
        """[(coef, LowRank)] as given."""
        return self.lr_terms

    def mult(self, x, y):
        K = self.csr_part()
        shard = getattr(x, "shard", None)
        if shard is not None:            # row-sharded operand: its ghost entries first (MatMult's scatter)
            shard.halo(x.t)
        if K is None:
            y.t.zero_()
        else:
            K.spmv(x.t, y.t)
        for coef, lr in self.lr_terms:
            if lr.rank == 0:
                continue
            ptr, idx, val = lr.right_packed()
            t = torch.empty(lr.rank, dtype=torch.complex128, device=x.t.device)
            _hx.check(_hx.lib.hx_zsparse_dots(lr.rank, _hx.ptr(ptr), _hx.ptr(idx), _hx.ptr(val), _hx.ptr(x.t),
                                              _hx.ptr(t), _hx.stream()))
            if shard is not None and lr.local:   # v_f^T x: the ranks hold their owned entries of v_f
                shard.all_reduce_(t)
            # y += coef * sum_f (v_f^T x) u_f: ONE scatter launch over the supports of the u_f, coefficients read on the device
            # (until round 5: coefficients to the host -- a stream synchronisation per product -- and one launch per flame)
            uptr, uidx, uval, disjoint = lr.left_packed_all()
            _hx.check(_hx.lib.hx_zsparse_axpys(lr.rank, _hx.ptr(uptr), _hx.ptr(uidx), _hx.ptr(uval), _hx.ptr(t), _hx.z2(coef),
                                               _hx.ptr(y.t), disjoint, _hx.stream()))

    def getValuesCSR(self):
        """Explicit CSR including the outer-product blocks -- what the reference stores (small cases / tests only)."""
        M = sp.csr_matrix(self.shape, dtype=np.complex128)
        K = self.csr_part()
        if K is not None:
            M = M + K.to_scipy()
        for coef, lr in self.lr_terms:
            M = M + coef * lr.to_scipy()
        M = sp.csr_matrix(M)
        M.sort_indices()
        return M.indptr, M.indices, M.data

    def to_scipy(self):
        ai, aj, av = self.getValuesCSR()
        return sp.csr_matrix((av, aj, ai), shape=self.shape)


from collections import OrderedDict

_SLOT_CACHE = OrderedDict()     # (base pattern id, other pattern id) -> device slot map; LRU bounded in entries and bytes
_SLOT_CACHE_MAX_BYTES = 64 << 20


def clear_caches():
    _SLOT_CACHE.clear()
    _ws_cache.clear()


def locate_slots(ncols, brp, bci, rp, ci):
    """Slots of the entries (rp, ci) inside the pattern (brp, bci), or None if some entry is missing.  Only the
    base rows that the smaller pattern touches are visited (B lives on boundary rows: ~2 % of the rows)."""
    cnt = np.diff(rp)
    touched = np.flatnonzero(cnt > 0)
    if touched.size == 0:
        return np.zeros(0, np.int32)
    seg_len = (brp[touched + 1] - brp[touched]).astype(np.int64)
    first = np.cumsum(seg_len) - seg_len
    idx = np.repeat(brp[touched].astype(np.int64) - first, seg_len) + np.arange(int(seg_len.sum()), dtype=np.int64)
    keys_b = np.repeat(touched.astype(np.int64), seg_len) * ncols + bci[idx]
    keys_m = np.repeat(touched.astype(np.int64), cnt[touched]) * ncols + ci
    pos = np.searchsorted(keys_b, keys_m)
    pos_c = np.minimum(pos, len(keys_b) - 1)
    if len(keys_b) == 0 or not np.array_equal(keys_b[pos_c], keys_m):
        return None
    return idx[pos_c].astype(np.int32)


def locate_slots_device(base, m):
    """Device form of locate_slots: slots (int32 device tensor) of the entries of CSR ``m`` inside CSR ``base``, or None if
    some entry of ``m`` is missing from ``base`` (hx_csr_locate_slots: one binary search per entry, nothing leaves HBM)."""
    slots = torch.empty(max(m.nnz, 1), dtype=torch.int32, device=_dev())[:m.nnz]
    if m.nnz == 0:
        return slots
    missing = torch.zeros(1, dtype=torch.int32, device=_dev())
    _hx.check(_hx.lib.hx_csr_locate_slots(m.shape[0], _hx.ptr(m.rowptr), _hx.ptr(m.col), _hx.ptr(base.rowptr),
                                          _hx.ptr(base.col), _hx.ptr(slots), _hx.ptr(missing), _hx.stream()))
    return None if int(missing.item()) else slots


def combine_csr(shape, terms):
    """sum_i coef_i CSR_i on the device.  Same-pattern terms are one fused lincomb kernel; terms on a different
    pattern are scatter-added through a host-computed slot map (integer work only)."""
    terms = [(c, m) for c, m in terms if m is not None]
    if not terms:
        return None
    if len(terms) > 1:              # operators without entries (a rank's part of a boundary operator) contribute nothing
        kept = [(c, m) for c, m in terms if m.nnz > 0]
        terms = kept if kept else terms[:1]
    if len(terms) == 1 and terms[0][0] == 1.0:
        return terms[0][1]
    # base pattern = the one with most nonzeros; if some term's pattern is not contained in it, use the union
    base = max(terms, key=lambda cm: cm[1].nnz)[1]
    others = [m for _, m in terms if m.pattern_id != base.pattern_id]
    slot_maps = {}
    if others and all((base.pattern_id, m.pattern_id) in _SLOT_CACHE for m in others):
        slot_maps = {m.pattern_id: _SLOT_CACHE[(base.pattern_id, m.pattern_id)] for m in others}
        others = []
    if others:
        n = shape[0]
        found = {m.pattern_id: locate_slots_device(base, m) for m in others}       # patterns stay in HBM
        need_union = any(v is None for v in found.values())
        if need_union:
            brp, bci = base.host_pattern()
        if need_union:
            U = sp.csr_matrix((np.ones(len(bci), np.int8), bci, brp), shape=shape)
            for m in others:
                rp, ci = m.host_pattern()
                U = U + sp.csr_matrix((np.ones(len(ci), np.int8), ci, rp), shape=shape)
            U = sp.csr_matrix(U)
            U.sort_indices()
            urp, uci = U.indptr.astype(np.int32), U.indices.astype(np.int32)
            base = CSR(shape, to_device(urp), to_device(uci), None, urp, uci)
            others = [m for _, m in terms]
            brp, bci = urp, uci
            found = {m.pattern_id: locate_slots(shape[1], brp, bci, *m.host_pattern()) for m in others}
        for m in others:
            if m.pattern_id in slot_maps:
                continue
            f = found[m.pattern_id]
            slot_maps[m.pattern_id] = f if isinstance(f, torch.Tensor) else to_device(f)
            if not need_union and 4 * m.nnz <= _SLOT_CACHE_MAX_BYTES // 4:
                # (pattern ids are never reused: the cache holds the small maps only, no operator -- it used to keep up to
                # eight (base, other) operator pairs with their values alive across fixed-point calls, 1.4 GiB each at 5 M DOF)
                _SLOT_CACHE[(base.pattern_id, m.pattern_id)] = slot_maps[m.pattern_id]
                while len(_SLOT_CACHE) > 8 or sum(4 * v.numel() for v in _SLOT_CACHE.values()) > _SLOT_CACHE_MAX_BYTES:
                    _SLOT_CACHE.popitem(last=False)
    same = [(c, m) for c, m in terms if m.pattern_id == base.pattern_id and m.val is not None]
    out = None
    i = 0
    while i < len(same):
        take = 4 if out is None else 3
        chunk = same[i:i + take]
        i += take
        vals = [m.val for _, m in chunk]
        coefs = [c for c, _ in chunk]
        if out is None:
            out = torch.empty(base.nnz, dtype=torch.complex128, device=_dev())
        else:
            vals, coefs = [out] + vals, [1.0] + coefs
        arr = (C.c_void_p * len(vals))(*[v.data_ptr() for v in vals])
        cf = (C.c_double * (2 * len(coefs)))(*[p for c in coefs for p in (complex(c).real, complex(c).imag)])
        _hx.check(_hx.lib.hx_zlincomb(base.nnz, len(vals), cf, arr, _hx.ptr(out), _hx.stream()))
    if out is None:
        out = torch.zeros(base.nnz, dtype=torch.complex128, device=_dev())
    for c, m in terms:
        if m.pattern_id != base.pattern_id:
            _hx.check(_hx.lib.hx_zscatter_axpy(m.nnz, _hx.z2(c), _hx.ptr(m.val), _hx.ptr(slot_maps[m.pattern_id]),
                                               _hx.ptr(out), _hx.stream()))
    return base.with_values(out)


class PETSc:
    """Namespace with the names a driver written against ``from petsc4py import PETSc`` touches on the hot path
    (``PETSc.ScalarType(+1000)`` in active_modes.py:42, ``PETSc.Mat().create``, ``PETSc.InsertMode.ADD_VALUES``, ``PETSc.Options()``)."""
    Mat = Mat
    Vec = Vec
    Options = Options
    ScalarType = np.complex128
    IntType = np.int32
    DECIDE = -1

    class InsertMode:
        INSERT_VALUES = INSERT = "insert"
        ADD_VALUES = ADD = "add"
