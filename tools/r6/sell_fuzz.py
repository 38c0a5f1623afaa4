#!/usr/bin/env python3
"""Random stress of the SELL-64 format with overflow tails through the C-ABI (hx_sell_*): row-length distributions (uniform, normal,
a few very long rows, bimodal, empty rows), banded and far columns (16-bit deltas / wide slices), rectangular operators (ghost
columns), value types (complex128, complex64, complex-half); operator product, residual, smoother step and the block product
against SciPy.  usage: sell_fuzz.py [cases [seed0]]"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "helmholtz-x_amd")]
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402
import torch  # noqa: E402
from helmholtz_x import _hx as hx  # noqa: E402

dev = hx.require_gpu()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0


def rand_c(rng, *shape):
    return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)


def to_dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


bad = []
stats = dict(split=0, tails=0)
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    n = int(rng.choice([1, 5, 63, 64, 65, 127, 1000, 4097, int(rng.integers(2, 30000))]))
    dist = rng.choice(["uniform", "normal", "long", "bimodal", "equal", "sparse"])
    if dist == "uniform":
        lens = rng.integers(0, 40, n)
    elif dist == "normal":
        lens = np.clip(np.rint(rng.normal(14.5, rng.uniform(0.5, 4.0), n)), 1, 60).astype(np.int64)
    elif dist == "long":
        lens = rng.integers(8, 20, n)
        lens[rng.integers(0, n, max(1, n // 200))] = rng.integers(70, 400)           # rows far beyond any body (and the 63-entry tail limit)
    elif dist == "bimodal":
        lens = np.where(rng.random(n) < 0.5, 9, 27)
    elif dist == "equal":
        lens = np.full(n, int(rng.integers(1, 33)))
    else:
        lens = (rng.random(n) < 0.3) * rng.integers(1, 12, n)
    n_cols = n + (int(rng.integers(1, max(2, n // 4))) if rng.random() < 0.3 else 0)       # rectangular: [owned | ghost]
    lens = np.minimum(lens, n_cols)
    spread = int(rng.choice([30, 400, 40000, n_cols]))
    rows = np.repeat(np.arange(n), lens)
    cols = (rows + rng.integers(-spread, spread + 1, rows.size)) % n_cols
    M = sp.csr_matrix((rand_c(rng, rows.size), (rows, cols)), shape=(n, n_cols))     # (duplicates are summed: lengths shrink a little)
    M.sort_indices()
    want = [(1, 0), (1, 1), (1, 2), (0, 2), (0, 1)][int(rng.integers(0, 5))]
    tail_opt = None if rng.random() < 0.8 else rng.choice(["0", "1"])
    step = None if rng.random() < 0.7 else str(int(rng.choice([16, 64, 512])))
    gran = None if rng.random() < 0.7 else str(int(rng.choice([1, 2, 4])))
    for k, v in (("sell_tail", tail_opt), ("sell_tail_step", step), ("sell_tail_gran", gran)):
        hx.check(hx.lib.hx_set_option(k.encode(), None if v is None else str(v).encode()))
    rp, ci, va = to_dev(M.indptr.astype(np.int32)), to_dev(M.indices.astype(np.int32)), to_dev(M.data.astype(np.complex128))
    h = C.c_void_p()
    if M.nnz == 0:
        continue
    hx.check(hx.lib.hx_sell_create(C.byref(h), n, hx.ptr(rp), hx.ptr(ci), hx.ptr(va), want[0], want[1], hx.stream()))
    for k in ("sell_tail", "sell_tail_step", "sell_tail_gran"):
        hx.check(hx.lib.hx_set_option(k.encode(), None))
    ns, pad, tail, split = C.c_int32(), C.c_int64(), C.c_int64(), C.c_int32()
    hx.check(hx.lib.hx_sell_info(h, C.byref(ns), C.byref(pad)))
    hx.check(hx.lib.hx_sell_tail_info(h, C.byref(tail), C.byref(split)))
    stats["split"] += int(split.value > 0)
    stats["tails"] += tail.value
    errs = {}
    x, y0 = rand_c(rng, n_cols), rand_c(rng, n)
    scale = abs(M) @ np.abs(x) + np.abs(y0) + 1e-300
    if pad.value + tail.value < M.nnz:
        errs["stored entries"] = 1.0
    if want[0]:
        xd, yd = to_dev(x), to_dev(y0.copy())
        a, b = 0.7 - 0.2j, -1.3 + 0.4j
        hx.check(hx.lib.hx_sell_spmv_z(h, hx.ptr(xd), hx.ptr(yd), hx.z2(a), hx.z2(b), hx.stream()))
        errs["product"] = float(np.max(np.abs(yd.cpu().numpy() - (a * (M @ x) + b * y0)) / scale)) / 1e-13
        if n_cols == n:
            nb = hx.lib.hx_block_width()
            X = rand_c(rng, n, nb)
            Xd, Yd = to_dev(X.copy()), to_dev(np.zeros((n, nb), dtype=np.complex128))
            hx.check(hx.lib.hx_sell_spmm_z(h, hx.ptr(Xd), hx.ptr(Yd), hx.z2(1.0), hx.z2(0.0), hx.stream()))
            sc = (abs(M) @ np.abs(X)) + 1e-300
            errs["block product"] = float(np.max(np.abs(Yd.cpu().numpy() - M @ X) / sc)) / 1e-13
    if want[1] and n_cols == n:
        tol = 3e-5 if want[1] == 1 else 4e-3
        d = M.diagonal().copy()
        d[d == 0] = 1.0
        x32, y32 = to_dev(x.astype(np.complex64)), to_dev(y0.astype(np.complex64))
        hx.check(hx.lib.hx_sell_spmv_c64(h, hx.ptr(x32), hx.ptr(y32), -1.0, 1.0, hx.stream()))
        errs["residual"] = float(np.max(np.abs(y32.cpu().numpy() - (y0 - M @ x)) / scale)) / tol
        d32, b32, o32 = to_dev((1.0 / d).astype(np.complex64)), to_dev(y0.astype(np.complex64)), torch.empty_like(x32)
        hx.check(hx.lib.hx_sell_smooth_c64(h, hx.ptr(d32), hx.ptr(b32), hx.ptr(x32), hx.ptr(o32), 0.7, None, 0.25, hx.stream()))
        ref = x + 0.7 * (1.0 / d) * (y0 - M @ x) + 0.25 * x
        errs["smoother"] = float(np.max(np.abs(o32.cpu().numpy() - ref) / (np.abs(x) * 1.25 + 0.7 * np.abs(1.0 / d) * scale))) / tol
    hx.check(hx.lib.hx_sell_destroy(h))
    ok = all(v <= 1.0 for v in errs.values())
    if not ok or case % 25 == 0:
        print("%s case %3d n %5d x %5d %-7s spread %5d want %s tail/step/gran %s/%s/%s: stored %d + tail %d (nnz %d) in %d of %d slices; errors / tolerance %s"
              % ("ok " if ok else "BAD", seed0 + case, n, n_cols, dist, spread, want, tail_opt, step, gran, pad.value, tail.value, M.nnz, split.value, ns.value,
                 {k: round(v, 3) for k, v in errs.items()}), flush=True)
    if not ok:
        bad.append(seed0 + case)
print("cases %d, operators with tails %d (%d tail entries in all), failures %s" % (n_cases, stats["split"], stats["tails"], bad))
sys.exit(1 if bad else 0)
