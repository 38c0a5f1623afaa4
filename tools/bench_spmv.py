#!/usr/bin/env python3
"""Kernel micro-benchmark: assembly + CSR SpMV on a synthetic annulus, every lanes-per-row variant.
Usage: python tools/bench_spmv.py annulus-1M [annulus-5M ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "helmholtz-x_amd"), os.path.join(ROOT, "tests")]
os.environ.setdefault("HX_QUIET", "1")
import torch  # noqa: E402

import bench  # noqa: E402
from helmholtz_x.acoustic_matrices import AcousticMatrices  # noqa: E402
from helmholtz_x.petsc import combine_csr  # noqa: E402

for name in sys.argv[1:]:
    t0 = time.perf_counter()
    case = bench.build_case(bench.WORKLOADS[name])
    t1 = time.perf_counter()
    mesh = case["mesh"]
    asm = mesh.assembler()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    P = case["P"]
    M = AcousticMatrices(mesh, case["facet_tags"], {11: {'Robin': P.R_outlet}}, case["c"])
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    asm.rebuild_maps()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    K = combine_csr(M.A.shape, M.A.csr_terms + [(P.target * c, m) for c, m in M.B.csr_terms]
                    + [(P.target ** 2 * c, m) for c, m in M.C.csr_terms])
    n, nnz = K.shape[0], K.nnz
    x = torch.randn(n, dtype=torch.complex128, device=K.val.device)
    y = torch.empty_like(x)
    res = {"workload": name, "n": n, "nnz": nnz, "mesh_gen_s": round(t1 - t0, 2), "upload+maps_s": round(t2 - t1, 3),
           "assemble_ABC_s": round(t3 - t2, 3), "rebuild_maps_s": round(t4 - t3, 3)}
    alg = nnz * 20 + n * 36
    for lanes in (0, 1, 2, 4, 8, 16, 32):
        for _ in range(3):
            K.spmv(x, y, lanes=lanes)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            K.spmv(x, y, lanes=lanes)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 30
        res["lanes%d" % lanes] = {"ms": round(ms, 4), "GBs": round(alg / ms / 1e6, 1)}
    print(json.dumps(res), flush=True)
    del case, M, K, x, y, asm, mesh
    torch.cuda.empty_cache()
