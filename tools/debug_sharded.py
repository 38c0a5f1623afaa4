#!/usr/bin/env python3
"""Two ranks on one GPU: inner-solve convergence of the sharded hierarchy against the single-GPU one on the same operator.
usage: python tools/debug_sharded.py [nr nth nz]   (spawns 2 ranks, gloo)"""
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "helmholtz-x_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ.setdefault("HX_QUIET", "1")

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port, shape):
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import bench
    from helmholtz_x import sharded as shd
    from helmholtz_x.acoustic_matrices import AcousticMatrices
    from helmholtz_x.amg import AMGSolver
    from helmholtz_x.petsc import combine_csr
    case = bench.build_case(shape)
    P = case["P"]

    def shifted(M):
        A, B, Cm = M.A, M.B, M.C
        return combine_csr(A.shape, A.csr_terms + [(P.target * c, m) for c, m in B.csr_terms]
                           + [(P.target ** 2 * c, m) for c, m in Cm.csr_terms])

    Mg = AcousticMatrices(case["mesh"], case["facet_tags"], {11: {'Robin': P.R_outlet}}, case["c"])
    Kg = shifted(Mg)
    n = Kg.shape[0]
    rng = np.random.default_rng(0)
    bg = torch.from_numpy(rng.standard_normal(n) + 1j * rng.standard_normal(n)).cuda()
    single = AMGSolver(Kg)
    xg = torch.zeros_like(bg)
    single.solve(bg, xg)
    if rank == 0:
        print("single :", single.stats()["levels"], "its", single.last, [round(lv.get("rho", 0), 3) for lv in single.levels], flush=True)
    d = shd.distributed_case(case)
    sh = d["shard"]
    Ml = AcousticMatrices(d["mesh"], d["facet_tags"], {11: {'Robin': P.R_outlet}}, d["c"])
    Kl = shifted(Ml)
    # the local rows equal the global rows
    ref = sh.localize(Kg) if False else None
    Ks = Kg.to_scipy()[sh.r0:sh.r1]
    l2g = d["mesh"].part["local_to_global"]
    Kloc = Kl.to_scipy()
    import scipy.sparse as sp
    Pm = sp.csr_matrix((np.ones(l2g.size), (np.arange(l2g.size), l2g)), shape=(l2g.size, n))
    print("rank", rank, "local rows vs global rows:", abs(Kloc @ Pm - Ks).max() / abs(Ks).max(), flush=True)
    os.environ["HX_GMRES_TRACE"] = "1" if rank == 0 and os.environ.get("TRACE") else ""
    sharded = AMGSolver(Kl, shard=sh)
    xl = torch.zeros(sh.n_own, dtype=torch.complex128, device="cuda")
    try:
        sharded.solve(bg[sh.r0:sh.r1].contiguous(), xl)
    except ArithmeticError as e:
        print("rank", rank, "solve failed:", e, flush=True)
    err = float(torch.linalg.vector_norm(xl - xg[sh.r0:sh.r1]) / torch.linalg.vector_norm(xg[sh.r0:sh.r1]))
    if rank == 0:
        print("sharded:", sharded.stats()["levels"], sharded.stats()["sharded_levels"], "its", sharded.last,
              [round(lv.get("rho", 0), 3) for lv in sharded.levels], "x diff", err, flush=True)
        for a, b in zip(single.levels, sharded.levels):
            print("   level nnz single %d  sharded(local) %d  n %d / %d" % (a["K"].nnz, b["K"].nnz, a["n"], b["n"]), flush=True)
    # V-cycle as an operator: compare M b (sharded) with the single-GPU cycle on the same vector -- differs (other
    # aggregates) but must be of the same quality: |b - K M b| / |b|
    from helmholtz_x import _hx
    z = torch.zeros_like(bg)
    _hx.check(_hx.lib.hx_amg_vcycle(single.handle, _hx.ptr(bg), _hx.ptr(z), _hx.stream()))
    r = torch.empty_like(bg)
    Kg.spmv(z, r)
    q1 = float(torch.linalg.vector_norm(bg - r) / torch.linalg.vector_norm(bg))
    zl = torch.zeros(sh.n_own, dtype=torch.complex128, device="cuda")
    bl = bg[sh.r0:sh.r1].contiguous()
    _hx.check(_hx.lib.hx_amg_vcycle(sharded.handle, _hx.ptr(bl), _hx.ptr(zl), _hx.stream()))
    ze = sh.ext_vector()
    ze[:sh.n_own] = zl
    sh.halo(ze)
    rl = torch.empty_like(zl)
    Kl.spmv(ze, rl)
    t = torch.stack([torch.linalg.vector_norm(bl - rl) ** 2, torch.linalg.vector_norm(bl) ** 2]).to(torch.float64)
    sh.all_reduce_(t)
    if rank == 0:
        print("|b - K M b|/|b|  single %.4f  sharded %.4f" % (q1, float(torch.sqrt(t[0] / t[1]))), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    shape = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (10, 192, 11)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(worker, args=(2, port, shape), nprocs=2, join=True)
