#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X hot path (BASELINE.json metric).

One "step" = one pass of the hot path over one synthetic annular-combustor case already resident in HBM:
P1 assembly of A, B, C (+ integer maps), the 16 pointwise flame vectors, and the whole
``fixed_point_iteration`` call (passive PEP solve + relaxed fixed-point loop, every factorisation / hierarchy
set-up included).  ``value`` = converged eigenpairs per second = (pairs the final solve returns, capped at
nev) x steps / wall time, summed over ranks.

Extra objects on the JSON line:
  roofline      the CSR SpMV of the shifted operator K inside the shift-invert Arnoldi loop, timed live with
                HIP events on the launch stream; achieved = (nnz*20 + n*36) bytes / average launch time.
  cpu_baseline  the CPU oracle (NumPy/SciPy restatement of the reference: explicit D, SuperLU re-factorisation
                every iteration, ARPACK) timed on this box's host on a bounded sample of the same workload.

Multi-GPU (--gpus N via torch.distributed.run): one process per GPU; see DESIGN.md "Multi-GPU".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "helmholtz-x_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

os.environ.setdefault("HX_QUIET", "1")
# The host side of the path is small dense algebra (projected problems of <= 50 columns): a BLAS thread pool sized for the
# box's 256 cores only spins there (measured: 38 s of CPU time instead of 20 s per 20 s of wall time, and 3 % slower) and
# would fight the other ranks of a multi-GPU run.
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "4")

WORKLOADS = {
    # name: (nr, ntheta, nz)  -> DOFs = (nr+1)*ntheta*(nz+1)
    "annulus-20M": (88, 2816, 76),     # 19,298,048 DOF: capacity point (one GPU, ~170 GB of HBM), not a BASELINE config
    "annulus-5M": (56, 1792, 48),      # 5,005,056 DOF (BASELINE configs[3], the metric's headline size)
    "annulus-1M": (32, 1024, 30),      # 1,047,552 DOF (configs[2])
    "annulus-260k": (20, 640, 18),     # 255,360 DOF
    "annulus-66k": (12, 384, 12),      # 64,896 DOF
    "annulus-34k": (10, 256, 11),      # 33,792 DOF ~ the reference's real MICCA mesh (34,787 DOF)
    "annulus-16k": (8, 160, 10),       # 15,840 DOF (bounded CPU-baseline sample)
}
HBM_PEAK_GBS = 8000.0                   # MI355X_MICROARCH.md: 8 TB/s spec


def build_case(shape):
    import numpy as np
    from scipy.io import loadmat
    from cases import Micca
    from helmholtz_x.flame_transfer_function import stateSpace
    from helmholtz_x.synthetic import annulus, micca_sound_speed
    mesh, subdomains, facet_tags = annulus(*shape)
    c = micca_sound_speed(mesh)
    import torch
    if torch.cuda.is_available():
        c.resident()                    # input field in HBM before the timed region (contract: inputs resident)
    mat = loadmat(os.path.join(ROOT, "tests", "golden", "micca_ftf.mat"))
    ftf = stateSpace(mat['A'], mat['b'], mat['c'], mat['d'])
    return dict(mesh=mesh, subdomains=subdomains, facet_tags=facet_tags, c=c, ftf=ftf, P=Micca, np=np)


def hot_path(case, st_type=None, nev=4, tol=1e-3, history=None, sharded=False):
    """One full pass: assembly + flame vectors + fixed-point iteration.  Returns (E, matrices).
    ``sharded``: collective row-sharded solve over the default torch.distributed group (theta-slab partition)."""
    from helmholtz_x.acoustic_matrices import AcousticMatrices
    from helmholtz_x.eigensolvers import fixed_point_iteration
    from helmholtz_x.flame_matrices import PointwiseFlameMatrix
    from helmholtz_x.parameters_utils import Q_multiple
    P, mesh = case["P"], case["mesh"]
    mesh.assembler().rebuild_maps()
    matrices = AcousticMatrices(mesh, case["facet_tags"], {11: {'Robin': P.R_outlet}}, case["c"])
    h = case.get("h")
    if h is None:
        h = case["h"] = Q_multiple(mesh, case["subdomains"], P.N_sector).resident()     # input field, kept in HBM
    D = PointwiseFlameMatrix(mesh, case["subdomains"], P.x_r(), h, P.rho_amb, P.q_0, P.u_b, case["ftf"])
    D.assemble_submatrices('direct')
    shard = None
    if sharded:
        from helmholtz_x.sharded import ShardContext
        x = mesh.geometry.x
        plane = int(np_count_plane(x))                     # rows per theta-plane: cut the slabs between planes
        shard = case["shard"] = ShardContext(matrices.A.shape[0], align=plane)
    E = fixed_point_iteration(matrices, D, P.target, i=0, nev=nev, tol=tol, st_type=st_type, history=history,
                              shard=shard)
    return E, matrices


def np_count_plane(x):
    """Vertices per theta-plane of the synthetic annulus (theta-major numbering: the first plane is theta = theta_0)."""
    import numpy as np
    th = np.arctan2(x[:, 1], x[:, 0])
    return int(np.argmax(np.abs(th - th[0]) > 1e-12))


def spmv_roofline(matrices, target, reps=50, in_situ=None):
    """Average launch time of the operator product y = K x (K = A + sigma B + sigma^2 C) as the inner solver runs it:
    SELL-64 layout (`k_sell<c128, 0>`, hx_sell_spmv_z).  achieved = CSR-algorithmic bytes (nnz*20 + n*36) /
    average launch duration, where the duration is measured IN SITU: HIP events on the launch stream around every such
    launch of the timed region (``in_situ`` = (launches, summed ms) from hx_profile_sell_spmv_read) -- the number
    rocprofv3's per-kernel average reproduces.  A back-to-back loop of the same kernel and of the plain CSR kernel is
    timed beside it (on some boxes the loop runs ~10 % faster than the in-situ launches)."""
    import ctypes as C
    import torch
    from helmholtz_x import _hx
    from helmholtz_x.petsc import combine_csr
    A, B, Cm = matrices.A, matrices.B, matrices.C
    K = combine_csr(A.shape, A.csr_terms + [(target * c, m) for c, m in B.csr_terms]
                    + [(target ** 2 * c, m) for c, m in Cm.csr_terms])
    n, nnz = K.shape[0], K.nnz
    x = torch.randn(n, dtype=torch.complex128, device=K.val.device)
    y = torch.empty_like(x)
    sell = C.c_void_p()
    _hx.check(_hx.lib.hx_sell_create(C.byref(sell), n, _hx.ptr(K.rowptr), _hx.ptr(K.col), _hx.ptr(K.val), 1, 0, _hx.stream()))
    one, zero = _hx.z2(1.0), _hx.z2(0.0)

    def timed(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    ms_csr = timed(lambda: K.spmv(x, y))
    ms = timed(lambda: _hx.check(_hx.lib.hx_sell_spmv_z(sell, _hx.ptr(x), _hx.ptr(y), one, zero, _hx.stream())))
    _hx.lib.hx_sell_destroy(sell)
    alg_bytes = nnz * 20 + n * 36
    loop_ms = ms
    if in_situ is not None and in_situ[0] > 0:
        ms = in_situ[1] / in_situ[0]
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    return dict(bound="hbm", kernel="k_sell<c128, 0> (hx_sell_spmv_z)", achieved=round(gbs, 1), peak=HBM_PEAK_GBS,
                unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4), traffic=measured_traffic(n, nnz), n_rows=n, nnz=nnz,
                bytes_per_launch=alg_bytes, avg_launch_ms=round(ms, 5),
                timing="in situ: HIP events around %d launches of the timed region" % in_situ[0]
                if (in_situ is not None and in_situ[0] > 0) else "back-to-back loop",
                back_to_back_loop_ms=round(loop_ms, 5),
                csr_kernel={"kernel": "k_spmv<8, 0, c128> (hx_csr_spmv_z)", "avg_launch_ms": round(ms_csr, 5),
                            "achieved": round(alg_bytes / (ms_csr * 1e-3) / 1e9, 1)})


def smoother_roofline(inner, n, in_situ):
    """The kernel with the largest share of the run time: one Chebyshev smoother step of the V-cycle on the fine level
    (`k_sell<c64, 1, half>`: x' = x + a D^-1 (b - K x) + c (x - x_prev) in complex64 on complex-half values with 16-bit
    column deltas, or no index at all in stencil slices).  Bytes per launch = what the kernel must stream: stored entries * 4 +
    indexed entries * index bytes + n * 5 complex64 vectors (b, x gathered once, x_prev, D^-1, x').  Same in-situ HIP-event timing as the operator product."""
    import ctypes as C
    from helmholtz_x import _hx
    sell = getattr(inner, "sell", None)
    if sell is None or in_situ[0] == 0:
        return None
    ns, padded = C.c_int32(0), C.c_int64(0)
    _hx.check(_hx.lib.hx_sell_info(sell, C.byref(ns), C.byref(padded)))
    idx_bytes = _hx.lib.hx_sell_index_bits(sell) // 8
    val_bytes = {"fp16": 4, "c64": 8, "c128": 16}.get(getattr(inner, "fine_values", "fp16"), 4)
    indexed = padded.value - _hx.lib.hx_sell_stencil_entries(sell)        # stencil slices carry no per-entry index
    nbytes = padded.value * val_bytes + indexed * idx_bytes + n * 5 * 8
    ms = in_situ[1] / in_situ[0]
    gbs = nbytes / (ms * 1e-3) / 1e9
    return dict(kernel="k_sell<c64, 1, half> (hx_sell_smooth_c64, fine level)", launches=in_situ[0],
                avg_launch_ms=round(ms, 5), bytes_per_launch=nbytes, achieved=round(gbs, 1), unit="GB/s",
                frac=round(gbs / HBM_PEAK_GBS, 4))


def measured_traffic(n, nnz):
    """HBM bytes per SpMV launch from the committed rocprofv3 PMC passes (profiles/spmv_traffic.json): 2 x FETCH_SIZE
    (gfx950 correction of MI355X_MICROARCH.md, HBM section) + WRITE_SIZE, for the matrix of exactly this size; else None."""
    try:
        with open(os.path.join(ROOT, "profiles", "spmv_traffic.json")) as fh:
            for rec in json.load(fh)["records"]:
                if rec["n_rows"] == n and rec["nnz"] == nnz and rec["kernel"].startswith("k_sell"):
                    return rec["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def cpu_baseline(sample="annulus-16k"):
    """Oracle ("port") fixed-point iteration on the host, bounded sample of the same workload."""
    import numpy as np
    from oracle import eigen, fem
    case = build_case(WORKLOADS[sample])
    P, mesh = case["P"], case["mesh"]
    x, cells = mesh.geometry.x, mesh.topology.cells
    n = x.shape[0]
    t0 = time.perf_counter()
    c = case["c"].x.array
    A = fem.assemble_A(x, cells, c, c_is_cellwise=True)
    C = fem.assemble_C(x, cells)
    sel = case["facet_tags"].find(11)
    B = fem.assemble_B(x, n, [(mesh.facets[sel], fem.robin_Z(P.R_outlet), c, mesh.facet_cells[sel])])
    tags = np.full(cells.shape[0], -1, np.int32)
    tags[case["subdomains"].indices] = case["subdomains"].values
    q = fem.q_multiple(x, cells, tags, P.N_sector)
    D = None
    for f, pt in enumerate(P.x_r()):
        left = fem.threshold_pack(fem.pointwise_left(x, cells, tags, f, q, 1.4, P.q_0, P.u_b), 1e-10)
        Df = fem.flame_outer(n, left, fem.pointwise_right(x, cells, pt, P.rho_amb))
        D = Df if D is None else D + Df
    E, hist = eigen.fixed_point_iteration(A, B, C, eigen.OracleFlame(D, case["ftf"]), P.target, nev=4, i=0, tol=1e-3)
    dt = time.perf_counter() - t0
    return dict(value=round(4 / dt, 5), unit="eigenpairs/s", cores=1, kind="port",
                sample="%s (%d DOF, %d tets), 1 full fixed_point_iteration (nev=4, tol=1e-3, %d iterations), "
                       "SciPy SuperLU + ARPACK, %.1f s" % (sample, n, cells.shape[0], len(hist) - 1, dt),
                omega=[hist[-1].real, hist[-1].imag])


_RESULT_FD = None


def emit(out):
    """The result line, on the process's ORIGINAL stdout (see main)."""
    line = (json.dumps(out) + "\n").encode()
    if _RESULT_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_RESULT_FD, line)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--workload", default=os.environ.get("HX_BENCH_WORKLOAD", "annulus-5M"))
    ap.add_argument("--st-type", default=None, help="inner solver for the shifted operator: lu | amg | auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", default="annulus-16k")
    args = ap.parse_args()

    # Exactly ONE line on stdout: the process's file descriptor 1 is pointed at stderr for the whole run (communication
    # libraries print banners and warnings to stdout from C++: gloo's "[Gloo] Rank 0 is connected ...", RCCL's WARN lines),
    # and the JSON line is written to the saved descriptor at the end.
    sys.stdout.flush()
    global _RESULT_FD
    _RESULT_FD = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    local_rank %= torch.cuda.device_count()       # (rehearsals with more ranks than GPUs share a device)
    torch.cuda.set_device(local_rank)
    if world > 1:
        backend = os.environ.get("HX_DIST_BACKEND", "nccl")     # "gloo" only for single-GPU rehearsals of the N>1 path
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    shape = WORKLOADS[args.workload]
    case = build_case(shape)
    mesh = case["mesh"]
    mesh.assembler()                           # mesh arrays -> HBM (outside the timed region)
    n_dof, n_cells = mesh.num_vertices, mesh.num_cells

    import contextlib
    with contextlib.redirect_stdout(sys.stderr):          # the library logs like the reference; stdout = the JSON line
        import gc
        for _ in range(args.warmup):
            hot_path(case, args.st_type)
            gc.collect()
        from helmholtz_x import _hx
        import ctypes as C
        # in-situ HIP-event timing of the roofline kernel (operator product on the fine operator) over the timed region
        _hx.check(_hx.lib.hx_profile_sell_spmv(1, n_dof if n_dof >= 50_000 else 2 ** 31 - 1))
        barrier()
        t0 = time.perf_counter()
        pairs = 0
        hist = []
        E = matrices = None
        for _ in range(args.steps):
            hist = []
            if E is not None:                     # release the previous call's solver objects first, as a driver that
                E.destroy()                       # follows the reference does (eigensolvers.py:161 E.destroy())
            E = matrices = None
            gc.collect()
            E, matrices = hot_path(case, args.st_type, history=hist)
            pairs += min(E.getConverged(), 4)
        barrier()
        dt = time.perf_counter() - t0
        n_launch, tot_ms = C.c_int64(0), C.c_double(0.0)
        _hx.check(_hx.lib.hx_profile_sell_spmv_read(C.byref(n_launch), C.byref(tot_ms)))
        n_sm, sm_ms = C.c_int64(0), C.c_double(0.0)
        _hx.check(_hx.lib.hx_profile_sell_read(1, C.byref(n_sm), C.byref(sm_ms)))
        _hx.check(_hx.lib.hx_profile_sell_spmv(0, 0))
        in_situ = (n_launch.value, tot_ms.value)
        in_situ_smoother = (n_sm.value, sm_ms.value)
        free_b, total_b = torch.cuda.mem_get_info()            # device-wide: torch's pool + the library's own hipMallocs
        hbm_used_gb = round((total_b - free_b) / 2 ** 30, 1)
    red_dev = "cuda" if (world == 1 or dist.get_backend() == "nccl") else "cpu"
    tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    tot = torch.tensor([float(pairs)], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    dt, pairs_all = float(tmax.item()), float(tot.item())

    omegas = [E.getEigenvalue(i) for i in range(min(E.getConverged(), 4))]
    out = None
    if rank == 0:
        inner = next((v for v in (E.inner_cache or {}).values() if hasattr(v, "stats")), None)
        with contextlib.redirect_stdout(sys.stderr):
            roof = spmv_roofline(matrices, case["P"].target, in_situ=in_situ)
            roof["smoother_kernel"] = smoother_roofline(inner, n_dof, in_situ_smoother)
        out = {
            "metric": "converged eigenpairs/sec + SpMV GB/s (% HBM roofline), AnnularCombustor",
            "value": round(pairs_all / dt, 5), "unit": "eigenpairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / max(args.steps, 1), 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "c128", "data": "synthetic",
            "config": {"workload": args.workload, "grid_nr_ntheta_nz": list(shape), "dofs": n_dof, "tets": n_cells,
                       "nev": 4, "fpi_tol": 1e-3, "fpi_iterations": len(hist) - 1, "flames": 16,
                       "inner_solver": getattr(inner, "name", None),
                       "inner_stats": inner.stats() if inner is not None else None,
                       "krylov": {k: v for k, v in (E.state or {}).items() if not k.startswith(("start", "_"))},
                       "parallelism": "replicas x%d" % world if world > 1 else "1 gpu",
                       "hbm_in_use_after_step_gib": hbm_used_gb,
                       "omega": [[round(w.real, 6), round(w.imag, 6)] for w in omegas]},
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1:            # the CPU baseline is an N = 1 leg
            with contextlib.redirect_stdout(sys.stderr):
                out["cpu_baseline"] = cpu_baseline(args.cpu_sample)
    if world > 1 and os.environ.get("HX_BENCH_SHARDED", "1") != "0":
        # Beside the timed replicas: ONE row-sharded solve of the same workload over all N ranks (halo P2P + all-reduce,
        # helmholtz_x/sharded.py).  Reported as an extra object; it never enters `value`.
        sharded = sharded_solve_extra(case, omegas, barrier, rank, out)
        if rank == 0:
            out["sharded_solve"] = sharded
    if rank == 0:
        emit(out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def sharded_solve_extra(case, ref_omegas, barrier, rank, out):
    """Collective row-sharded fixed-point solve, guarded by a watchdog: this leg exercises the RCCL exchange path,
    which cannot be rehearsed on a single-GPU box (RCCL refuses two ranks on one device), so a stall must not cost
    the replicas result -- on time-out rank 0 prints the JSON line with the error recorded and every rank exits."""
    import contextlib
    import threading
    budget = float(os.environ.get("HX_BENCH_SHARDED_TIMEOUT", "240"))

    def fire():
        if rank == 0:
            out["sharded_solve"] = {"error": "timed out after %.0f s" % budget}
            emit(out)
        os._exit(0)

    dog = threading.Timer(budget, fire)
    dog.daemon = True
    dog.start()
    res = {}
    try:
        with contextlib.redirect_stdout(sys.stderr):
            barrier()
            t0 = time.perf_counter()
            E, _ = hot_path(case, "amg", sharded=True)
            barrier()
            dt = time.perf_counter() - t0
        sh = case["shard"]
        inner = next((v for v in (E.inner_cache or {}).values() if hasattr(v, "stats")), None)
        om = [E.getEigenvalue(i) for i in range(min(E.getConverged(), 4))]
        diff = max((abs(a - b) / max(abs(b), 1e3) for a, b in zip(om, ref_omegas)), default=None)
        res = {"seconds": round(dt, 3), "eigenpairs_per_s": round(min(E.getConverged(), 4) / dt, 5),
               "partition": "theta-slabs, %d rows/rank + %d ghost rows" % (sh.n_own, sh.n_ext - sh.n_own),
               "halo_exchanges": sh.plan.n_exchanges, "all_reduces": sh.n_allreduce,
               "max_rel_diff_vs_replica_eigenvalues": diff,
               "inner_stats": inner.stats() if inner is not None else None,
               "krylov_ops": (E.state or {}).get("krylov_ops")}
    except Exception as e:                                    # recorded, never fatal for the replicas line
        # the other ranks may be waiting in a collective this rank left: finish here instead of joining a last barrier
        if rank == 0:
            out["sharded_solve"] = {"error": repr(e)[:500]}
            emit(out)
        sys.stderr.flush()
        os._exit(0)
    dog.cancel()
    return res


if __name__ == "__main__":
    main()
