#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X hot path (BASELINE.json metric).

One "step" = one pass of the hot path over one synthetic annular-combustor case already resident in HBM:
P1 assembly of A, B, C (+ integer maps), the 16 pointwise flame vectors, and the whole
``fixed_point_iteration`` call (passive PEP solve + relaxed fixed-point loop, every factorisation / hierarchy
set-up included).  ``value`` = converged eigenpairs per second = (pairs the final solve returns, capped at
nev) x steps / wall time.

N = 1:  one GPU runs the step.
N > 1:  the SAME problem is partitioned over the N ranks (theta-slab row ranges of the mesh; what the reference gets
        from ``mpirun -np N``, fullAnnulus/runAll.sh:9-10): every rank assembles and solves its rows, halo exchanges and
        small all-reduces over RCCL.  ``value`` = eigenpairs/s of that ONE row-sharded solve ("scaling": "strong").
        ``python bench.py --gpus N`` without a launcher starts the N ranks itself (torch.distributed.run, before any
        GPU call); under the driver's own ``python -m torch.distributed.run ... bench.py --gpus N`` the ranks are there
        already.  A failed or stalled collective leg exits NON-ZERO after printing its record.

Extra objects on the JSON line:
  roofline      the operator product y = K x of the shifted operator inside the shift-invert Arnoldi loop, timed live
                with HIP events on the launch stream.  achieved / frac = CSR-algorithmic bytes (nnz*20 + n*36, SURVEY
                8d) / average launch time; beside it the bytes the stored SELL-64 format must move (format_bytes,
                frac_format_bytes), the PMC traffic and its ratio, and ``dominant_by_time`` = the kernel with the
                largest share of the run time (fine-level smoother step) with its own byte count and PMC ratio.
  cpu_baseline  the CPU oracle (NumPy/SciPy restatement of the reference: explicit D, SuperLU re-factorisation
                every iteration, ARPACK) timed on this box's host on a bounded sample of the same workload, with the
                GPU path's time on that SAME sample beside it.
"""
import argparse
import json
import os
import socket
import subprocess
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
BLAS_THREADS = 4
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, str(BLAS_THREADS))

WORKLOADS = {
    # name: (nr, ntheta, nz)  -> DOFs = (nr+1)*ntheta*(nz+1)
    "annulus-25M": (96, 3072, 83),     # 25,030,656 DOF: capacity point after the round-4 memory work (one GPU), not a BASELINE config
    "annulus-20M": (88, 2816, 76),     # 19,298,048 DOF: capacity point (one GPU, 155 GiB of HBM), not a BASELINE config
    "annulus-5M": (56, 1792, 48),      # 5,005,056 DOF (BASELINE configs[3], the metric's headline size)
    "annulus-5M-iso": (42, 666, 176),  # 5,068,926 DOF with near-cubic cells (h = 1.65 mm in r, theta, z): what an unstructured
                                       # mesher produces; the headline grid above has 10:1 cells (0.6 x 1.25 x 6 mm)
    "annulus-1M": (32, 1024, 30),      # 1,047,552 DOF (configs[2])
    "annulus-260k": (20, 640, 18),     # 255,360 DOF
    "annulus-66k": (12, 384, 12),      # 64,896 DOF
    "annulus-34k": (10, 256, 11),      # 33,792 DOF ~ the reference's real MICCA mesh (34,787 DOF)
    "annulus-16k": (8, 160, 10),       # 15,840 DOF
    "annulus-8k": (6, 128, 8),         # 8,064 DOF   (the small sizes: crossover of the direct and the iterative inner solver)
    "annulus-4k": (5, 96, 6),          # 4,032 DOF
    "annulus-2k": (4, 64, 6),          # 2,240 DOF
    # the reference's REAL MICCA mesh (tests/golden/meshes/micca_full: 34,787 vertices, unstructured, the mesh of its logged
    # run) as it is, and refined uniformly once / twice (synthetic.refine_tets): 0.25 M / 1.9 M DOF with the real geometry
    # and none of the generator's regularity; vertices renumbered by reverse Cuthill-McKee as XDMFReader does for large meshes
    "micca-real": ("real", 0),
    "micca-real-r1": ("real", 1),
    "micca-real-r2": ("real", 2),
}
HBM_PEAK_GBS = 8000.0                   # MI355X_MICROARCH.md: 8 TB/s spec
METRIC = "converged eigenpairs/sec + SpMV GB/s (% HBM roofline), AnnularCombustor 5M DOF"


def build_case(shape, numbering="native", jitter=0.0, resident=True, slab_only=False):
    """The synthetic annular combustor.  ``numbering``: "native" = theta-major grid numbering of the generator (periodic,
    extruded: what a structured mesher writes); "rcm" = a random numbering passed through reverse Cuthill-McKee (what a
    bandwidth-reducing reader such as DOLFINx's leaves of an unstructured mesh file); "morton" = vertices renumbered along
    a space-filling curve; "random" = a random permutation (worst case).  ``jitter``: interior nodes displaced by +-jitter*h (numpy.random.default_rng(0)), so that no two element matrices
    are equal either."""
    import numpy as np
    from scipy.io import loadmat
    from cases import Micca
    from helmholtz_x.fem import Mesh, MeshTags, locality_permutation
    from helmholtz_x.flame_transfer_function import stateSpace
    from helmholtz_x.synthetic import annulus, micca_sound_speed
    if shape[0] == "real":
        return build_case_real(shape[1], resident)
    if slab_only and numbering == "native" and jitter == 0.0:
        # N > 1 ranks on the analytic annulus: no rank builds the global mesh; helmholtz_x.sharded.distributed_case lets every
        # rank generate its own theta-slab (synthetic.annulus_part)
        mat = loadmat(os.path.join(ROOT, "tests", "golden", "micca_ftf.mat"))
        return dict(mesh=None, subdomains=None, facet_tags=None, c=None, ftf=stateSpace(mat['A'], mat['b'], mat['c'], mat['d']),
                    P=Micca, np=np, numbering="native", jitter=0.0, generator=tuple(shape))
    mesh, subdomains, facet_tags = annulus(*shape)
    if jitter > 0.0:
        x = mesh.geometry.x.copy()
        nr, nth, nz = shape
        h = min(0.07 / nr, 0.29 / nz, 2 * np.pi * 0.14 / nth)
        r = np.hypot(x[:, 0], x[:, 1])
        interior = (r > 0.14 + 1e-9) & (r < 0.21 - 1e-9) & (x[:, 2] > -0.09 + 1e-9) & (x[:, 2] < 0.2 - 1e-9)
        x[interior] += np.random.default_rng(0).uniform(-jitter * h, jitter * h, size=(int(interior.sum()), 3))
        mesh = Mesh(x, mesh.topology.cells, facets=mesh.facets, facet_cells=mesh.facet_cells)
    if numbering != "native":
        n = mesh.num_vertices
        if numbering == "morton":
            perm = locality_permutation(mesh.geometry.x)
        elif numbering in ("random", "rcm"):
            perm = np.random.default_rng(1).permutation(n).astype(np.int64)
        else:
            raise ValueError("numbering must be native, rcm, morton or random")
        mesh = mesh.renumbered(perm)
        if numbering == "rcm":
            # what a bandwidth-reducing dof reordering (DOLFINx applies one to every mesh it reads) leaves of an arbitrary
            # file numbering: reverse Cuthill-McKee of the vertex graph, here taken from the P1 sparsity pattern the
            # assembler builds on the device for the randomly numbered mesh (host graph pass, outside the timed region)
            mesh.assembler()                               # (pattern on the device: the graph of the host pass)
            mesh = mesh.locality_renumbered("rcm")         # fem.bandwidth_permutation: what XDMFReader applies at ingestion
    if jitter > 0.0 or numbering != "native":       # cells and tagged facets keep their order: tags carry over
        facet_tags = MeshTags(mesh, facet_tags.dim, facet_tags.indices, facet_tags.values)
        subdomains = MeshTags(mesh, subdomains.dim, subdomains.indices, subdomains.values)
    c = micca_sound_speed(mesh)
    import torch
    if resident and torch.cuda.is_available():
        c.resident()                    # input field in HBM before the timed region (contract: inputs resident)
    mat = loadmat(os.path.join(ROOT, "tests", "golden", "micca_ftf.mat"))
    ftf = stateSpace(mat['A'], mat['b'], mat['c'], mat['d'])
    return dict(mesh=mesh, subdomains=subdomains, facet_tags=facet_tags, c=c, ftf=ftf, P=Micca, np=np,
                numbering=numbering, jitter=jitter)


def build_case_real(refinements, resident=True):
    """The reference's own MICCA mesh (read by the package's XDMF / HDF5 reader), refined uniformly ``refinements`` times,
    with the set-up of its active_fpi.py: DG0 sound speed by the cell-midpoint rule, 16 flames, Robin outlet."""
    import numpy as np
    from scipy.io import loadmat
    from cases import Micca
    from helmholtz_x.fem import Function, FunctionSpace, MeshTags
    from helmholtz_x.flame_transfer_function import stateSpace
    from helmholtz_x.io_utils import XDMFReader
    from helmholtz_x.synthetic import refine_tets
    mesh, subdomains, facet_tags = XDMFReader(os.path.join(ROOT, "tests", "golden", "meshes", "micca_full", "mesh")).getAll()
    for _ in range(refinements):
        mesh, subdomains, facet_tags = refine_tets(mesh, subdomains, facet_tags)
    numbering = "file"
    if refinements > 0:                                     # what the reader does to a mesh of this size (>= 200 k vertices)
        import torch
        if torch.cuda.is_available():
            mesh.assembler()                                # (pattern on the device: the graph of the host pass)
        mesh = mesh.locality_renumbered("rcm")
        facet_tags = MeshTags(mesh, facet_tags.dim, facet_tags.indices, facet_tags.values)
        subdomains = MeshTags(mesh, subdomains.dim, subdomains.indices, subdomains.values)
        numbering = "rcm"
    c = Function(FunctionSpace(mesh, "DG", 0), "soundspeed")
    c.x.array[:] = Micca.c_of_z(c.function_space.tabulate_dof_coordinates()[:, 2])
    import torch
    if resident and torch.cuda.is_available():
        c.resident()
    mat = loadmat(os.path.join(ROOT, "tests", "golden", "micca_ftf.mat"))
    ftf = stateSpace(mat['A'], mat['b'], mat['c'], mat['d'])
    return dict(mesh=mesh, subdomains=subdomains, facet_tags=facet_tags, c=c, ftf=ftf, P=Micca, np=np,
                numbering=numbering, jitter=0.0)


def hot_path(case, st_type=None, nev=4, tol=1e-3, history=None, sharded=False):
    """One full pass: assembly + flame vectors + fixed-point iteration.  Returns (E, matrices).
    ``sharded``: collective solve of the ONE problem partitioned over the default torch.distributed group (theta-slab row
    ranges; every rank assembles and solves its own rows)."""
    from helmholtz_x.acoustic_matrices import AcousticMatrices
    from helmholtz_x.eigensolvers import fixed_point_iteration
    from helmholtz_x.flame_matrices import PointwiseFlameMatrix
    from helmholtz_x.parameters_utils import Q_multiple
    P = case["P"]
    if sharded:
        from helmholtz_x import sharded as shd
        return shd.bench_hot_path(case, st_type, nev, tol, history)
    mesh = case["mesh"]
    mesh.assembler().rebuild_maps()
    matrices = AcousticMatrices(mesh, case["facet_tags"], {11: {'Robin': P.R_outlet}}, case["c"])
    h = case.get("h")
    if h is None:
        h = case["h"] = Q_multiple(mesh, case["subdomains"], P.N_sector).resident()     # input field, kept in HBM
    D = PointwiseFlameMatrix(mesh, case["subdomains"], P.x_r(), h, P.rho_amb, P.q_0, P.u_b, case["ftf"])
    D.assemble_submatrices('direct')
    E = fixed_point_iteration(matrices, D, P.target, i=0, nev=nev, tol=tol, st_type=st_type, history=history)
    return E, matrices


def np_count_plane(x):
    """Vertices per theta-plane of the synthetic annulus (theta-major numbering: the first plane is theta = theta_0)."""
    import numpy as np
    th = np.arctan2(x[:, 1], x[:, 0])
    return int(np.argmax(np.abs(th - th[0]) > 1e-12))


def shifted_operator(matrices, target):
    from helmholtz_x.petsc import combine_csr
    A, B, Cm = matrices.A, matrices.B, matrices.C
    return combine_csr(A.shape, A.csr_terms + [(target * c, m) for c, m in B.csr_terms]
                       + [(target ** 2 * c, m) for c, m in Cm.csr_terms])


def spmv_roofline(K, reps=50, in_situ=None, pmc=None):
    """Average launch time of the operator product y = K x (K = A + sigma B + sigma^2 C; on N > 1 ranks: this rank's rows
    over [owned | ghost] columns) as the inner solver runs it: SELL-64 layout (`k_sell<c128, 0>`, hx_sell_spmv_z).
    The duration is measured IN SITU: HIP events on the launch stream around every such launch of the timed region
    (``in_situ`` = (launches, summed ms) from hx_profile_sell_spmv_read) -- the number rocprofv3's per-kernel average
    reproduces.  Two byte counts over that duration:
      achieved / frac                    CSR-ALGORITHMIC bytes nnz*20 + n*36 (values + int32 columns, row
                                         pointer, x read once, y written once) -- the contract's figure;
      achieved_format / frac_format_bytes  the bytes the stored SELL-64 format itself must move: padded entries * 16 +
                                         indexed entries * index bytes (stencil slices carry NO per-entry index) + n * 32.
    A back-to-back loop of the same kernel and of the plain CSR kernel is timed beside it."""
    import ctypes as C
    import torch
    from helmholtz_x import _hx
    n, nnz = K.shape[0], K.nnz
    x = torch.randn(K.shape[1], dtype=torch.complex128, device=K.val.device)
    y = torch.empty(n, dtype=torch.complex128, device=K.val.device)
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
    ns, padded = C.c_int32(0), C.c_int64(0)
    _hx.check(_hx.lib.hx_sell_info(sell, C.byref(ns), C.byref(padded)))
    idx_bytes = _hx.lib.hx_sell_index_bits(sell) // 8
    stencil_entries = _hx.lib.hx_sell_stencil_entries(sell)
    stencil_slices = _hx.lib.hx_sell_stencil_slices(sell)
    tail, split = C.c_int64(0), C.c_int32(0)
    _hx.check(_hx.lib.hx_sell_tail_info(sell, C.byref(tail), C.byref(split)))
    _hx.lib.hx_sell_destroy(sell)
    alg_bytes = nnz * 20 + n * 36
    # (overflow tails: value + 32-bit column per entry, 2 bytes per row of a split slice)
    fmt_bytes = padded.value * 16 + (padded.value - stencil_entries) * idx_bytes + n * 32 + tail.value * 20 + split.value * 128
    loop_ms = ms
    if in_situ is not None and in_situ[0] > 0:
        ms = in_situ[1] / in_situ[0]
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    gbs_fmt = fmt_bytes / (ms * 1e-3) / 1e9
    traffic = measured_traffic(n, nnz, pmc)
    prof = profile_record(n, nnz, pmc)
    prof_us = prof.get("rocprof_avg_launch_us") if prof else None
    return dict(bound="hbm", kernel="k_sell<c128, 0> (hx_sell_spmv_z)", achieved=round(gbs, 1), peak=HBM_PEAK_GBS,
                unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4),
                # the same byte count over the average duration of the COMMITTED rocprofv3 kernel statistics (another box, another
                # day: the two boxes' numbers side by side)
                frac_profile=None if not prof_us else round(alg_bytes / (prof_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                profile_avg_launch_ms=None if not prof_us else round(prof_us * 1e-3, 5),
                profile_source=None if not prof else prof.get("rocprof_source"),
                traffic=traffic, n_rows=n, nnz=nnz,
                bytes_per_launch=alg_bytes, bytes_model="CSR-algorithmic: nnz*20 + n*36 (SURVEY 8d)",
                format_bytes_per_launch=fmt_bytes,
                format="SELL-64: %d slices (%d index-free stencil slices), %d padded entries, %d-bit indices elsewhere%s"
                       % (ns.value, stencil_slices, padded.value, 8 * idx_bytes,
                          "; %d entries in the overflow tails of %d slices" % (tail.value, split.value) if tail.value else ""),
                achieved_format=round(gbs_fmt, 1), frac_format_bytes=round(gbs_fmt / HBM_PEAK_GBS, 4),
                traffic_ratio_vs_algorithmic=None if not traffic else round(traffic / alg_bytes, 3),
                traffic_ratio_vs_format=None if not traffic else round(traffic / fmt_bytes, 3),
                avg_launch_ms=round(ms, 5),
                timing="in situ: HIP events around %d launches of the timed region" % in_situ[0]
                if (in_situ is not None and in_situ[0] > 0) else "back-to-back loop",
                in_situ_launches_not_timed=(in_situ[2] if in_situ is not None and len(in_situ) > 2 else None),
                back_to_back_loop_ms=round(loop_ms, 5),
                csr_kernel={"kernel": "k_spmv<8, 0, c128> (hx_csr_spmv_z)", "avg_launch_ms": round(ms_csr, 5),
                            "achieved": round(alg_bytes / (ms_csr * 1e-3) / 1e9, 1),
                            "frac": round(alg_bytes / (ms_csr * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})


def smoother_roofline(inner, n, in_situ, numbering="native"):
    """The kernel with the largest share of the run time: one Chebyshev smoother step of the V-cycle on the fine level
    (`k_sell<c64, 1, half>`: x' = x + a D^-1 (b - K x) + c (x - x_prev) in complex64 on complex-half values with 16-bit
    column deltas, or no index at all in stencil slices).  Bytes per launch = what the kernel must stream: stored entries * 4 +
    indexed entries * index bytes + n * 5 complex64 vectors (b, x gathered once, x_prev, D^-1, x').  Same in-situ HIP-event
    timing as the operator product; the PMC traffic of the same kernel (committed profile) and its ratio beside it."""
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
    tail, split = C.c_int64(0), C.c_int32(0)
    _hx.check(_hx.lib.hx_sell_tail_info(sell, C.byref(tail), C.byref(split)))
    nbytes = padded.value * val_bytes + indexed * idx_bytes + n * 5 * 8 + tail.value * (val_bytes + 4) + split.value * 128
    ms = in_situ[1] / in_situ[0]
    gbs = nbytes / (ms * 1e-3) / 1e9
    traffic = measured_traffic(n, None, numbering, kernel="k_sell<c64,1,half>")
    return dict(kernel="k_sell<c64, 1, half> (hx_sell_smooth_c64, fine level)", launches=in_situ[0],
                timing="in situ: HIP events around every 8th launch of the timed region (the roofline kernel above: every launch)",
                avg_launch_ms=round(ms, 5), bytes_per_launch=nbytes,
                bytes_model="stored SELL-64 entries * 4 + indexed entries * index bytes + 5 complex64 vectors",
                achieved=round(gbs, 1), unit="GB/s",
                frac=round(gbs / HBM_PEAK_GBS, 4), traffic=traffic,
                traffic_ratio=None if not traffic else round(traffic / nbytes, 3))


def profile_record(n, nnz, pmc=None, kernel="k_sell"):
    """The committed profile record (profiles/spmv_traffic.json) of ``kernel`` for the matrix of exactly this size; else None."""
    try:
        with open(os.path.join(ROOT, "profiles", "spmv_traffic.json")) as fh:
            for rec in json.load(fh)["records"]:
                if rec["n_rows"] == n and (nnz is None or rec["nnz"] == nnz) and rec["kernel"].startswith(kernel) \
                        and (pmc is None or rec.get("numbering", "native") == pmc):
                    return rec
    except (OSError, KeyError, ValueError):
        pass
    return None


def measured_traffic(n, nnz, pmc=None, kernel="k_sell"):
    """HBM bytes per launch from the committed rocprofv3 PMC passes: 2 x FETCH_SIZE (gfx950 correction of
    MI355X_MICROARCH.md, HBM section) + WRITE_SIZE, for the matrix of exactly this size; else None."""
    rec = profile_record(n, nnz, pmc, kernel)
    return rec["hbm_bytes_per_launch"] if rec else None


def cpu_baseline(sample="micca-real", gpu_same_sample=True):
    """Oracle ("port") fixed-point iteration on the host, bounded sample of the same workload; beside it the GPU path's time
    on that SAME sample.  Default sample: the reference's own MICCA mesh (34,787 DOF, its flagship run: same set-up, same
    eigenvalue), for which the reference committed its own wall-clock times -- quoted in ``reference_log``."""
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
    out = dict(value=round(4 / dt, 5), unit="eigenpairs/s", cores=BLAS_THREADS, blas_threads=BLAS_THREADS,
               host_cores=os.cpu_count(), kind="port",
               note="SciPy SuperLU factorises on one core (sequential library); BLAS / LAPACK pools capped at %d threads "
                    "(cores = threads actually used; host_cores = os.cpu_count() of the box)" % BLAS_THREADS,
               sample="%s (%d DOF, %d tets), 1 full fixed_point_iteration (nev=4, tol=1e-3, %d iterations), "
                      "SciPy SuperLU + ARPACK, %.1f s" % (sample, n, cells.shape[0], len(hist) - 1, dt),
               seconds=round(dt, 2), omega=[hist[-1].real, hist[-1].imag])
    if sample == "micca-real":
        out["reference_log"] = {"seconds_1_rank": 125.6, "seconds_8_ranks": 16.4, "hardware": "unknown CPU",
                                "source": "BASELINE.md: Results/Active/FPI/active.log:99, activeParallel.log:99 "
                                          "(Total Execution Time of this case: same mesh, nev = 4, 1 + 8 PEP solves)"}
    if gpu_same_sample:
        import gc
        import torch
        hot_path(case, None)                        # warm-up (first launches, allocator)
        gc.collect()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        h2 = []
        E2, _ = hot_path(case, None, history=h2)
        torch.cuda.synchronize()
        dg = time.perf_counter() - t0
        out["gpu_same_sample"] = dict(seconds=round(dg, 3), value=round(min(E2.getConverged(), 4) / dg, 4),
                                      unit="eigenpairs/s", omega=[h2[-1].real, h2[-1].imag],
                                      rel_diff_vs_cpu=abs(h2[-1] - hist[-1]) / abs(hist[-1]))
    return out


_RESULT_FD = None


def emit(out):
    """The result line, on the process's ORIGINAL stdout (see main)."""
    line = (json.dumps(out) + "\n").encode()
    if _RESULT_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_RESULT_FD, line)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)      # (no flags: one warm-up call + three timed ones, a few minutes with the CPU sample)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("HX_BENCH_WORKLOAD", "annulus-5M"))
    ap.add_argument("--st-type", default=None, help="inner solver for the shifted operator: lu | amg | auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", default="micca-real",
                    help="workload of the bounded CPU sample (default: the reference's own MICCA mesh, 34,787 DOF)")
    ap.add_argument("--numbering", default="native", choices=["native", "rcm", "morton", "random"],
                    help="vertex numbering of the synthetic mesh (native = the generator's periodic grid numbering)")
    ap.add_argument("--jitter", type=float, default=0.0, help="displace interior nodes by +-JITTER*h")
    ap.add_argument("--host-inputs", action="store_true",
                    help="every step starts from HOST arrays (vertex coordinates, cells, tags, sound speed, heat-release field go over "
                         "PCIe and the sparsity pattern is rebuilt inside the timed region): the PCIe-inclusive rate DESIGN.md quotes "
                         "beside the contract's value -- never the contract's value itself")
    return ap.parse_args(argv)


def launch_ranks(n, argv):
    """``python bench.py --gpus N`` with no launcher around it: start the N ranks (one process per GPU) under
    torch.distributed.run, as the driver does for N > 1, BEFORE this process imports torch or touches a GPU; forward the
    one JSON line of rank 0 and the exit status."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = os.environ.copy()
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    lines = [ln for ln in proc.stdout.decode(errors="replace").splitlines() if ln.startswith("{") and '"metric"' in ln]
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    rc = proc.returncode
    if rc == 0 and not lines:
        rc = 1
    return rc


def launch_check(args):
    """Test hook (tests/test_host_logic.py, no GPU needed): every rank joins a gloo group, the ranks count themselves with
    one all-reduce and rank 0 prints a JSON line -- exercises launcher, rendezvous, line forwarding and exit status."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo")
    t = torch.ones(1, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t)
    if os.environ.get("HX_BENCH_LAUNCH_CHECK") == "fail" and int(os.environ.get("RANK", "0")) == world - 1:
        return 7                                   # one rank fails: the launcher must report a non-zero status
    if int(os.environ.get("RANK", "0")) == 0:
        print(json.dumps({"metric": METRIC, "launch_check": True, "n_gpus": int(t.item()), "gpus_arg": args.gpus}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, argv))
    if os.environ.get("HX_BENCH_LAUNCH_CHECK"):
        sys.exit(launch_check(args))

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
    n_dev = torch.cuda.device_count()
    backend = os.environ.get("HX_DIST_BACKEND")
    if backend is None:
        # RCCL refuses two ranks on one device: more ranks than GPUs is a REHEARSAL of the N > 1 path on a smaller box
        # (ranks share the devices, exchanges staged through the host by gloo); said so in the record
        backend = "nccl" if n_dev >= world else "gloo"
    local_rank %= n_dev
    torch.cuda.set_device(local_rank)
    if world > 1:
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
    case = build_case(shape, args.numbering, args.jitter, resident=(world == 1), slab_only=(world > 1))   # N > 1: only the rank's part goes to HBM
    args.numbering = case["numbering"]         # (the real-mesh workloads: "file" / "rcm")
    if world > 1:
        rc = run_sharded(args, case, shape, rank, world, backend, n_dev, barrier)
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(rc)          # (a rank that failed inside a collective cannot join destroy_process_group)
    mesh = case["mesh"]
    mesh.assembler()                           # mesh arrays -> HBM (outside the timed region)
    n_dof, n_cells = mesh.num_vertices, mesh.num_cells

    import contextlib
    with contextlib.redirect_stdout(sys.stderr):          # the library logs like the reference; stdout = the JSON line
        import gc
        for _ in range(args.warmup):
            hot_path(case, args.st_type)
            gc.collect()
        # What is alive now -- the imported modules, the case -- leaves the collector's generations: the full collections inside the
        # timed region (the explicit one per step and the automatic ones) then walk the objects of the calls only.  Measured
        # (tools/step_time_drift.py): releasing a call's objects 42-77 ms -> 0.7 ms, 1 M DOF 0.625 -> 0.58 s per call, 5 M DOF
        # 2.41 -> 2.36 s.  A long-running driver of the library does the same after its set-up (gc.freeze(), Python >= 3.7).
        gc.freeze()
        from helmholtz_x import _hx
        import ctypes as C
        # in-situ HIP-event timing of the roofline kernel (operator product on the fine operator) over the timed region
        _hx.check(_hx.lib.hx_profile_sell_spmv(1, n_dof if n_dof >= 50_000 else 2 ** 31 - 1))
        _hx.check(_hx.lib.hx_pool_stats(None, None, None, None, None, 1))          # device-buffer pool: count the timed region
        barrier()
        t0 = time.perf_counter()
        pairs = 0
        hist = []
        step_marks = [t0]
        E = matrices = None
        for _ in range(args.steps):
            hist = []
            if E is not None:                     # release the previous call's solver objects first, as a driver that
                E.destroy()                       # follows the reference does (eigensolvers.py:161 E.destroy())
            E = matrices = None
            gc.collect()
            if args.host_inputs:                  # drop every device copy of the inputs: this step uploads them again
                mesh._assembler = None
                case["c"].release()
                case.pop("h", None)
            E, matrices = hot_path(case, args.st_type, history=hist)
            pairs += min(E.getConverged(), 4)
            step_marks.append(time.perf_counter())      # (a call ends with its eigenvalues on the host: the queue is empty)
        barrier()
        dt = time.perf_counter() - t0
        in_situ, in_situ_smoother = read_in_situ()
        pr, ph, pd, pb = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int64(0)
        ps = C.c_double(0.0)
        _hx.check(_hx.lib.hx_pool_stats(C.byref(pr), C.byref(ph), C.byref(pd), C.byref(ps), C.byref(pb), 0))
        pool = {"requests_per_step": round(pr.value / max(args.steps, 1), 1), "served_from_cache_per_step": round(ph.value / max(args.steps, 1), 1),
                "driver_calls_per_step": round(pd.value / max(args.steps, 1), 1), "driver_ms_per_step": round(1e3 * ps.value / max(args.steps, 1), 2),
                "cached_gib": round(pb.value / 2 ** 30, 2)}
        free_b, total_b = torch.cuda.mem_get_info()            # device-wide: torch's pool + the library's own hipMallocs
        hbm_used_gb = round((total_b - free_b) / 2 ** 30, 1)

    omegas = [E.getEigenvalue(i) for i in range(min(E.getConverged(), 4))]
    inner = next((v for v in (E.inner_cache or {}).values() if hasattr(v, "stats")), None)
    with contextlib.redirect_stdout(sys.stderr):
        roof = spmv_roofline(shifted_operator(matrices, case["P"].target), in_situ=in_situ, pmc=args.numbering)
        roof["dominant_by_time"] = smoother_roofline(inner, n_dof, in_situ_smoother, args.numbering)
    out = {
        "metric": METRIC,
        "value": round(pairs / dt, 5), "unit": "eigenpairs/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / max(args.steps, 1), 2),
        "ms_each_step": [round(1e3 * (b - a), 1) for a, b in zip(step_marks[:-1], step_marks[1:])],
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "c128", "data": "synthetic",
        "config": {"workload": args.workload, "grid_nr_ntheta_nz": list(shape), "dofs": n_dof, "tets": n_cells,
                   "numbering": args.numbering, "jitter": args.jitter,
                   "inputs": ("host arrays uploaded inside every step (PCIe-inclusive: NOT the contract's value)" if args.host_inputs
                              else "resident in HBM before the timed region"),
                   "nev": 4, "fpi_tol": 1e-3, "fpi_iterations": len(hist) - 1, "flames": 16,
                   "inner_solver": getattr(inner, "name", None),
                   "inner_stats": inner.stats() if inner is not None else None,
                   "krylov": {k: v for k, v in (E.state or {}).items() if not k.startswith(("start", "_"))},
                   "parallelism": "1 gpu",
                   "hbm_in_use_after_step_gib": hbm_used_gb, "device_buffer_pool": pool,
                   "omega": [[round(w.real, 6), round(w.imag, 6)] for w in omegas]},
        "roofline": roof,
    }
    if not args.no_cpu_baseline:
        with contextlib.redirect_stdout(sys.stderr):
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample)
    emit(out)


def read_in_situ():
    import ctypes as C
    from helmholtz_x import _hx
    n_launch, tot_ms = C.c_int64(0), C.c_double(0.0)
    _hx.check(_hx.lib.hx_profile_sell_spmv_read(C.byref(n_launch), C.byref(tot_ms)))
    n_sm, sm_ms = C.c_int64(0), C.c_double(0.0)
    _hx.check(_hx.lib.hx_profile_sell_read(1, C.byref(n_sm), C.byref(sm_ms)))
    skipped = (int(_hx.lib.hx_profile_sell_skipped(0)), int(_hx.lib.hx_profile_sell_skipped(1)))
    _hx.check(_hx.lib.hx_profile_sell_spmv(0, 0))
    return (n_launch.value, tot_ms.value, skipped[0]), (n_sm.value, sm_ms.value, skipped[1])


def run_sharded(args, case, shape, rank, world, backend, n_dev, barrier):
    """N > 1: K timed steps of ONE row-sharded solve over all ranks.  Returns the process exit status: 0 only when the
    collective leg finished and its record was printed; a stall (watchdog) or an exception prints the record with the
    error and returns / exits non-zero."""
    import contextlib
    import gc
    import threading
    import torch
    import torch.distributed as dist
    from helmholtz_x import _hx
    budget = float(os.environ.get("HX_BENCH_SHARDED_TIMEOUT", "900"))
    if backend == "nccl":
        # on real GPUs the exchanges of the solves are the library's own RCCL calls or the run is void: no silent fall-back to
        # host-driven torch.distributed (a communicator that cannot be created ends the run with the reason in the record)
        os.environ.setdefault("HX_NATIVE_COMM", "1")
    out = {"metric": METRIC, "value": None, "unit": "eigenpairs/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "c128", "data": "synthetic",
           "config": {"workload": args.workload, "grid_nr_ntheta_nz": list(shape), "numbering": args.numbering,
                      "jitter": args.jitter, "parallelism": "row-sharded x%d" % world, "backend": backend,
                      "devices_visible": n_dev,
                      "host_setup": ("every rank generates its own theta-slab (+ one plane either side) from the grid formulas: host work and "
                                     "memory per rank shrink like 1 / N (config.partition.per_rank: host_setup_s, host_rss_gib)"
                                     if case.get("generator") is not None else
                                     "every rank builds the GLOBAL host mesh (build_case) before it cuts its part (mesh read from a file / "
                                     "renumbered: no distributing reader in the hot path)"),
                      "rehearsal": None if n_dev >= world else
                      "%d ranks share %d GPU(s): gloo stages every exchange through the host" % (world, n_dev)}}

    def fire():
        if rank == 0:
            out["error"] = "sharded solve stalled: no result after %.0f s" % budget
            emit(out)
        sys.stderr.write("[bench] rank %d: watchdog fired after %.0f s\n" % (rank, budget))
        sys.stderr.flush()
        os._exit(3)

    dog = threading.Timer(budget, fire)
    dog.daemon = True
    dog.start()
    try:
        with contextlib.redirect_stdout(sys.stderr):
            for _ in range(args.warmup):
                E, _m = hot_path(case, args.st_type or "amg", sharded=True)
                E.destroy()
                E = _m = None
                gc.collect()
            gc.freeze()                                   # (as in the single-GPU leg: modules and the case leave the collector)
            from helmholtz_x import sharded as shd
            sh = shd.distributed_case(case)["shard"]      # partition + this rank's mesh part in HBM (outside the timed region)
            n_loc = sh.n_own
            _hx.check(_hx.lib.hx_profile_sell_spmv(1, n_loc if n_loc >= 20_000 else 2 ** 31 - 1))
            sh.reset_counters()
            barrier()
            t0 = time.perf_counter()
            pairs, hist, E, matrices = 0, [], None, None
            for _ in range(args.steps):
                hist = []
                if E is not None:
                    E.destroy()
                E = matrices = None
                gc.collect()
                E, matrices = hot_path(case, args.st_type or "amg", history=hist, sharded=True)
                pairs += min(E.getConverged(), 4)
            barrier()
            dt = time.perf_counter() - t0
            in_situ, in_situ_smoother = read_in_situ()
            free_b, total_b = torch.cuda.mem_get_info()
        red_dev = "cuda" if backend == "nccl" else "cpu"
        tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        mem = torch.tensor([(total_b - free_b) / 2 ** 30], dtype=torch.float64, device=red_dev)
        dist.all_reduce(mem, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        sh = case["shard"]
        inner = next((v for v in (E.inner_cache or {}).values() if hasattr(v, "stats")), None)
        omegas = [E.getEigenvalue(i) for i in range(min(E.getConverged(), 4))]
        if rank == 0:
            with contextlib.redirect_stdout(sys.stderr):
                roof = spmv_roofline(case["K_local"], in_situ=in_situ, reps=20) if case.get("K_local") is not None else None
                if roof is not None:
                    roof["scope"] = "rank 0's rows of the shifted operator over [owned | ghost] columns"
                    roof["dominant_by_time"] = smoother_roofline(inner, sh.n_own, in_situ_smoother)
            out.update({"value": round(pairs / dt, 5), "ms_per_step": round(1e3 * dt / max(args.steps, 1), 2),
                        "exchanges_issued_by": sh.counters()["issued_by"]})
            if backend == "nccl" and sh.comm is None:
                raise RuntimeError("nccl backend without the library's own communicator: the exchanges went through torch.distributed")
            out["config"].update({
                "dofs": sh.n_global, "nev": 4, "fpi_tol": 1e-3, "fpi_iterations": len(hist) - 1, "flames": 16,
                "partition": "%s: %d rows + %d ghost rows on rank 0" % (
                    "contiguous row ranges of the RCM numbering" if shape[0] == "real" else "theta-slabs", sh.n_own,
                    sh.n_ext - sh.n_own),
                "setup": case.get("setup_mode"),
                "partition": case.get("partition"),
                "inner_solver": getattr(inner, "name", None),
                "inner_stats": inner.stats() if inner is not None else None,
                "krylov": {k: v for k, v in (E.state or {}).items() if not k.startswith(("start", "_"))},
                "exchanges_per_step": {k: (round(v / max(args.steps, 1), 1) if isinstance(v, (int, float)) else v)
                                       for k, v in sh.counters().items()},
                "hbm_in_use_after_step_gib_max_rank": round(float(mem.item()), 1),
                "omega": [[round(w.real, 6), round(w.imag, 6)] for w in omegas]})
            out["roofline"] = roof
            emit(out)
        dog.cancel()
        try:
            dist.barrier()
        except Exception:
            pass
        return 0
    except BaseException as e:                                    # recorded AND fatal: non-zero exit status
        import traceback
        traceback.print_exc(file=sys.stderr)
        if rank == 0:
            out["error"] = repr(e)[:800]
            emit(out)
        return 1


if __name__ == "__main__":
    main()
