#!/usr/bin/env python3
"""BASELINE config 5 on the ROW-SHARDED path: the Bloch-periodic 1/16 sector partitioned in z-slabs over the ranks of a
torch.distributed job (every master/slave pair rank-local), direct AND adjoint fixed-point runs, against the single-GPU run of the
same sector (rank 0 computes it first unless --no-single).

Usage (N ranks; on a one-GPU box the ranks share the device: a rehearsal, exchanges staged through the host by gloo):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29533 \
        tools/run_config5_sharded.py [nr ntheta nz] [--no-single]            (default 76 76 316 -> 1.88 M DOF)"""
import contextlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "helmholtz-x_amd"), os.path.join(ROOT, "tests")]
os.environ.setdefault("HX_QUIET", "1")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from scipy.io import loadmat  # noqa: E402

from cases import Micca  # noqa: E402
from helmholtz_x.acoustic_matrices import AcousticMatrices  # noqa: E402
from helmholtz_x.bloch_operator import Blochifier, sector_slab_numbering  # noqa: E402
from helmholtz_x.eigensolvers import fixed_point_iteration  # noqa: E402
from helmholtz_x.fem import MeshTags  # noqa: E402
from helmholtz_x.flame_matrices import PointwiseFlameMatrix  # noqa: E402
from helmholtz_x.flame_transfer_function import stateSpace  # noqa: E402
from helmholtz_x.parameters_utils import Q_multiple  # noqa: E402
from helmholtz_x.sharded import ShardContext  # noqa: E402
from helmholtz_x.synthetic import annulus, micca_sound_speed  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
shape = tuple(int(a) for a in args[:3]) if len(args) >= 3 else (76, 76, 316)
rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
n_dev = torch.cuda.device_count()
backend = "nccl" if n_dev >= world else "gloo"
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)) % max(n_dev, 1))
dist.init_process_group(backend)
half = np.pi / 16


class Geometry:
    pass


mesh0, sub0, ft0 = annulus(*shape, n_flames=1, sector=(-half, half))
perm, planes = sector_slab_numbering(mesh0, "z")
G = Geometry()
G.mesh = mesh0.renumbered(perm)
G.facet_tags = MeshTags(G.mesh, ft0.dim, ft0.indices, ft0.values)
sub = MeshTags(G.mesh, sub0.dim, sub0.indices, sub0.values)
bcs = {11: 'Neumann', 12: 'Master', 13: 'Slave'}
mat = loadmat(os.path.join(ROOT, "tests", "golden", "micca_ftf.mat"))
FTF = stateSpace(mat['A'], mat['b'], mat['c'], mat['d'])
x_r = np.array([[Micca.r_r, 0.0, Micca.z_r]])
c = micca_sound_speed(G.mesh)
n = G.mesh.num_vertices
out = {"config": "bloch sector 1/16, b=1, row-sharded in z-slabs", "grid": list(shape), "dofs": n, "ranks": world, "backend": backend,
       "devices_visible": n_dev}
with contextlib.redirect_stdout(sys.stderr):
    single = None
    if "--no-single" not in sys.argv and rank == 0:
        t0 = time.perf_counter()
        M = AcousticMatrices(G.mesh, G.facet_tags, bcs, c)
        bloch = Blochifier(G, bcs, 16, M)
        D = PointwiseFlameMatrix(G.mesh, sub, x_r, Q_multiple(G.mesh, sub, 1), Micca.rho_amb, Micca.q_0, Micca.u_b, FTF, bloch_object=bloch)
        D.assemble_submatrices('direct')
        D.blochify('direct')
        E = fixed_point_iteration(bloch, D, Micca.target, i=0, nev=2, tol=1e-3)
        torch.cuda.synchronize()
        single = np.sqrt(E.getEigenvalue(0)) if E.getType() == "krylovschur" else E.getEigenvalue(0)    # (no Robin wall: linear problem in omega^2)
        out["single_gpu_s"] = round(time.perf_counter() - t0, 3)
        del M, bloch, D, E
    dist.barrier()
    cuts = [0] + [int(planes[np.argmin(np.abs(planes - (q * n) // world))]) for q in range(1, world)] + [n]
    ranges = [(cuts[q], cuts[q + 1]) for q in range(world)]
    t0 = time.perf_counter()
    ctx = ShardContext(n, ranges=ranges)
    local, sub_l, ft_l, (c_l,) = ctx.distribute(G.mesh, sub, G.facet_tags, [c])
    L = Geometry()
    L.mesh, L.facet_tags = local, ft_l
    M_l = AcousticMatrices(local, ft_l, bcs, c_l)
    bloch_l = Blochifier(L, bcs, 16, M_l, global_geometry=G)
    D_l = PointwiseFlameMatrix(local, sub_l, x_r, Q_multiple(local, sub_l, 1), Micca.rho_amb, Micca.q_0, Micca.u_b, FTF, bloch_object=bloch_l)
    D_l.assemble_submatrices('direct')
    D_l.blochify('direct')
    torch.cuda.synchronize()
    dist.barrier()
    out["assembly_and_bloch_reduction_s"] = round(time.perf_counter() - t0, 3)
    t0 = time.perf_counter()
    E_l = fixed_point_iteration(bloch_l, D_l, Micca.target, i=0, nev=2, tol=1e-3, shard=bloch_l.shard)
    torch.cuda.synchronize()
    dist.barrier()
    out["direct_s"] = round(time.perf_counter() - t0, 3)
    t0 = time.perf_counter()
    D_l.assemble_submatrices('adjoint')
    D_l.blochify('adjoint')
    E_a = fixed_point_iteration(bloch_l, D_l, Micca.target.conjugate(), i=0, nev=2, tol=1e-3, problem_type='adjoint', shard=bloch_l.shard)
    torch.cuda.synchronize()
    dist.barrier()
    out["adjoint_s"] = round(time.perf_counter() - t0, 3)
    w, wa = E_l.getEigenvalue(0), E_a.getEigenvalue(0)
    if E_l.getType() == "krylovschur":
        w, wa = np.sqrt(w), np.sqrt(wa)
    inner = next(v for v in E_l.inner_cache.values() if hasattr(v, "stats"))
    out.update(omega=[w.real, w.imag], omega_adjoint=[wa.real, wa.imag], adjoint_is_conjugate=abs(wa - np.conj(w)) / abs(w),
               rows_per_rank=[b - a for a, b in ranges], pairs_on_rank0=int(bloch_l.dofs_master.size),
               reduced_rows_ghosts_rank0=[bloch_l.shard.n_own, bloch_l.shard.n_ext - bloch_l.shard.n_own],
               inner={k: v for k, v in inner.stats().items() if k in ("krylov", "levels", "sharded_levels", "solves", "gmres_iters", "iters_per_solve")},
               exchanges=bloch_l.shard.counters())
    if single is not None:
        out["single_gpu_omega"] = [single.real, single.imag]
        out["rel_diff_vs_single_gpu"] = abs(w - single) / abs(single)
if rank == 0:
    print(json.dumps(out))
dist.destroy_process_group()
