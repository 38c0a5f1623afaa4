"""World-size-2 gloo tests (CPU) of the row-sharded path: partition, owner-computes local meshes, halo plan and
exchange.  The local arithmetic in these tests is the oracle's (SciPy) -- the product kernels are GPU-only -- so what
is verified here is exactly the distributed logic: that [owned rows of the locally assembled operator] x
[owned + exchanged ghost entries] reproduces the global operator product, with no assembly communication."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, case, out):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "helmholtz-x_amd"), os.path.join(root, "tests")]
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        from helmholtz_x.distributed import HaloPlan, partition_rows, shard_mesh
        from oracle import fem
        if case == "annulus":
            from helmholtz_x.synthetic import annulus, micca_sound_speed
            mesh, _, _ = annulus(3, 16, 4)
            x, cells = mesh.geometry.x, mesh.topology.cells
            c = micca_sound_speed(mesh).x.array
            align = 4 * 5
        else:
            from oracle import meshes
            m = meshes.rectangle(12, 5, 0.4, 0.1)
            x, cells = m.x, m.cells
            c = 300.0 + np.arange(cells.shape[0]) % 5
            align = 1
        n = x.shape[0]
        ranges = partition_rows(n, world, align)
        r0, r1 = ranges[rank]
        sh = shard_mesh(x, cells, r0, r1)
        n_own = sh["n_own"]
        assert n_own == r1 - r0
        # owner computes: assemble on the local mesh only, keep the owned rows
        A_loc = fem.assemble_A(sh["x"], sh["cells"], c[sh["cell_ids"]], c_is_cellwise=True)[:n_own]
        C_loc = fem.assemble_C(sh["x"], sh["cells"])[:n_own]
        ghosts = sh["local_to_global"][n_own:]
        plan = HaloPlan(ranges, rank, ghosts)
        # global reference
        A = fem.assemble_A(x, cells, c, c_is_cellwise=True)
        C = fem.assemble_C(x, cells)
        rng = np.random.default_rng(7)
        xg = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        x_ext = torch.zeros(n_own + len(ghosts), dtype=torch.complex128)
        x_ext[:n_own] = torch.from_numpy(xg[r0:r1])
        plan.exchange(x_ext, n_own)
        assert np.array_equal(x_ext[n_own:].numpy(), xg[ghosts])          # halo values are the owners' values
        for loc, glob in ((A_loc, A), (C_loc, C)):
            y = loc @ x_ext.numpy()
            ref = (glob @ xg)[r0:r1]
            assert np.max(np.abs(y - ref)) <= 1e-12 * np.max(np.abs(ref))
        # dots: one all-reduce of the local partial sums
        part = torch.tensor([np.vdot(xg[r0:r1], xg[r0:r1]).real, float(n_own)], dtype=torch.float64)
        dist.all_reduce(part)
        assert abs(part[0].item() - np.vdot(xg, xg).real) < 1e-9 and int(part[1].item()) == n
        out[rank] = (n_own, len(ghosts), sorted(plan.send_to), sorted(plan.recv_from))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["rectangle", "annulus"])
def test_sharded_operator_matches_global_world2(case):
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, case, out), nprocs=world, join=True)
    assert len(out) == world
    assert sum(v[0] for v in out.values()) > 0
    for rank, (n_own, n_ghost, send, recv) in out.items():
        assert n_ghost > 0 and send == recv == [1 - rank]      # two slabs talk to each other only


def test_partition_rows_is_contiguous_and_aligned():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "helmholtz-x_amd"))
    from helmholtz_x.distributed import partition_rows
    for n, w, al in ((10, 1, 1), (1000, 8, 1), (57 * 1792 * 49, 8, 57 * 49), (5, 8, 1)):
        r = partition_rows(n, w, al)
        assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
        assert all((a % al == 0) or a == n for a, _ in r)


def _gpu_worker(rank, world, port, out):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "helmholtz-x_amd"), os.path.join(root, "tests")]
    os.environ["HX_QUIET"] = "1"
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        from helmholtz_x.distributed import HaloPlan, ShardedOperator, partition_rows, shard_mesh, sharded_panel_dot
        from helmholtz_x.fem import Function, FunctionSpace, Mesh
        from helmholtz_x.petsc import CSR, to_device
        from helmholtz_x.synthetic import annulus, micca_sound_speed
        from oracle import fem
        mesh, _, _ = annulus(4, 32, 6)
        x, cells = mesh.geometry.x, mesh.topology.cells
        c = micca_sound_speed(mesh).x.array
        n = x.shape[0]
        ranges = partition_rows(n, world, align=5 * 7)
        r0, r1 = ranges[rank]
        sh = shard_mesh(x, cells, r0, r1)
        n_own = sh["n_own"]
        local = Mesh(sh["x"], sh["cells"], facets=np.zeros((0, 3), np.int32), facet_cells=np.zeros(0, np.int32))
        cf = Function(FunctionSpace(local, "DG", 0), "soundspeed")
        cf.x.array[:] = c[sh["cell_ids"]]
        A, C = local.assembler().stiffness_mass(cf)           # HIP assembly of the local mesh (owner computes)
        full = A.csr_part()
        nnz_own = int(full.rowptr[n_own].item())
        A_own = CSR((n_own, full.shape[1]), full.rowptr[:n_own + 1], full.col[:nnz_own], full.val[:nnz_own])
        plan = HaloPlan(ranges, rank, sh["local_to_global"][n_own:])
        op = ShardedOperator(A_own, n_own, plan)
        rng = np.random.default_rng(3)
        xg = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        x_ext = torch.zeros(full.shape[1], dtype=torch.complex128, device="cuda")
        x_ext[:n_own] = to_device(xg[r0:r1])
        y = torch.empty(n_own, dtype=torch.complex128, device="cuda")
        op.mult(x_ext, y)
        ref = (fem.assemble_A(x, cells, c, c_is_cellwise=True) @ xg)[r0:r1]
        err = np.max(np.abs(y.cpu().numpy() - ref)) / np.max(np.abs(ref))
        V = to_device(np.stack([xg[r0:r1], 1j * xg[r0:r1] + 1.0]))
        h = sharded_panel_dot(V, 2, x_ext[:n_own].clone())
        href = np.array([np.vdot(xg, xg), np.vdot(1j * xg + 1.0, xg), np.vdot(xg, xg)])
        out[rank] = (float(err), float(np.max(np.abs(h - href)) / np.abs(href[0])))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_spmv_and_dots_two_ranks_one_gpu():
    """Two ranks sharing the one GPU (gloo stages the halo through the host): HIP owner-computes assembly of each
    rank's slab, halo exchange, local hx_csr_spmv_z and the all-reduced CGS panel reproduce the global results."""
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_gpu_worker, args=(world, port, out), nprocs=world, join=True)
    assert len(out) == world
    for err_spmv, err_dot in out.values():
        assert err_spmv < 1e-12 and err_dot < 1e-12
