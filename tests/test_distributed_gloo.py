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
        # ShardContext: the same rows cut out of the GLOBAL operator (integer work in torch; CPU tensors here), the
        # working-precision (complex64) halo and the complex all-reduce used by the row-sharded solve
        from helmholtz_x.petsc import CSR
        from helmholtz_x.sharded import ShardContext
        ctx = ShardContext(n, align=align)
        assert (ctx.r0, ctx.r1) == (r0, r1)
        Ag = A.tocsr()
        Ag.sort_indices()
        G = CSR(Ag.shape, torch.from_numpy(Ag.indptr.astype(np.int32)), torch.from_numpy(Ag.indices.astype(np.int32)),
                torch.from_numpy(Ag.data.astype(np.complex128)))
        L = ctx.localize(G)
        assert L.shape == (n_own, ctx.n_ext) and np.array_equal(ctx.ghost.numpy(), np.unique(Ag[r0:r1].indices[
            (Ag[r0:r1].indices < r0) | (Ag[r0:r1].indices >= r1)]))
        import scipy.sparse as sp
        Ls = sp.csr_matrix((L.val.numpy(), L.col.numpy(), L.rowptr.numpy()), shape=L.shape)
        xe = torch.zeros(ctx.n_ext, dtype=torch.complex64)
        xe[:n_own] = torch.from_numpy(xg[r0:r1]).to(torch.complex64)
        ctx.halo(xe)
        assert np.array_equal(xe[n_own:].numpy(), xg[ctx.ghost.numpy()].astype(np.complex64))
        xe128 = torch.zeros(ctx.n_ext, dtype=torch.complex128)
        xe128[:n_own] = torch.from_numpy(xg[r0:r1])
        ctx.halo(xe128)
        ref = (A @ xg)[r0:r1]
        assert np.max(np.abs(Ls @ xe128.numpy() - ref)) <= 1e-12 * np.max(np.abs(ref))
        t = torch.tensor([complex(rank + 1, -rank)], dtype=torch.complex128)
        ctx.all_reduce_(t)
        assert t[0].item() == complex(sum(range(1, world + 1)), -sum(range(world)))
        out[rank] = (n_own, len(ghosts), sorted(plan.send_to), sorted(plan.recv_from))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case,world", [("rectangle", 2), ("annulus", 2), ("annulus", 4)])
def test_sharded_operator_matches_global(case, world):
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, case, out), nprocs=world, join=True)
    assert len(out) == world
    assert sum(v[0] for v in out.values()) > 0
    for rank, (n_own, n_ghost, send, recv) in out.items():
        # theta-slabs of the periodic annulus (and the two halves of the box) talk to their two ring neighbours only
        ring = sorted({(rank - 1) % world, (rank + 1) % world})
        assert n_ghost > 0 and send == recv == ring


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


def _gpu_fpi_worker(rank, world, port, out, krylov):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "helmholtz-x_amd"), os.path.join(root, "tests")]
    os.environ["HX_QUIET"] = "1"
    os.environ["HX_WOODBURY_WARM"] = "0"      # translated initial guesses exist on one GPU only: equal footing for the iteration counts
    if krylov == "cocg":            # "auto" keeps GMRES below 1e5 rows: force the deflated COCG path (hx_cocg_sharded)
        os.environ["HX_AMG_KRYLOV"] = "cocg"
        os.environ["HX_AMG_RECYCLE"] = "6"
        os.environ["HX_SELL_MIN_ROWS"] = "0"    # SELL-64 (stencil slices included) on the rectangular local operator
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        import contextlib
        import io
        import bench
        case = bench.build_case((10, 192, 11))                     # 25,344 DOF, 16 flames, Robin outlet -> PEP
        res = {}
        for mode in ("single", "sharded"):
            hist = []
            with contextlib.redirect_stdout(io.StringIO()):
                E, _ = bench.hot_path(case, st_type="amg", nev=4, tol=1e-3, history=hist, sharded=(mode == "sharded"))
            inner = [v for v in E.inner_cache.values() if hasattr(v, "stats")][0]
            res[mode] = dict(eigs=[E.getEigenvalue(i) for i in range(4)], hist=list(hist), stats=inner.stats(),
                             ops=E.state.get("krylov_ops"))
        sh = case["shard"]
        res["exchanges"] = (sh.plan.n_exchanges, sh.n_allreduce, sh.n_own, sh.n_ext)
        out[rank] = res
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("krylov", ["auto", "cocg"])
def test_row_sharded_fixed_point_matches_single_gpu_two_ranks(krylov):
    """The whole fixed-point solve, row-sharded over two ranks (sharing the one GPU; gloo stages the exchanges through
    the host): sharded level-0 V-cycle + replicated coarse levels, hx_gmres_sharded / hx_cocg_sharded, all-reduced CGS2
    panels and Woodbury dots.  Eigenvalues and the fixed-point trajectory must match the single-GPU solve of the same problem
    to 1e-9 (same algorithm, same hierarchy; only the summation order of the inner products differs)."""
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_gpu_fpi_worker, args=(world, port, out, krylov), nprocs=world, join=True)
    assert len(out) == world
    for rank in range(world):
        r = out[rank]
        single, sharded = r["single"], r["sharded"]
        assert len(single["hist"]) == len(sharded["hist"])
        for a, b in zip(single["hist"], sharded["hist"]):
            assert abs(a - b) / abs(a) < 1e-9
        for a, b in zip(single["eigs"], sharded["eigs"]):      # (omega = 0 is among the four on this coarse mesh)
            assert abs(a - b) < 1e-9 * max(abs(a), 3000.0), (a, b)
        # same preconditioner => (nearly) the same number of inner iterations and Krylov steps
        assert abs(single["stats"]["gmres_iters"] - sharded["stats"]["gmres_iters"]) <= 0.05 * single["stats"]["gmres_iters"]
        assert abs(single["ops"] - sharded["ops"]) <= 0.1 * single["ops"]
        n_halo, n_allreduce, n_own, n_ext = r["exchanges"]
        assert n_halo > 0 and n_allreduce > 0 and n_ext > n_own
        assert sharded["stats"]["krylov"] == ("cocg" if krylov == "cocg" else "fgmres")
        assert sharded["stats"]["cocg_fallbacks"] == 0 and (krylov != "cocg" or sharded["stats"]["cocg_solves"] > 50)
        if krylov == "cocg":
            assert sharded["stats"]["fine_format"].startswith("sell64") and sharded["stats"]["stencil_slices"] > 0
    # both ranks hold identical (all-reduced) results
    assert out[0]["sharded"]["eigs"] == out[1]["sharded"]["eigs"]
    print("sharded stats:", out[0]["sharded"]["stats"], "single:", out[0]["single"]["stats"], out[0]["exchanges"])
