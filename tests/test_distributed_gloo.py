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
        # the replicated operator of the direct route (slepc.ReplicatedLUSolver): every rank rebuilds the GLOBAL matrix from the
        # ranks' row blocks -- entry for entry the matrix the rows were cut from, and recognised as structurally symmetric
        from helmholtz_x.slepc import replicated_operator
        R = replicated_operator(L, ctx)
        assert R.shape == Ag.shape and R.symmetric
        assert np.array_equal(R.rowptr.numpy(), Ag.indptr) and np.array_equal(R.col.numpy(), Ag.indices)
        assert np.array_equal(R.val.numpy(), Ag.data.astype(np.complex128))
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


def _gpu_fpi_worker(rank, world, port, out, krylov, transport="gloo"):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "helmholtz-x_amd"), os.path.join(root, "tests")]
    os.environ["HX_QUIET"] = "1"
    if transport.startswith("fake_rccl"):    # exchanges issued BY THE LIBRARY (hx_comm) through the host-staged librccl stand-in
        os.environ["HX_NATIVE_COMM"] = "1"
        os.environ["HX_RCCL_LIB"] = os.path.join(root, "tests", "fake_rccl", "libfakerccl.so")
    # ghost layers per sharded level: "_deep" three (the default: 2 (1) exchanges per level and cycle), "_deep2" two (3 (2)); otherwise
    # the classic schedule, one layer, 5 (3) exchanges
    os.environ["HX_AMG_DEEP_HALO"] = "1" if "_deep" in transport else "0"
    os.environ["HX_AMG_DEEP_LAYERS"] = "2" if transport.endswith("_deep2") else "3"
    if krylov == "fgmres":          # the flexible GMRES path (hx_gmres_sharded): "auto" picks COCG for this symmetric operator
        os.environ["HX_AMG_KRYLOV"] = "fgmres"
    if krylov == "cocg":            # the deflated COCG path (hx_cocg_sharded) with recycling, SELL-64 and two sharded levels
        os.environ["HX_AMG_KRYLOV"] = "cocg"
        os.environ["HX_AMG_RECYCLE"] = "6"
        os.environ["HX_SELL_MIN_ROWS"] = "0"    # SELL-64 (stencil slices included) on the rectangular local operator
        os.environ["HX_AMG_REPLICATE_MAX_ROWS"] = "2000"     # levels 0 AND 1 stay row-sharded (25 k -> 4.3 k -> replicated tail)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        import contextlib
        import io
        import bench
        from helmholtz_x.sharded import ShardContext
        case = bench.build_case((10, 192, 11))                     # 25,344 DOF, 16 flames, Robin outlet -> PEP
        res = {}
        for mode in ("single", "distributed", "global_operators"):
            hist = []
            with contextlib.redirect_stdout(io.StringIO()):
                if mode == "global_operators":
                    # the operators assembled on the whole mesh by every rank, then cut to the rank's rows by the solver
                    from helmholtz_x.acoustic_matrices import AcousticMatrices
                    from helmholtz_x.eigensolvers import fixed_point_iteration
                    from helmholtz_x.flame_matrices import PointwiseFlameMatrix
                    from helmholtz_x.parameters_utils import Q_multiple
                    P, mesh = case["P"], case["mesh"]
                    M = AcousticMatrices(mesh, case["facet_tags"], {11: {'Robin': P.R_outlet}}, case["c"])
                    D = PointwiseFlameMatrix(mesh, case["subdomains"], P.x_r(), Q_multiple(mesh, case["subdomains"], P.N_sector),
                                             P.rho_amb, P.q_0, P.u_b, case["ftf"])
                    D.assemble_submatrices('direct')
                    ctx = ShardContext(mesh.num_vertices, align=11 * 12)
                    E = fixed_point_iteration(M, D, P.target, i=0, nev=4, tol=1e-3, st_type="amg", history=hist, shard=ctx)
                else:
                    E, _ = bench.hot_path(case, st_type="amg", nev=4, tol=1e-3, history=hist, sharded=(mode == "distributed"))
            inner = [v for v in E.inner_cache.values() if hasattr(v, "stats")][0]
            res[mode] = dict(eigs=[E.getEigenvalue(i) for i in range(4)], hist=list(hist), stats=inner.stats(),
                             ops=E.state.get("krylov_ops"))
        sh = case["shard"]
        res["exchanges"] = (sh.counters(), sh.n_own, sh.n_ext)
        res["local_mesh"] = (case["dist"]["mesh"].num_vertices, case["dist"]["mesh"].num_cells,
                             case["mesh"].num_vertices, case["mesh"].num_cells)
        out[rank] = res
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("krylov,transport", [("fgmres", "gloo"), ("cocg", "gloo"), ("cocg", "fake_rccl"), ("cocg", "gloo_deep"),
                                              ("cocg", "fake_rccl_deep"), ("cocg", "fake_rccl_deep2"), ("fgmres", "fake_rccl_deep")])
def test_row_sharded_fixed_point_matches_single_gpu_two_ranks(krylov, transport):
    """The whole hot path partitioned over two ranks (sharing the one GPU; gloo stages the exchanges through the host): every
    rank assembles ITS rows on its part of the mesh (owner computes), builds its slab of the hierarchy (uncoupled aggregation,
    Galerkin products from owned rows + sparse-row halos; with krylov = "cocg" the first coarse level stays sharded too),
    and the ranks solve collectively (hx_gmres_sharded / hx_cocg_sharded, all-reduced CGS2 panels and Woodbury dots).
    Eigenvalues and the fixed-point trajectory must match the single-GPU solve of the same problem to 1e-9 -- the
    preconditioner differs near the cuts (aggregates do not cross them), the converged answers may not.  A third run hands
    the solver operators assembled on the WHOLE mesh by every rank (the round-2 interface) and must agree as well.
    transport "fake_rccl": the exchanges are issued by the library itself (hx_comm: the production path under RCCL) through
    tests/fake_rccl, a host-staged stand-in for librccl (RCCL refuses two ranks on one GPU): zero Python frames per
    exchange, same plans and call sequence as on two GPUs."""
    deep = "_deep" in transport
    layers = 0 if not deep else (2 if transport.endswith("_deep2") else 3)
    if transport.startswith("fake_rccl"):          # the stand-in is built by __graft_entry__.build(); build it here if it did not travel
        import subprocess
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        if not os.path.exists(os.path.join(root, "tests", "fake_rccl", "libfakerccl.so")):
            subprocess.check_call(["make", "-C", os.path.join(root, "tests", "fake_rccl")])
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_gpu_fpi_worker, args=(world, port, out, krylov, transport), nprocs=world, join=True)
    assert len(out) == world
    for rank in range(world):
        r = out[rank]
        single = r["single"]
        for mode in ("distributed", "global_operators"):
            sharded = r[mode]
            assert len(single["hist"]) == len(sharded["hist"])
            for a, b in zip(single["hist"], sharded["hist"]):
                assert abs(a - b) / abs(a) < 1e-9
            for a, b in zip(single["eigs"], sharded["eigs"]):      # (omega = 0 is among the four on this coarse mesh)
                assert abs(a - b) < 1e-9 * max(abs(a), 3000.0), (a, b)
            # uncoupled aggregation changes the hierarchy only along the two cut planes: nearly the same iteration counts
            assert abs(single["stats"]["gmres_iters"] - sharded["stats"]["gmres_iters"]) <= 0.15 * single["stats"]["gmres_iters"], \
                (single["stats"], sharded["stats"])
            assert abs(single["ops"] - sharded["ops"]) <= 0.1 * single["ops"]
            assert sharded["stats"]["krylov"] == ("cocg" if krylov == "cocg" else "fgmres")
            assert sharded["stats"]["cocg_fallbacks"] == 0 and (krylov != "cocg" or sharded["stats"]["cocg_solves"] > 50)
            assert len(sharded["stats"]["sharded_levels"]) == (2 if krylov == "cocg" else 1)
            if krylov == "cocg":
                assert sharded["stats"]["fine_format"].startswith("sell64") and sharded["stats"]["stencil_slices"] > 0
        counters, n_own, n_ext = r["exchanges"]
        assert counters["halo_exchanges"] > 0 and counters["all_reduces"] > 0 and n_ext > n_own
        assert ("libhx" in counters["issued_by"]) == transport.startswith("fake_rccl"), counters
        assert len(r["distributed"]["stats"]["deep_halo_levels"]) == (len(r["distributed"]["stats"]["sharded_levels"]) if deep else 0)
        if transport.startswith("fake_rccl"):      # most halos run split: interior slices of the product while the exchange is in flight
            # (deep-halo mode: the fine level's multi-layer exchange of b is split as well -- x1 on the owned block and the interior
            # slices of the first smoothing step run while it travels; the second level has no slice lists and its exchange blocks)
            # (the fgmres case keeps its small levels on the CSR kernels: no slice lists, nothing to split)
            if krylov == "cocg":
                assert counters["halos_overlapped_with_interior_slices"] > (0.25 if deep else 0.5) * counters["halo_exchanges"], counters
            # every operation of the communicator -- blocking and split halos, all-reduces -- was issued on ONE stream (RCCL
            # orders the operations of a communicator only within a stream; the asynchronous stand-in fails a run that mixes)
            assert counters["communicator_stream_switches"] == 0, counters
            # single-reduction COCG: ONE all-reduce of k + 3 scalars per inner iteration (+ one inside the V-cycle: the
            # restriction onto the first replicated level) and three per solve (|b|, the Galerkin start, the closing true
            # residual); round 3 needed 3.9 per inner iteration
            st = r["distributed"]["stats"]
            if krylov == "cocg":
                assert st["cocg_allreduces"] <= 2 * st["cocg_iters"] + 2 * st["cocg_solves"] + 3 * st["cocg_solves"], st
                assert st["cocg_allreduces"] / st["cocg_iters"] < 2.7, st
                # halos per inner iteration with two sharded levels: classic 9 (fine level 5, second level 3, operator product 1),
                # deep-halo schedule with two layers 6 (3 + 2 + 1), with three layers 4 (2 + 1 + 1); a few more per solve (start,
                # closing residual)
                per_it = st["cocg_halos"] / st["cocg_iters"]
                assert {0: 8.5 < per_it < 10.5, 2: 5.5 < per_it < 6.9, 3: per_it < 4.9}[layers], (per_it, st)
                assert st["deep_halo_layers"] == layers
        nv_loc, nc_loc, nv, nc = r["local_mesh"]
        assert nv_loc < 0.6 * nv and nc_loc < 0.6 * nc            # a rank holds its half of the mesh + one ghost layer
    # both ranks hold identical (all-reduced) results
    assert out[0]["distributed"]["eigs"] == out[1]["distributed"]["eigs"]
    print("sharded stats:", out[0]["distributed"]["stats"], "single:", out[0]["single"]["stats"], out[0]["exchanges"])


def _dist_worker(rank, world, port, out):
    """distribute_mesh + sparse-row halo + cyclic shift + native plan arrays on CPU tensors (gloo)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "helmholtz-x_amd"), os.path.join(root, "tests")]
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        import scipy.sparse as sp
        from helmholtz_x.distributed import HaloPlan, distribute_mesh, partition_rows
        from helmholtz_x.sharded import ShardContext
        from helmholtz_x.synthetic import annulus, micca_sound_speed
        from oracle import fem
        mesh, subdomains, facet_tags = annulus(3, 16, 4)
        c = micca_sound_speed(mesh)
        x, cells = mesh.geometry.x, mesh.topology.cells
        n = x.shape[0]
        ranges = partition_rows(n, world, 4 * 5)
        r0, r1 = ranges[rank]
        local, sub_l, ft_l, (c_l,) = distribute_mesh(mesh, ranges, rank, subdomains, facet_tags, [c])
        part = local.part
        n_own, l2g = part["n_own"], part["local_to_global"]
        assert n_own == r1 - r0 and np.array_equal(l2g[:n_own], np.arange(r0, r1)) and np.all(np.diff(l2g[n_own:]) > 0)
        # fields and tags follow their entities
        assert np.array_equal(c_l.x.array, c.x.array[part["cell_ids"]])
        gtag = np.full(cells.shape[0], -1, np.int64)
        gtag[subdomains.indices] = subdomains.values
        ltag = np.full(local.num_cells, -1, np.int64)
        ltag[sub_l.indices] = sub_l.values
        assert np.array_equal(ltag, gtag[part["cell_ids"]])
        assert np.array_equal(np.sort(l2g[local.facets], axis=1), np.sort(mesh.facets[part["facet_ids"]], axis=1))
        assert np.array_equal(part["cell_ids"][local.facet_cells], mesh.facet_cells[part["facet_ids"]])
        gft = np.full(mesh.facets.shape[0], -1, np.int64)
        gft[facet_tags.indices] = facet_tags.values
        lft = np.full(local.facets.shape[0], -1, np.int64)
        lft[ft_l.indices] = ft_l.values
        assert np.array_equal(lft, gft[part["facet_ids"]])
        # every cell / tagged facet is owned by exactly one rank
        owned = torch.tensor([float(part["cell_owned"].sum()), float(part["facet_owned"].sum())], dtype=torch.float64)
        dist.all_reduce(owned)
        assert int(owned[0]) == cells.shape[0] and int(owned[1]) == mesh.facets.shape[0]
        # owner computes: Robin matrix rows from the local facets = the global rows
        sel_g, sel_l = facet_tags.find(11), ft_l.find(11)
        Bg = fem.assemble_B(x, n, [(mesh.facets[sel_g], 2.0 + 1j, c.x.array, mesh.facet_cells[sel_g])]).tocsr()
        Bl = fem.assemble_B(local.geometry.x, local.num_vertices,
                            [(local.facets[sel_l], 2.0 + 1j, c_l.x.array, local.facet_cells[sel_l])]).tocsr()[:n_own]
        P = sp.csr_matrix((np.ones(l2g.size), (np.arange(l2g.size), l2g)), shape=(l2g.size, n))
        assert abs(Bl @ P - Bg[r0:r1]).max() < 1e-14 * max(abs(Bg).max(), 1.0)
        # mesh.comm: sums over ranks like MPI.COMM_WORLD.allreduce
        assert local.comm.allreduce(1.5) == 1.5 * world
        assert np.array_equal(local.comm.allreduce(np.array([float(rank), 1.0]), op="max"), [world - 1.0, 1.0])
        # plans: level 0 from the mesh ghosts; native arrays describe the same exchange
        ctx = ShardContext(n, ranges=ranges)
        ctx.set_ghosts(l2g[n_own:])
        sp_, sptr, sidx, rp_, roff, rcnt = ctx.plan.native_arrays()
        assert int(rcnt.sum()) == l2g.size - n_own and np.all(roff >= n_own) and np.all(sidx < n_own)
        assert sorted(sp_.tolist()) == sorted(rp_.tolist())
        # sparse-row halo: rows of a global matrix owned elsewhere arrive in ghost order
        rng = np.random.default_rng(5)
        G = sp.random(n, 40, density=0.2, random_state=rng, format="csr") + sp.eye(n, 40, format="csr")
        G = sp.csr_matrix(G, dtype=np.complex128)
        G.data = G.data * (1 + 0.5j)
        G.sort_indices()
        own = G[r0:r1]
        rp, gc, gv = ctx.plan.exchange_rows(torch.from_numpy(own.indptr.astype(np.int32)),
                                            torch.from_numpy(own.indices.astype(np.int32)), torch.from_numpy(own.data))
        ref = G[l2g[n_own:]]
        assert np.array_equal(rp.numpy(), ref.indptr) and np.array_equal(gc.numpy(), ref.indices)
        assert np.array_equal(gv.numpy(), ref.data)
        # int32 / float64 halos of the set-up
        ai = torch.zeros(ctx.n_ext, dtype=torch.int32)
        ai[:n_own] = torch.arange(r0, r1, dtype=torch.int32) * 3
        ctx.halo(ai)
        assert np.array_equal(ai[n_own:].numpy(), 3 * l2g[n_own:])
        # global cyclic shift of a row-sharded vector
        vg = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        for s in (0, 7, 20, n - 13, 3 * (n // world) + 5):
            got = ctx.cyclic_shift(torch.from_numpy(vg[r0:r1].copy()), s).numpy()
            assert np.array_equal(got, np.roll(vg, s)[r0:r1]), s
        out[rank] = True
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_distributed_mesh_rows_halo_and_shift(world):
    """The building blocks of the distributed set-up, on CPU under gloo: every rank's part of the mesh (cells + ghost layer,
    tags, fields, facet / cell ownership), owner-computes boundary rows, the exchange plan in the form the library's RCCL
    layer takes (hx_comm_set_plan), the sparse-row halo of the Galerkin products and the global cyclic shift used for the
    translated Woodbury guesses."""
    port = _free_port()
    out = mp.Manager().dict()
    mp.spawn(_dist_worker, args=(world, port, out), nprocs=world, join=True)
    assert len(out) == world and all(out.values())


def _p2_dofmap_worker(rank, world, port, out, golden_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "helmholtz-x_amd")]
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        from helmholtz_x.distributed import distribute_mesh, partition_rows
        from helmholtz_x.fem import FunctionSpace, MeshTags, locality_permutation
        from helmholtz_x.io_utils import XDMFReader
        mesh0, _, ft0 = XDMFReader(os.path.join(golden_dir, "meshes", "rijke3d")).getAll()
        mesh = mesh0.renumbered(locality_permutation(mesh0.geometry.x))
        ft = MeshTags(mesh, ft0.dim, ft0.indices, ft0.values)
        ranges = partition_rows(mesh.num_vertices, world)
        local, _, ft_l, _ = distribute_mesh(mesh, ranges, rank, None, ft, [])
        V = FunctionSpace(local, "Lagrange", 2)
        Vg = FunctionSpace(mesh, "Lagrange", 2)
        assert FunctionSpace(local, "CG", 2).cell_dofs is V.cell_dofs            # one numbering per mesh
        n_own, nv_own = V.n_own, local.part["n_own"]
        d = V._dist
        assert V.size == local.num_vertices + V.edges.shape[0] == d["dof_l2g"].size
        assert sorted(np.unique(V.cell_dofs).tolist()) == list(range(V.size))       # every local dof sits in a local cell
        assert np.all(np.diff(d["dof_l2g"][n_own:]) > 0)                            # ghosts ascending, as the exchange plan wants
        a, b = d["dof_ranges"][rank]
        assert b - a == n_own and np.array_equal(d["dof_l2g"][:n_own], np.arange(a, b))
        assert d["dof_ranges"][0][0] == 0 and d["dof_ranges"][-1][1] == d["n_global_dofs"] == Vg.size
        # owned edges: lower global vertex owned here; their rows are complete: every cell of the global mesh that holds the edge is local
        l2g = local.part["local_to_global"]
        e_own = V.edges[(V.edge_dof >= nv_own) & (V.edge_dof < n_own)]
        assert np.all(np.minimum(l2g[e_own[:, 0]], l2g[e_own[:, 1]]) < ranges[rank][1])
        assert np.all(np.minimum(l2g[e_own[:, 0]], l2g[e_own[:, 1]]) >= ranges[rank][0])
        xg = Vg.tabulate_dof_coordinates()
        xl = V.tabulate_dof_coordinates()
        key = lambda x: np.round(x * 1e9).astype(np.int64)                          # dof coordinates identify dofs (midpoints are distinct)
        gmap = {tuple(k): i for i, k in enumerate(key(xg))}
        serial_of_local = np.array([gmap[tuple(k)] for k in key(xl)])
        cells_g = {tuple(sorted(c)) for c in Vg.cell_dofs.tolist()}
        incident = {}
        for c in Vg.cell_dofs.tolist():
            for dof in c:
                incident[dof] = incident.get(dof, 0) + 1
        local_incident = {}
        for c in V.cell_dofs.tolist():
            assert tuple(sorted(serial_of_local[c].tolist())) in cells_g
            for dof in c:
                local_incident[dof] = local_incident.get(dof, 0) + 1
        assert all(local_incident[i] == incident[serial_of_local[i]] for i in range(n_own))
        # the space's row-sharding context: one halo of the owned dof coordinates reproduces the ghost dofs' coordinates
        ctx = V.shard
        assert ctx is FunctionSpace(local, "CG", 2).shard and ctx.n_own == n_own and ctx.n_ext == V.size
        for comp in range(3):
            t = torch.zeros(V.size, dtype=torch.float64)
            t[:n_own] = torch.from_numpy(xl[:n_own, comp].copy())
            ctx.halo(t)
            assert np.array_equal(t.numpy(), xl[:, comp])
        # boundary facets: vertex and edge-midpoint dofs in the local numbering
        asm_like = type("A", (), {"V": V, "mesh": local})()
        from helmholtz_x.fem import P2Assembler
        fd = P2Assembler.facet_dofs(asm_like, local.facets[ft_l.find(2)]) if ft_l.find(2).size else np.zeros((0, 6), np.int32)
        for row, fv in zip(fd, local.facets[ft_l.find(2)]):
            X = local.geometry.x[fv]
            assert np.allclose(xl[row[:3]], X) and np.allclose(xl[row[3:]], 0.5 * (X[[0, 0, 1]] + X[[1, 2, 2]]))
        out[rank] = (n_own, V.size, tuple(sorted(serial_of_local[:n_own].tolist())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_quadratic_dofmap_of_a_distributed_mesh(world, golden_dir):
    """Degree 2 on a distributed mesh (VERDICT r4 missing 3; the reference's signatures take ``degree`` under mpirun,
    acoustic_matrices.py:25, flame_matrices.py:131,193): edges owned by the rank of their lower global vertex, global numbering
    rank after rank (owned vertices, then owned edges), local numbering [owned | ghosts ascending].  On the Rijke 3D mesh cut into
    2 / 3 parts, CPU + gloo: the owned dofs of the ranks partition the serial space's dofs, every owned dof has all its cells
    locally (complete rows), and one halo through the space's own ShardContext carries the owned dof coordinates onto the ghosts."""
    port = _free_port()
    out = mp.Manager().dict()
    mp.spawn(_p2_dofmap_worker, args=(world, port, out, golden_dir), nprocs=world, join=True)
    assert len(out) == world
    owned = [set(out[r][2]) for r in range(world)]
    total = sum(out[r][0] for r in range(world))
    assert len(set().union(*owned)) == total == sum(len(o) for o in owned)          # a partition of the serial dofs


def _degree2_distributed_worker(rank, world, port, out, golden_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "helmholtz-x_amd"), os.path.join(root, "tests")]
    os.environ["HX_QUIET"] = "1"
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        import contextlib
        import io
        from scipy.io import loadmat
        from cases import Micca, RijkeFFD as F
        from helmholtz_x.acoustic_matrices import AcousticMatrices
        from helmholtz_x.eigensolvers import fixed_point_iteration
        from helmholtz_x.eigenvectors import normalize_eigenvector
        from helmholtz_x.fem import MeshTags, locality_permutation
        from helmholtz_x.flame_matrices import DistributedFlameMatrix, PointwiseFlameMatrix
        from helmholtz_x.flame_transfer_function import nTau, stateSpace
        from helmholtz_x.io_utils import XDMFReader
        from helmholtz_x.parameters_utils import Q_multiple, gaussianFunction, rho_step, temperature_step
        from helmholtz_x.sharded import ShardContext
        key = lambda x: [tuple(k) for k in np.round(np.asarray(x) * 1e9).astype(np.int64).tolist()]

        def rijke(mesh, ft, shard):
            T = temperature_step(mesh, F.x_f, F.T_in, F.T_out)
            matrices = AcousticMatrices(mesh, ft, F.bcs, T, degree=2)
            rho = rho_step(mesh, F.x_f, F.a_f, F.rho_d, F.rho_u)
            w = gaussianFunction(mesh, F.x_r, F.a_r)
            h = gaussianFunction(mesh, F.x_f, F.a_f)
            D = DistributedFlameMatrix(mesh, w, h, rho, T, F.q_0, F.u_b, nTau(F.n, F.tau), degree=2, gamma=F.gamma)
            D.assemble_submatrices('direct')
            hist = []
            E = fixed_point_iteration(matrices, D, F.target, nev=2, i=0, st_type="amg", history=hist,
                                      shard=matrices.V.shard if shard else None)
            omega, p = normalize_eigenvector(mesh, E, 0, degree=2, which='right', matrices=matrices, print_eigs=False)
            inner = [v for v in E.inner_cache.values() if hasattr(v, "stats")][0]
            return matrices.V, list(hist), complex(omega), p, inner.stats()

        res = {}
        mesh0, _, ft0 = XDMFReader(os.path.join(golden_dir, "meshes", "rijkeffd_original")).getAll()
        mesh = mesh0.renumbered(locality_permutation(mesh0.geometry.x))
        ft = MeshTags(mesh, ft0.dim, ft0.indices, ft0.values)
        ctx = ShardContext(mesh.num_vertices)
        local, _, ft_l, _ = ctx.distribute(mesh, None, ft, [])
        with contextlib.redirect_stdout(io.StringIO()):
            V, hist, omega, p, stats = rijke(local, ft_l, True)
        n_own = V.n_own
        xl = V.tabulate_dof_coordinates()
        res["rijke"] = dict(hist=hist, omega=omega, keys=key(xl[:n_own]), p_own=p.x.array[:n_own].copy(), stats=stats,
                            sizes=(local.part["n_own"], n_own, V.size), n_global=V.shard.n_global,
                            ghost_keys=key(xl[n_own:]), p_ghost=p.x.array[n_own:].copy())
        if rank == 0:                                        # the same run on the whole mesh, one GPU
            with contextlib.redirect_stdout(io.StringIO()):
                Vs, hist_s, omega_s, p_s, stats_s = rijke(mesh, ft, False)
            res["rijke_serial"] = dict(hist=hist_s, omega=omega_s, keys=key(Vs.tabulate_dof_coordinates()), p=p_s.x.array.copy(),
                                       stats=stats_s, size=Vs.size)

        # pointwise flames with degree 2 on the real MICCA mesh: the factors of D (no solve)
        def micca(mesh, sub):
            mat = loadmat(os.path.join(golden_dir, "micca_ftf.mat"))
            FTF = stateSpace(mat['A'], mat['b'], mat['c'], mat['d'])
            h = Q_multiple(mesh, sub, Micca.N_sector)
            D = PointwiseFlameMatrix(mesh, sub, Micca.x_r(), h, Micca.rho_amb, Micca.q_0, Micca.u_b, FTF, degree=2)
            D.assemble_submatrices('direct')
            xd = D.V.tabulate_dof_coordinates()
            (_, lr), = D.submatrices.lr_terms
            return [(dict(zip(key(xd[li]), lv.tolist())), dict(zip(key(xd[ri]), rv.tolist()))) for (li, lv), (ri, rv) in lr.pairs]

        m0, s0, _ = XDMFReader(os.path.join(golden_dir, "meshes", "micca_full", "mesh")).getAll()
        m = m0.renumbered(locality_permutation(m0.geometry.x))
        sub = MeshTags(m, s0.dim, s0.indices, s0.values)
        ctx2 = ShardContext(m.num_vertices)
        loc2, sub_l, _, _ = ctx2.distribute(m, sub, None, [])
        with contextlib.redirect_stdout(io.StringIO()):
            res["micca"] = micca(loc2, sub_l)
            if rank == 0:
                res["micca_serial"] = micca(m, sub)
        out[rank] = res
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_degree2_on_a_distributed_mesh_two_ranks(golden_dir):
    """Quadratic elements under the row-sharded path (VERDICT r4 missing 3: the reference's signatures accept ``degree`` under
    mpirun, acoustic_matrices.py:25, flame_matrices.py:131,193).  (i) ShapeSensitivities/RijkeFFD (main_shape.py:56-83: degree 2,
    Robin ends, distributed n-tau flame) on its 'Original' mesh cut into two parts: each rank assembles the rows of ITS vertices
    and edges on its part (edges belong to the owner of their lower vertex), the quadratic space brings its own ShardContext, the
    solve is collective.  Trajectory and eigenvalue of the one-GPU run on the whole mesh to 1e-9, the normalised mode dof by dof
    (matched through the dof coordinates) to 1e-7 of its peak, ghost entries equal to their owners'.  (ii) the factors of the
    pointwise flame matrix with degree 2 on the real MICCA mesh (16 flames: load vectors over owned vertices and edges, the ten
    quadratic gradients at the reference point on the rank that holds its cell): the union over the ranks is the serial factor."""
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_degree2_distributed_worker, args=(world, port, out, golden_dir), nprocs=world, join=True)
    assert len(out) == world
    ser = out[0]["rijke_serial"]
    p_ser = dict(zip(ser["keys"], ser["p"]))
    peak = np.abs(ser["p"]).max()
    assert sum(out[r]["rijke"]["sizes"][1] for r in range(world)) == ser["size"] == out[0]["rijke"]["n_global"] == 41135
    for rank in range(world):
        r = out[rank]["rijke"]
        assert len(r["hist"]) == len(ser["hist"])
        for a, b in zip(r["hist"], ser["hist"]):
            assert abs(a - b) / abs(b) < 1e-9, (a, b)
        assert abs(r["omega"] - ser["omega"]) / abs(ser["omega"]) < 1e-9
        assert r["stats"]["krylov"] == ser["stats"]["krylov"] and len(r["stats"]["sharded_levels"]) >= 1
        err = max(abs(v - p_ser[k]) for k, v in zip(r["keys"], r["p_own"]))
        assert err < 1e-7 * peak, err
        assert max(abs(v - p_ser[k]) for k, v in zip(r["ghost_keys"], r["p_ghost"])) < 1e-7 * peak
        nv_own, n_own, n_loc = r["sizes"]
        assert nv_own < n_own < n_loc                        # owned vertices + owned edges, then ghosts
    for f, (left_s, right_s) in enumerate(out[0]["micca_serial"]):
        left, right = {}, {}
        for rank in range(world):
            l, rgt = out[rank]["micca"][f]
            assert not (set(l) & set(left)) and not (set(rgt) & set(right))            # every entry on exactly one rank
            left.update(l)
            right.update(rgt)
        assert set(left) == set(left_s) and set(right) == set(right_s) and len(right_s) == 10
        for k, v in left_s.items():
            assert abs(left[k] - v) <= 1e-14 * abs(v)
        for k, v in right_s.items():
            assert abs(right[k] - v) <= 1e-12 * max(abs(v), 1.0)
    print("degree 2 distributed:", out[0]["rijke"]["stats"]["levels"], out[0]["rijke"]["stats"]["sharded_levels"],
          [out[r]["rijke"]["sizes"] for r in range(world)], ser["stats"]["gmres_iters"], out[0]["rijke"]["stats"]["gmres_iters"])


def _fallback_worker(rank, world, port, out):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "helmholtz-x_amd")]
    os.environ["HX_NATIVE_COMM"] = "try"                       # ask for the library's communicator, do not insist
    os.environ["HX_RCCL_LIB"] = "/nonexistent/librccl.so"      # ... which cannot be bound
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        from helmholtz_x.sharded import ShardContext
        ctx = ShardContext(1000)
        t = torch.full((3,), float(rank + 1), dtype=torch.float64)
        ctx.all_reduce_(t)
        out[rank] = (ctx.comm is None, t.tolist(), ctx.counters()["issued_by"])
        os.environ["HX_NATIVE_COMM"] = "1"                     # insisting on it is an error on every rank
        try:
            ShardContext(1000)
            out[("raised", rank)] = False
        except RuntimeError:
            out[("raised", rank)] = True
    finally:
        dist.destroy_process_group()


def test_missing_rccl_library_falls_back_to_torch_distributed_on_every_rank():
    """The library-issued RCCL path is an optimisation of the transport, not a requirement: when librccl cannot be bound the ranks
    agree (one all-reduce of a flag) to drive the exchanges through torch.distributed instead -- no rank is left with a
    communicator the others do not have -- and HX_NATIVE_COMM=1 turns that into an error."""
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_fallback_worker, args=(world, port, out), nprocs=world, join=True)
    for rank in range(world):
        no_comm, summed, issued = out[rank]
        assert no_comm and summed == [3.0, 3.0, 3.0] and issued.startswith("torch.distributed")
        assert out[("raised", rank)]


def _rccl_one_rank_worker(rank, world, port, out):
    import ctypes as C
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "helmholtz-x_amd"), os.path.join(root, "tests")]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        from helmholtz_x import _hx
        from helmholtz_x.sharded import ShardContext
        ctx = ShardContext(64)
        assert ctx.comm is not None and "rccl" in ctx.comm_library
        t = torch.arange(12, dtype=torch.float64, device="cuda").view(6, 2).contiguous()
        z = torch.view_as_complex(t)
        ref = z.clone()
        ctx.all_reduce_(z)                       # ncclAllReduce issued by libhx on torch's current stream (one rank: identity)
        ctx.all_reduce_(z, op="max")
        torch.cuda.synchronize()
        assert torch.equal(z, ref)
        # a plan whose only peer is this rank itself: one list sent in place (contiguous run), one gathered into the staging
        # buffer; grouped ncclSend / ncclRecv on the compute stream
        sp = np.array([0, 0], np.int32)
        sptr = np.array([0, 4, 7], np.int64)
        sidx = np.array([10, 11, 12, 13, 1, 7, 2], np.int32)
        rp_ = np.array([0, 0], np.int32)
        roff, rcnt = np.array([20, 30], np.int64), np.array([4, 3], np.int64)
        _hx.check(_hx.lib.hx_comm_set_plan(ctx.comm, 5, 2, _hx.nptr(sp), _hx.nptr(sptr), _hx.nptr(sidx), 2, _hx.nptr(rp_),
                                           _hx.nptr(roff), _hx.nptr(rcnt)))
        for dtype in (torch.complex128, torch.complex64, torch.float32):
            x = torch.arange(40, device="cuda").to(dtype) * (1 + (2j if dtype.is_complex else 0))
            before = x.clone()
            _hx.check(_hx.lib.hx_comm_exchange(ctx.comm, 5, _hx.ptr(x), _hx.ptr(x), x.element_size(), _hx.stream()))
            torch.cuda.synchronize()
            want = before.clone()
            want[20:24] = before[10:14]
            want[30:33] = before[torch.tensor([1, 7, 2], device="cuda")]
            assert torch.equal(x, want), dtype
        # split form: the exchange stays on the compute stream (every operation of the communicator does); begin hands out the
        # communicator's side stream, ordered after the compute stream's work so far; work launched THERE between begin and end
        # (here: an update of an entry the exchange does not touch) overlaps the exchange, end joins it
        x = torch.arange(40, device="cuda").to(torch.complex128)
        before = x.clone()
        side = C.c_void_p()
        _hx.check(_hx.lib.hx_comm_halo_begin(ctx.comm, 5, _hx.ptr(x), 16, _hx.stream(), C.byref(side)))
        assert side.value is not None and side.value != _hx.stream()
        with torch.cuda.stream(torch.cuda.ExternalStream(side.value)):
            x[39] = -1.0
        rc = _hx.lib.hx_comm_halo_begin(ctx.comm, 5, _hx.ptr(x), 16, _hx.stream(), C.byref(C.c_void_p()))
        assert rc != 0                               # a second begin before the end is refused
        _hx.check(_hx.lib.hx_comm_halo_end(ctx.comm, 5, _hx.stream()))
        assert _hx.lib.hx_comm_halo_end(ctx.comm, 5, _hx.stream()) != 0     # ... and so is an end without a begin
        y = x * 2                                    # ordered after the exchange (stream order) and the side stream (end event)
        torch.cuda.synchronize()
        want = before.clone()
        want[20:24], want[30:33], want[39] = before[10:14], before[torch.tensor([1, 7, 2], device="cuda")], -1.0
        assert torch.equal(x, want) and torch.equal(y, 2 * want)
        assert _hx.lib.hx_comm_overlapped(ctx.comm) == 1
        assert _hx.lib.hx_comm_stream_switches(ctx.comm) == 0          # everything above was issued on ONE stream
        ne, na = C.c_int64(0), C.c_int64(0)
        _hx.check(_hx.lib.hx_comm_counters(ctx.comm, C.byref(ne), C.byref(na), 0))
        out[0] = (ne.value, na.value, ctx.comm_library)
        ctx.destroy()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_library_issued_rccl_calls_on_one_rank():
    """The REAL librccl through libhx's run-time binding (csrc/comm.hip), as far as a one-GPU box can take it: communicator
    from an id drawn by the library, ncclAllReduce (sum, max) and a grouped ncclSend/ncclRecv exchange (in-place run +
    gathered list, three element sizes) issued by the library on torch's current stream -- with this rank as its own peer; and
    the split form of the exchange (hx_comm_halo_begin / _end: exchange on the compute stream, interior work on the communicator's
    side stream, ordered by events; all operations of the communicator on one stream)."""
    port = _free_port()
    out = mp.Manager().dict()
    mp.spawn(_rccl_one_rank_worker, args=(1, port, out), nprocs=1, join=True)
    assert out[0][0] == 4 and out[0][1] == 2


def _micca_distributed_worker(rank, world, port, out, golden_dir, mode="fpi"):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "helmholtz-x_amd"), os.path.join(root, "tests")]
    os.environ["HX_QUIET"] = "1"
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        import contextlib
        import io
        from scipy.io import loadmat
        from cases import Micca
        from helmholtz_x.acoustic_matrices import AcousticMatrices
        from helmholtz_x.eigensolvers import fixed_point_iteration
        from helmholtz_x.fem import Function, FunctionSpace, MeshTags, locality_permutation
        from helmholtz_x.flame_matrices import PointwiseFlameMatrix
        from helmholtz_x.flame_transfer_function import stateSpace
        from helmholtz_x.io_utils import XDMFReader
        from helmholtz_x.parameters_utils import Q_multiple
        from helmholtz_x.sharded import ShardContext
        mesh0, sub0, ft0 = XDMFReader(os.path.join(golden_dir, "meshes", "micca_full", "mesh")).getAll()
        # the file's numbering is arbitrary: renumber along a space-filling curve (what a distributing reader does before it
        # cuts the mesh), then contiguous index ranges are compact pieces of the annulus
        mesh = mesh0.renumbered(locality_permutation(mesh0.geometry.x))
        subdomains = MeshTags(mesh, sub0.dim, sub0.indices, sub0.values)
        facet_tags = MeshTags(mesh, ft0.dim, ft0.indices, ft0.values)
        c = Function(FunctionSpace(mesh, "DG", 0), "soundspeed")
        c.x.array[:] = Micca.c_of_z(c.function_space.tabulate_dof_coordinates()[:, 2])
        ctx = ShardContext(mesh.num_vertices)
        local, sub_l, ft_l, (c_l,) = ctx.distribute(mesh, subdomains, facet_tags, [c])
        with contextlib.redirect_stdout(io.StringIO()):
            matrices = AcousticMatrices(local, ft_l, {11: {'Robin': Micca.R_outlet}}, c_l, degree=1)
            mat = loadmat(os.path.join(golden_dir, "micca_ftf.mat"))
            FTF = stateSpace(mat['A'], mat['b'], mat['c'], mat['d'])
            h = Q_multiple(local, sub_l, Micca.N_sector)
            D = PointwiseFlameMatrix(local, sub_l, Micca.x_r(), h, Micca.rho_amb, Micca.q_0, Micca.u_b, FTF, degree=1)
            D.assemble_submatrices('direct')
            if mode == "newton":
                import re
                from helmholtz_x.eigensolvers import newtonSolver
                buf = io.StringIO()
                with contextlib.redirect_stdout(buf):
                    omega, p = newtonSolver(matrices, D, 3260 + 460j, i=0, nev=2, tol=1e-2, st_type="amg", shard=ctx)
                lines = re.findall(r"iter =\s*(\d+),\s+omega = ([+-][\d.]+)\s+([+-][\d.]+)j,\s+\|domega\| = ([\d.e+-]+)",
                                   buf.getvalue())
                n_own = local.part["n_own"]
                out[rank] = dict(omega=complex(omega), lines=[(float(a), float(b), float(c)) for _, a, b, c in lines],
                                 p_own=p.x.array[:n_own].copy(), x_own=local.geometry.x[:n_own].copy())
                return
            hist = []
            st = "lu" if mode == "lu" else ("auto" if mode == "auto" else "amg")
            if mode == "ladder":                 # two iterations are too few for every iterative rung (the reference's own option name)
                from helmholtz_x.petsc import Options
                Options()["st_ksp_max_it"] = 2
            E = fixed_point_iteration(matrices, D, Micca.target, i=0, nev=4, tol=1e-3, st_type=st, history=hist, shard=ctx)
        inner = [v for v in E.inner_cache.values() if hasattr(v, "stats")][0]
        if mode in ("lu", "auto", "ladder"):
            out[rank] = dict(hist=list(hist), eigs=[E.getEigenvalue(i) for i in range(2)], stats=inner.stats(),
                             solver=type(getattr(inner, "inner", inner)).__name__)
            return
        # the adjoint problem as active_fpi.py:64-69 runs it: transposed flame factors, B^H, conjugate target (default nev = 2)
        with contextlib.redirect_stdout(io.StringIO()):
            D.assemble_submatrices('adjoint')
            E_adj = fixed_point_iteration(matrices, D, Micca.target.conjugate(), i=0, tol=1e-3, problem_type='adjoint',
                                          st_type="amg", shard=ctx)
        out[rank] = dict(hist=list(hist), eigs=[E.getEigenvalue(i) for i in range(2)], stats=inner.stats(),
                         eigs_adj=[E_adj.getEigenvalue(i) for i in range(2)],
                         peers=(sorted(ctx.plan.send_to), sorted(ctx.plan.recv_from)),
                         sizes=(local.part["n_own"], local.num_vertices, local.num_cells, mesh.num_vertices, mesh.num_cells),
                         contiguous_sends=[bool(np.all(np.diff(v) == 1)) for v in ctx.plan.send_to.values()])
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_micca_real_mesh_distributed_over_three_ranks(ref_logs, golden_dir):
    """The reference's flagship case on its REAL (unstructured, gmsh) mesh -- 34,787 DOF / 163,141 tets, 16 pointwise flames
    located by point search, state-space FTF, Robin outlet -- partitioned over THREE ranks (one GPU shared, gloo): every rank
    holds a third of the mesh + ghost layer, assembles its rows, finds the cells of the reference points it holds (the ranks
    agree on the lowest global cell id), builds its slab of the hierarchy and takes part in the collective solve.  The
    logged trajectory (3 decimals) and the two stored 16-digit eigenvalues of the reference's own `mpirun` runs
    (fullAnnulus/Results/Active/FPI/activeParallel.log, eigenvalues_dir.txt) must be reproduced to 1e-8, and so must the two
    adjoint eigenvalues (eigenvalues_adj.txt: transposed flame factors, B^H, conjugate target).  The partition of
    an unstructured mesh is no ring: ranks have several peers and send gathered (non-contiguous) lists."""
    world, port = 3, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_micca_distributed_worker, args=(world, port, out, golden_dir), nprocs=world, join=True)
    assert len(out) == world
    g = ref_logs["micca_full"]
    for rank in range(world):
        r = out[rank]
        assert len(r["hist"]) - 1 == len(g["trajectory_3dec"])
        for z, ref in zip(r["hist"][1:], g["trajectory_3dec"]):
            assert abs(z - complex(*ref)) < 7.5e-4
        for i, ref in enumerate(g["direct"]):
            ref = complex(*ref)
            assert abs(r["eigs"][i] - ref) / abs(ref) < 1e-8, (i, r["eigs"][i], ref)
        for i, ref in enumerate(g["adjoint"]):
            ref = complex(*ref)
            assert abs(r["eigs_adj"][i] - ref) / abs(ref) < 1e-8, (i, r["eigs_adj"][i], ref)
        n_own, nv_loc, nc_loc, nv, nc = r["sizes"]
        assert nv_loc < 0.55 * nv and nc_loc < 0.55 * nc
        assert len(r["peers"][0]) >= 1 and not all(r["contiguous_sends"])
    assert sum(out[r]["sizes"][0] for r in range(world)) == out[0]["sizes"][3]
    print("micca distributed:", out[0]["stats"], [out[r]["peers"] for r in range(world)], [out[r]["sizes"] for r in range(world)])


@pytest.mark.gpu
@pytest.mark.parametrize("st_type", ["lu", "auto", "ladder"])
def test_micca_real_mesh_direct_solver_replicated_over_three_ranks(ref_logs, golden_dir, st_type):
    """st_type='lu' on the row-sharded path (what MUMPS / PCLU is to the reference's `mpirun -np 8` runs; SURVEY 8(e): "replicas only
    for the solve"): the ranks' row blocks of K are all-gathered once, every rank factorises the global operator on the device and
    a solve is one all-gather of the right-hand side + the local sweeps.  The reference's flagship case on its real mesh over
    three ranks: the logged trajectory and the stored 16-digit eigenvalues to 1e-8, identical on every rank; 'auto' picks the
    replicated factor too at this size (34,787 <= lu_max_dofs); "ladder": st_type='amg' with -st_ksp_max_it 2 -- every iterative
    rung stalls on every rank at once and the ladder ends in the replicated factor (amg.AMGSolver._go_direct, collective)."""
    world, port = 3, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_micca_distributed_worker, args=(world, port, out, golden_dir, st_type), nprocs=world, join=True)
    assert len(out) == world
    g = ref_logs["micca_full"]
    for rank in range(world):
        r = out[rank]
        assert len(r["hist"]) - 1 == len(g["trajectory_3dec"])
        for z, ref in zip(r["hist"][1:], g["trajectory_3dec"]):
            assert abs(z - complex(*ref)) < 7.5e-4
        for i, ref in enumerate(g["direct"]):
            ref = complex(*ref)
            assert abs(r["eigs"][i] - ref) / abs(ref) < 1e-8, (i, r["eigs"][i], ref)
        st = r["stats"]
        if st_type == "ladder":
            assert st["rung"] == "direct" and [q["rung"] for q in st["ladder"]] == ["wide", "shifted", "direct"], st
            st = st["direct"]
        assert st["replicated_over"] == world and st["rhs_all_gathers"] == st["solves"] > 0 and st["nnz_L"] > 0, st
        assert r["eigs"] == out[0]["eigs"]                      # the same factors, the same sweeps: the same bits on every rank
    print("micca replicated LU:", out[0]["stats"])


def _rijke_distributed_worker(rank, world, port, out, golden_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "helmholtz-x_amd"), os.path.join(root, "tests")]
    os.environ["HX_QUIET"] = "1"
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        import contextlib
        import io
        from cases import Rijke as R
        from helmholtz_x.acoustic_matrices import AcousticMatrices
        from helmholtz_x.eigensolvers import fixed_point_iteration
        from helmholtz_x.fem import MeshTags, locality_permutation
        from helmholtz_x.flame_matrices import DistributedFlameMatrix
        from helmholtz_x.flame_transfer_function import nTau
        from helmholtz_x.io_utils import XDMFReader
        from helmholtz_x.parameters_utils import c_step, gaussianFunction, rho_step, temperature_step
        from helmholtz_x.sharded import ShardContext
        res = {}
        for dim in (2, 3):
            mesh0, sub0, ft0 = XDMFReader(os.path.join(golden_dir, "meshes", "rijke%dd" % dim)).getAll()
            mesh = mesh0.renumbered(locality_permutation(mesh0.geometry.x))
            ft = MeshTags(mesh, ft0.dim, ft0.indices, ft0.values)
            ctx = ShardContext(mesh.num_vertices)
            local, _, ft_l, _ = ctx.distribute(mesh, None, ft, [])
            with contextlib.redirect_stdout(io.StringIO()):
                # every field is built ON the rank's part of the mesh, as the reference's script does under mpirun
                T = temperature_step(local, R.x_f(dim), R.T_u, R.T_d)
                param = T if dim == 2 else c_step(local, R.x_f(dim), R.c_u, R.c_d)        # (as the reference's two scripts)
                M = AcousticMatrices(local, ft_l, {t: {'Neumann'} for t in (1, 2, 3, 4)}, param)
                rho = rho_step(local, R.x_f(dim), R.a_f, R.rho_d, R.rho_u)
                w = gaussianFunction(local, R.x_r(dim), R.a_r)            # normalised by a global integral (allreduce)
                h = gaussianFunction(local, R.x_f(dim), R.a_f)
                D = DistributedFlameMatrix(local, w, h, rho, T, R.q_0, R.u_b, nTau(R.n(dim), R.tau))
                D.assemble_submatrices()
                hist = []
                E = fixed_point_iteration(M, D, R.target, nev=2, i=0, st_type="amg", history=hist, shard=ctx)
                from helmholtz_x.eigenvectors import normalize_eigenvector
                omega, p = normalize_eigenvector(local, E, 0, degree=1, which='right', print_eigs=False)
            n_own = local.part["n_own"]
            res[dim] = (list(hist), complex(omega), local.geometry.x[:n_own].copy(), p.x.array[:n_own].copy(),
                        # ghost entries of the normalised mode equal their owners' values: checked below
                        local.part["local_to_global"][n_own:].copy(), p.x.array[n_own:].copy(), ctx.r0)
        out[rank] = res
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_rijke_distributed_flame_eps_path_over_two_ranks(ref_logs, golden_dir):
    """The linear (EPS) branch of the fixed-point iteration with a DistributedFlameMatrix (Gaussian heat-release and
    measurement fields, flame_matrices.py:191-244) on distributed meshes: the reference's Rijke 2-D and 3-D cases
    (NetworkCode/RijkeTube2D, RijkeTube3D: gmsh meshes, Neumann walls, n-tau flame) over two ranks.  Every coefficient field is
    built on the rank's part of the mesh (the Gaussians are normalised by a global integral), the flame vectors keep the owned
    entries, the logged 5-step trajectories of the reference's runs are reproduced to 1e-8, and ``normalize_eigenvector`` on
    the distributed mesh returns the committed eigenvector (Results/Active/p.h5) to 1e-7."""
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_rijke_distributed_worker, args=(world, port, out, golden_dir), nprocs=world, join=True)
    assert len(out) == world
    from test_oracle_goldens import match_nodes, sign_free_error
    vecs = np.load(os.path.join(golden_dir, "eigenvectors", "reference_eigenvectors.npz"))
    for dim, key in ((2, "rijke2d"), (3, "rijke3d")):
        for rank in range(world):
            hist, traj = out[rank][dim][0], ref_logs[key]["trajectory"]
            assert len(hist) == len(traj)
            for z, ref in zip(hist, traj):
                ref = complex(*ref)
                assert abs(z - ref) / abs(ref) < 1e-8, (dim, z, ref)
            assert abs(out[rank][dim][1] - hist[-1]) < 1e-6 * abs(hist[-1])
        # normalize_eigenvector on the distributed mesh (phase on global entry 0, int p p dx = 1 by all-reduce): the ranks'
        # owned parts together are the mode the reference committed (Results/Active/p.h5), matched by coordinates
        x = np.concatenate([out[r][dim][2] for r in range(world)])[:, :dim]
        p = np.concatenate([out[r][dim][3] for r in range(world)])
        assert sign_free_error(p, vecs[key + "_dir_p"][match_nodes(x, vecs[key + "_dir_x"])]) < 1e-7
        for r in range(world):          # ghost entries = the owners' values
            ghosts, vals = out[r][dim][4], out[r][dim][5]
            assert np.allclose(vals, p[ghosts], rtol=0, atol=1e-13 * np.abs(p).max())


def _dirichlet_distributed_worker(rank, world, port, out):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "helmholtz-x_amd"), os.path.join(root, "tests")]
    os.environ["HX_QUIET"] = "1"
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        import contextlib
        import io
        import bench
        from helmholtz_x.acoustic_matrices import AcousticMatrices
        from helmholtz_x.eigensolvers import eps_solver
        from helmholtz_x.sharded import ShardContext
        case = bench.build_case((6, 96, 8))                       # 6,048 DOF annulus; tag 11 = outlet plane
        mesh, ft, c = case["mesh"], case["facet_tags"], case["c"]
        bcs = {11: {'Dirichlet'}}
        target = 3000.0                                           # (rad/s: the solver shifts at target ** 2)
        with contextlib.redirect_stdout(io.StringIO()):
            M0 = AcousticMatrices(mesh, ft, bcs, c)
            E0 = eps_solver(M0.A, M0.C, target, nev=6, st_type="amg")
            ctx = ShardContext(mesh.num_vertices, align=7 * 9)    # cut between theta-planes
            local, _, ft_l, (c_l,) = ctx.distribute(mesh, None, ft, [c])
            M1 = AcousticMatrices(local, ft_l, bcs, c_l)
            E1 = eps_solver(M1.A, M1.C, target, nev=6, st_type="amg", shard=ctx)
        # Dirichlet rows of the local operators: unit diagonal, nothing else; Dirichlet COLUMNS zero, ghost ones included
        A = M1.A.csr_part()
        rp, ci, va = A.rowptr.cpu().numpy(), A.col.cpu().numpy(), A.val.cpu().numpy()
        n_own = local.part["n_own"]
        glob = local.part["local_to_global"]
        is_bc_global = np.zeros(mesh.num_vertices, dtype=bool)
        is_bc_global[np.unique(mesh.facets[ft.find(11)])] = True
        bc_loc = is_bc_global[glob]
        row_of = np.repeat(np.arange(n_own), np.diff(rp))
        offdiag_bc = (bc_loc[row_of] | bc_loc[ci]) & (ci != row_of)
        diag_bc = bc_loc[row_of] & (ci == row_of)
        out[rank] = dict(e0=[E0.getEigenvalue(i) for i in range(6)], e1=[E1.getEigenvalue(i) for i in range(6)],
                         offdiag_zero=bool(np.all(va[offdiag_bc] == 0)), diag_one=bool(np.all(va[diag_bc] == 1.0)),
                         n_bc_rows=int(bc_loc[:n_own].sum()), n_bc_ghost=int(bc_loc[n_own:].sum()))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_dirichlet_rows_on_a_distributed_mesh_two_ranks():
    """AcousticMatrices with a Dirichlet boundary (acoustic_matrices.py:43-50, 101-103, 121-123: DOLFINx zeroes rows and columns and
    puts 1 on the diagonal) on a distributed mesh: the owner flags its Dirichlet vertices, the ghosts' flags travel by one halo
    exchange, and the rows / columns of the local operators are treated exactly as on one GPU -- same eigenvalues of the
    passive problem to 1e-9, Dirichlet columns zero also where the column is a ghost."""
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_dirichlet_distributed_worker, args=(world, port, out), nprocs=world, join=True)
    assert len(out) == world
    for rank in range(world):
        r = out[rank]
        assert r["offdiag_zero"] and r["diag_one"] and r["n_bc_rows"] > 0 and r["n_bc_ghost"] > 0
        # (the unit rows give the pencil the eigenvalue -1 once per Dirichlet vertex, as in the reference: members of that cluster
        # come back in arbitrary combinations; the physical eigenvalues -- 3824.64^2, the 4884.90^2 pair -- must agree)
        phys0 = [z for z in r["e0"] if abs(z + 1.0) > 0.5]
        phys1 = [z for z in r["e1"] if abs(z + 1.0) > 0.5]
        assert len(phys0) >= 3 and len(phys1) >= 3 and abs(np.sqrt(phys0[0]) - 3824.644273) < 1e-4
        for a, b in zip(phys0[:3], phys1[:3]):
            assert abs(a - b) < 1e-9 * abs(a), (a, b)


def _flamedduct_distributed_worker(rank, world, port, out, golden_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "helmholtz-x_amd"), os.path.join(root, "tests")]
    os.environ["HX_QUIET"] = "1"
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        import contextlib
        import io
        from cases import FlamedDuct as params
        from helmholtz_x.acoustic_matrices import AcousticMatrices
        from helmholtz_x.eigensolvers import fixed_point_iteration
        from helmholtz_x.fem import MeshTags, locality_permutation
        from helmholtz_x.flame_matrices import DistributedFlameMatrix
        from helmholtz_x.flame_transfer_function import nTau
        from helmholtz_x.io_utils import XDMFReader
        from helmholtz_x.parameters_utils import gaussianFunction, halfGaussianFunction, rho_ideal, temperature_step
        from helmholtz_x.sharded import ShardContext
        mesh0, _, ft0 = XDMFReader(os.path.join(golden_dir, "meshes", "flamedduct", "mesh")).getAll()
        mesh = mesh0.renumbered(locality_permutation(mesh0.geometry.x))
        ft = MeshTags(mesh, ft0.dim, ft0.indices, ft0.values)
        ctx = ShardContext(mesh.num_vertices)
        local, _, ft_l, _ = ctx.distribute(mesh, None, ft, [])
        with contextlib.redirect_stdout(io.StringIO()):
            T = temperature_step(local, params.x_flame, params.T_passive, params.T_flame)
            M = AcousticMatrices(local, ft_l, {3: {"ChokedInlet": params.M0}, 8: {"ChokedOutlet": params.M1}}, T, degree=1)
            rho = rho_ideal(T, params.p_gas, params.r_gas)
            w = gaussianFunction(local, params.x_ref, params.a_ref)
            h = halfGaussianFunction(local, params.x_flame, params.a_flame)
            D = DistributedFlameMatrix(local, w, h, rho, T, params.q_0, params.u_b, nTau(params.n, params.tau), degree=1)
            D.assemble_submatrices()
            hist = []
            fixed_point_iteration(M, D, params.target_active, nev=2, st_type="amg", history=hist, shard=ctx)
        has = [int(ft_l.find(t).size) for t in (3, 8)]
        out[rank] = (list(hist), has)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_flamedduct_distributed_over_two_ranks(ref_logs, golden_dir):
    """Longitudinal/NetworkCode/FlamedDuct/active.py on a distributed mesh: choked inlet and outlet whose reflection
    coefficients need the FACET-AVERAGED gamma (every facet counted on the rank that owns it, sums all-reduced:
    acoustic_matrices.py:76-90), temperature-driven variable gamma, ideal-gas density, a half-Gaussian heat-release field
    normalised twice by global integrals, Gaussian flame vectors, quadratic (PEP) branch.  The reference's own 8-rank run of
    this case differs from its serial run by 2e-4 (fields set without a ghost update); here the two-rank
    run reproduces the SERIAL logged trajectory to 1e-8."""
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_flamedduct_distributed_worker, args=(world, port, out, golden_dir), nprocs=world, join=True)
    assert len(out) == world
    traj = ref_logs["flamedduct"]["active_trajectory"]
    for rank in range(world):
        hist = out[rank][0]
        assert len(hist) - 1 >= len(traj)
        for z, ref in zip(hist[1:1 + len(traj)], traj):
            ref = complex(*ref)
            assert abs(z - ref) / abs(ref) < 1e-8, (z, ref)
    # inlet and outlet sit at opposite ends of the duct: with two ranks each end belongs (mostly) to another rank
    assert sum(out[r][1][0] for r in range(world)) > 0 and sum(out[r][1][1] for r in range(world)) > 0


@pytest.mark.gpu
def test_micca_newton_solver_distributed_over_two_ranks(ref_logs, golden_dir):
    """active_newton.py (eigensolvers.py:278-348) on the distributed real MICCA mesh: two-sided shift-invert solves (left
    eigenvectors from the transposed pencil, collectively), eigenvectors normalised on the distributed mesh, the brackets
    p_adj^H L'(omega) p and p_adj^H C p as sums over the ranks (row-sharded MatMult with the factored flame term + VecDot).  The
    reference's logged run for mode 0 -- 19 iterations, omega to its 2 printed decimals, |domega| to its 3 digits -- and the stored
    16-digit result to 1e-9, as on one GPU."""
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_micca_distributed_worker, args=(world, port, out, golden_dir, "newton"), nprocs=world, join=True)
    assert len(out) == world
    g = ref_logs["micca_full_newton"]
    ref = g["iterations_omega_2dec_and_domega"][0]
    final = complex(*g["final"][0])
    for rank in range(world):
        r = out[rank]
        assert len(r["lines"]) == len(ref)
        for (re_, im_, dom), (rr, ri, rd) in zip(r["lines"], ref):
            assert abs(re_ - rr) < 0.011 and abs(im_ - ri) < 0.011
            assert abs(dom - rd) <= 0.011 * rd
        assert abs(r["omega"] - final) / abs(final) < 1e-9, (r["omega"], final)


def _bloch_sharded_worker(rank, world, port, out, shape=(12, 12, 48), transport="gloo"):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "helmholtz-x_amd"), os.path.join(root, "tests")]
    os.environ["HX_QUIET"] = "1"
    if transport == "fake_rccl":    # exchanges issued BY THE LIBRARY (hx_comm) through the asynchronous librccl stand-in
        os.environ["HX_NATIVE_COMM"] = "1"
        os.environ["HX_RCCL_LIB"] = os.path.join(root, "tests", "fake_rccl", "libfakerccl.so")
        os.environ["HX_AMG_REPLICATE_MAX_ROWS"] = "20000"     # levels 0 AND 1 stay row-sharded (422 k -> 52 k -> replicated tail)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        import contextlib
        import io
        from scipy.io import loadmat
        from cases import Micca
        from helmholtz_x.acoustic_matrices import AcousticMatrices
        from helmholtz_x.bloch_operator import Blochifier, sector_slab_numbering
        from helmholtz_x.eigensolvers import fixed_point_iteration
        from helmholtz_x.eigenvectors import normalize_eigenvector
        from helmholtz_x.fem import MeshTags
        from helmholtz_x.flame_matrices import PointwiseFlameMatrix
        from helmholtz_x.flame_transfer_function import stateSpace
        from helmholtz_x.parameters_utils import Q_multiple
        from helmholtz_x.sharded import ShardContext
        from helmholtz_x.synthetic import annulus, micca_sound_speed
        half = np.pi / 16

        class Geometry:
            pass

        mesh0, sub0, ft0 = annulus(*shape, n_flames=1, sector=(-half, half))
        perm, planes = sector_slab_numbering(mesh0, "z")             # numbering that runs z-plane by z-plane
        G = Geometry()
        G.mesh = mesh0.renumbered(perm)
        G.facet_tags = MeshTags(G.mesh, ft0.dim, ft0.indices, ft0.values)
        sub = MeshTags(G.mesh, sub0.dim, sub0.indices, sub0.values)
        bcs = {11: 'Neumann', 12: 'Master', 13: 'Slave'}
        mat = loadmat(os.path.join(root, "tests", "golden", "micca_ftf.mat"))
        FTF = stateSpace(mat['A'], mat['b'], mat['c'], mat['d'])
        x_r = np.array([[Micca.r_r, 0.0, Micca.z_r]])
        c = micca_sound_speed(G.mesh)
        res = {}
        with contextlib.redirect_stdout(io.StringIO()):
            # one GPU, same numbering
            M = AcousticMatrices(G.mesh, G.facet_tags, bcs, c)
            bloch = Blochifier(G, bcs, 16, M)
            D = PointwiseFlameMatrix(G.mesh, sub, x_r, Q_multiple(G.mesh, sub, 1), Micca.rho_amb, Micca.q_0, Micca.u_b, FTF, bloch_object=bloch)
            D.assemble_submatrices('direct')
            D.blochify('direct')
            hist = []
            E = fixed_point_iteration(bloch, D, Micca.target, i=0, nev=2, tol=1e-3, history=hist)
            w1, p1 = normalize_eigenvector(G.mesh, E, 0, BlochRemapper=bloch.remapper, print_eigs=False)
            res["single"] = dict(eig=E.getEigenvalue(0), hist=list(hist), p=p1.x.array.copy())
            # two ranks: z-slabs cut between two planes, so that every master/slave pair is rank-local
            n = G.mesh.num_vertices
            cuts = [0] + [int(planes[np.argmin(np.abs(planes - (q * n) // world))]) for q in range(1, world)] + [n]
            ranges = [(cuts[q], cuts[q + 1]) for q in range(world)]
            ctx = ShardContext(n, ranges=ranges)
            local, sub_l, ft_l, (c_l,) = ctx.distribute(G.mesh, sub, G.facet_tags, [c])
            L = Geometry()
            L.mesh, L.facet_tags = local, ft_l
            M_l = AcousticMatrices(local, ft_l, bcs, c_l)
            bloch_l = Blochifier(L, bcs, 16, M_l, global_geometry=G)
            D_l = PointwiseFlameMatrix(local, sub_l, x_r, Q_multiple(local, sub_l, 1), Micca.rho_amb, Micca.q_0, Micca.u_b, FTF,
                                       bloch_object=bloch_l)
            D_l.assemble_submatrices('direct')
            D_l.blochify('direct')
            hist_l = []
            E_l = fixed_point_iteration(bloch_l, D_l, Micca.target, i=0, nev=2, tol=1e-3, history=hist_l, shard=bloch_l.shard,
                                        st_type="amg")          # (auto would take the replicated direct factor at this size)
            w2, p2 = normalize_eigenvector(local, E_l, 0, BlochRemapper=bloch_l.remapper, print_eigs=False)
            if transport == "gloo":                              # (the small sector: also through the replicated direct factor)
                E_lu = fixed_point_iteration(bloch_l, D_l, Micca.target, i=0, nev=2, tol=1e-3, shard=bloch_l.shard, st_type="lu")
                inner_lu = [v for v in E_lu.inner_cache.values() if hasattr(v, "stats")][0]
                res["lu"] = dict(eig=E_lu.getEigenvalue(0), stats=inner_lu.stats())
            D_l.assemble_submatrices('adjoint')
            D_l.blochify('adjoint')
            E_a = fixed_point_iteration(bloch_l, D_l, Micca.target.conjugate(), i=0, nev=2, tol=1e-3, problem_type='adjoint',
                                        shard=bloch_l.shard, st_type="amg")
            inner = [v for v in E_l.inner_cache.values() if hasattr(v, "stats")][0]
            l2g = local.part["local_to_global"]
            n_own = local.part["n_own"]
            res["sharded"] = dict(eig=E_l.getEigenvalue(0), hist=list(hist_l), adj=E_a.getEigenvalue(0), stats=inner.stats(),
                                  p_own=p2.x.array[:n_own].copy(), own_ids=np.asarray(l2g[:n_own]), pairs=int(bloch_l.dofs_master.size),
                                  n_red=(bloch_l.shard.n_own, bloch_l.shard.n_ext, bloch_l.shard.n_global),
                                  f_b=bloch_l.f_b, pm=p2.x.array[bloch_l.dofs_master].copy(), ps=p2.x.array[bloch_l.dofs_slave].copy(),
                                  counters=bloch_l.shard.counters())
        out[rank] = res
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bloch_sector_row_sharded_over_two_ranks_matches_single_gpu():
    """BASELINE config 5 on the row-sharded path (round 4): the Bloch-reduced sector problem distributed over two ranks in
    z-slabs (every master/slave pair rank-local), each rank forming ITS rows of NB M BN with the device sparse products over the
    [owned | ghost] columns of the reduced numbering, the non-symmetric reduced operator solved collectively by
    hx_gmres_sharded.  The fixed-point trajectory and the eigenvalue equal the single-GPU Bloch run to 1e-8, the adjoint
    eigenvalue is the conjugate of the direct one to 1e-8 (bloch_operator.py:104-111; Micca/bloch/active.py), the remapped
    eigenvector satisfies p(master) = f_b p(slave) and matches the single-GPU mode up to its sign."""
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_bloch_sharded_worker, args=(world, port, out), nprocs=world, join=True)
    assert len(out) == world
    n_pairs = 0
    for rank in range(world):
        s, d = out[rank]["single"], out[rank]["sharded"]
        assert len(s["hist"]) == len(d["hist"])
        for a, b in zip(s["hist"], d["hist"]):
            assert abs(a - b) / abs(a) < 1e-8, (a, b)
        assert abs(s["eig"] - d["eig"]) / abs(s["eig"]) < 1e-8
        assert abs(d["adj"] - np.conj(d["eig"])) / abs(d["eig"]) < 1e-8, (d["adj"], d["eig"])
        assert d["stats"]["krylov"] == "fgmres" and len(d["stats"]["sharded_levels"]) >= 1
        n_own_red, n_ext_red, n_glob_red = d["n_red"]
        assert n_ext_red > n_own_red and n_glob_red < len(s["p"])
        n_pairs += d["pairs"]
        assert np.max(np.abs(d["pm"] - d["f_b"] * d["ps"])) < 1e-12 * np.max(np.abs(d["ps"]))
        ref = s["p"][d["own_ids"]]
        err = min(np.max(np.abs(d["p_own"] - ref)), np.max(np.abs(d["p_own"] + ref))) / np.max(np.abs(ref))
        assert err < 1e-6, err
    assert n_pairs == 13 * 49                       # every pair of the (12 + 1) x (48 + 1) periodic face is owned by exactly one rank
    assert out[0]["sharded"]["eig"] == out[1]["sharded"]["eig"]


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bloch_sector_row_sharded_two_ranks_200k_rows_each_library_issued_exchanges():
    """Config 5's row-sharded split ABOVE toy size inside the driver-run suite: sector grid (40, 40, 250) ->
    422,091 DOF, two ranks with ~211 k rows each (3,300 SELL-64 slices per slab, two row-sharded hierarchy levels), the exchanges
    issued by the library through the asynchronous librccl stand-in -- the first time hx_gmres_sharded (the reduced operator is
    not complex symmetric) runs on the library's own communicator at a multi-slice size.  Eigenvalue and trajectory of the
    single-GPU Bloch run to 1e-8, adjoint = conjugate to 1e-8, p(master) = f_b p(slave), one stream per communicator."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "tests", "fake_rccl", "libfakerccl.so")):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(root, "tests", "fake_rccl")])
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_bloch_sharded_worker, args=(world, port, out, (40, 40, 250), "fake_rccl"), nprocs=world, join=True)
    assert len(out) == world
    for rank in range(world):
        s, d = out[rank]["single"], out[rank]["sharded"]
        assert len(s["hist"]) == len(d["hist"])
        for a, b in zip(s["hist"], d["hist"]):
            assert abs(a - b) / abs(a) < 1e-8, (a, b)
        assert abs(s["eig"] - d["eig"]) / abs(s["eig"]) < 1e-8
        # (no Robin wall in the sector: the linear branch, whose eigenvalue is omega^2 -- the parity bar is on omega)
        w, wa = np.sqrt(d["eig"]), np.sqrt(d["adj"])
        assert abs(wa - np.conj(w)) / abs(w) < 1e-8, (wa, w)
        assert d["stats"]["krylov"] == "fgmres" and len(d["stats"]["sharded_levels"]) >= 2, d["stats"]
        assert min(lv[0] for lv in d["stats"]["sharded_levels"][:1]) >= 200_000, d["stats"]["sharded_levels"]
        assert "libhx" in d["counters"]["issued_by"] and d["counters"]["communicator_stream_switches"] == 0, d["counters"]
        assert np.max(np.abs(d["pm"] - d["f_b"] * d["ps"])) < 1e-12 * np.max(np.abs(d["ps"]))
    print("bloch 422k sharded:", out[0]["sharded"]["stats"]["sharded_levels"], out[0]["sharded"]["counters"])


def _rcb_bench_worker(rank, world, port, out, workload, transport, partition="rcb"):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "helmholtz-x_amd"), os.path.join(root, "tests")]
    os.environ["HX_QUIET"] = "1"
    if transport == "fake_rccl":
        os.environ["HX_NATIVE_COMM"] = "1"
        os.environ["HX_RCCL_LIB"] = os.path.join(root, "tests", "fake_rccl", "libfakerccl.so")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        import contextlib
        import io
        import bench
        case = bench.build_case(bench.WORKLOADS[workload], resident=(rank == 0))
        if partition == "morton":
            # contiguous ranges of a space-filling-curve numbering: compact blobs, every rank a neighbour of every other one at
            # world 4 (recursive coordinate bisection cuts this ring-shaped geometry into a RING of parts: two neighbours each)
            from helmholtz_x.fem import MeshTags, locality_permutation
            os.environ["HX_PARTITION"] = "rows"
            m0 = case["mesh"]
            m1 = m0.renumbered(locality_permutation(m0.geometry.x))
            case["facet_tags"] = MeshTags(m1, case["facet_tags"].dim, case["facet_tags"].indices, case["facet_tags"].values)
            case["subdomains"] = MeshTags(m1, case["subdomains"].dim, case["subdomains"].indices, case["subdomains"].values)
            from helmholtz_x.fem import Function, FunctionSpace
            c1 = Function(FunctionSpace(m1, "DG", 0), "soundspeed")
            c1.x.array[:] = case["c"].x.array                      # cell-wise field: the cells keep their order
            case["mesh"], case["c"], case["numbering"] = m1, c1, "morton"
            if rank == 0:
                c1.resident()
        res = {}
        with contextlib.redirect_stdout(io.StringIO()):
            if rank == 0:                                   # the single-GPU answer, once
                h1 = []
                E1, _ = bench.hot_path(case, None, nev=4, tol=1e-3, history=h1)
                inner1 = [v for v in E1.inner_cache.values() if hasattr(v, "stats")][0]
                res["single"] = dict(eigs=[E1.getEigenvalue(i) for i in range(2)], hist=list(h1), stats=inner1.stats())
                E1.destroy()
                del E1
                case["c"].release()
                case["mesh"]._assembler = None
                torch.cuda.empty_cache()
            dist.barrier()
            hist = []
            E, _ = bench.hot_path(case, st_type="amg", nev=4, tol=1e-3, history=hist, sharded=True)
        inner = [v for v in E.inner_cache.values() if hasattr(v, "stats")][0]
        sh = case["shard"]
        res["sharded"] = dict(eigs=[E.getEigenvalue(i) for i in range(2)], hist=list(hist), stats=inner.stats(),
                              partition=case["partition"], counters=sh.counters(),
                              peers=(sorted(sh.plan.send_to), sorted(sh.plan.recv_from)))
        out[rank] = res
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("partition", ["rcb", "morton"])
def test_partitions_of_the_refined_real_mesh_four_ranks_library_issued_exchanges(partition):
    """Sharded parity on an UNSTRUCTURED partition inside the driver-run suite: the reference's real MICCA
    mesh refined once (249,497 DOF) cut into four parts by recursive coordinate bisection with reverse Cuthill-McKee inside
    every part (what the N > 1 bench uses for every non-native numbering), ~62 k rows = ~1000 SELL-64 slices per rank, gathered
    (non-contiguous) send lists, ranks with three neighbours -- the exchanges issued by the library through the asynchronous
    librccl stand-in.  Trajectory and eigenvalues of the single-GPU run on the same mesh to 1e-8; one stream per communicator.
    "rcb": the partitioner of the N > 1 bench (this ring-shaped geometry becomes a ring of parts: two neighbours per rank up to
    six ranks); "morton": contiguous ranges of a space-filling-curve numbering -- every rank a neighbour of the three others."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "tests", "fake_rccl", "libfakerccl.so")):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(root, "tests", "fake_rccl")])
    world, port = 4, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_rcb_bench_worker, args=(world, port, out, "micca-real-r1", "fake_rccl", partition), nprocs=world, join=True)
    assert len(out) == world
    single = out[0]["single"]
    rows = 0
    for rank in range(world):
        d = out[rank]["sharded"]
        assert len(d["hist"]) == len(single["hist"])
        for a, b in zip(single["hist"], d["hist"]):
            assert abs(a - b) / abs(a) < 1e-8, (a, b)
        for a, b in zip(single["eigs"], d["eigs"]):
            assert abs(a - b) / abs(a) < 1e-8, (a, b)
        assert "libhx" in d["counters"]["issued_by"] and d["counters"]["communicator_stream_switches"] == 0, d["counters"]
        assert d["partition"]["kind"].startswith("recursive coordinate bisection" if partition == "rcb" else "contiguous row ranges")
        rows += d["partition"]["per_rank"][rank]["rows"]
    per_rank = out[0]["sharded"]["partition"]["per_rank"]
    assert rows == 249497 and min(q["rows"] for q in per_rank) >= 60_000
    assert max(q["neighbours"] for q in per_rank) >= (3 if partition == "morton" else 2), per_rank
    print("rcb x4 micca-real-r1:", per_rank, out[0]["sharded"]["counters"], out[0]["sharded"]["stats"]["iters_per_solve"])


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_config4_annulus_5M_split_over_four_ranks_library_issued_exchanges():
    """BASELINE config 4 at its STATED size inside the driver-run suite: the 5,005,056-DOF annulus cut into four
    theta-slabs (1.25 M rows per rank, two row-sharded hierarchy levels), the solves run inside the library with its own exchanges
    (`hx_comm_*`, here through the asynchronous librccl stand-in: the four ranks share the one GPU of the box), deep-halo cycle as
    by default -- the reference's `mpirun -np 8` run of this case (AnnularCombustor/Micca/fullAnnulus/runAll.sh:9-10) in shape.
    Against the single-GPU run of the same problem in the same process group: trajectory and eigenvalues to 1e-9, the same number
    of inner solves and inner iterations, at most 4.5 halo exchanges per inner iteration (classic schedule: 9.6), one message
    per peer and exchange, no stream switch of the communicator."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "tests", "fake_rccl", "libfakerccl.so")):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(root, "tests", "fake_rccl")])
    world, port = 4, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_rcb_bench_worker, args=(world, port, out, "annulus-5M", "fake_rccl", "slabs"), nprocs=world, join=True)
    assert len(out) == world
    single = out[0]["single"]
    s1 = single["stats"]
    rows = 0
    for rank in range(world):
        d = out[rank]["sharded"]
        assert len(d["hist"]) == len(single["hist"])
        for a, b in zip(single["hist"], d["hist"]):
            assert abs(a - b) / abs(a) < 1e-9, (a, b)
        for a, b in zip(single["eigs"], d["eigs"]):
            assert abs(a - b) / abs(a) < 1e-9, (a, b)
        st, cn = d["stats"], d["counters"]
        inner, inner1 = st["iters_per_solve"] * st["solves"], s1["iters_per_solve"] * s1["solves"]
        # (single GPU: single-precision cycle on a complex-half copy through SELL-64 everywhere; the sharded cycle smooths its ghost
        # layers through the CSR kernel -- another summation order in complex64: the counts agree to a few iterations, not bitwise)
        assert abs(inner - inner1) <= 0.03 * inner1, (inner, inner1, st["solves"], s1["solves"])
        assert "libhx" in cn["issued_by"] and cn["communicator_stream_switches"] == 0, cn
        assert cn["halo_exchanges"] <= 4.5 * inner, (cn, inner)
        assert st["deep_halo_layers"] == 3 and len(st["sharded_levels"]) == 2, st
        rows += d["partition"]["per_rank"][rank]["rows"]
    assert rows == 5005056
    d0 = out[0]["sharded"]
    print("annulus-5M x 4 ranks: %.2f halo exchanges and %.2f all-reduces per inner iteration (%d inner iterations, %d solves), "
          "sharded levels %s, omega %s" % (d0["counters"]["halo_exchanges"] / (d0["stats"]["iters_per_solve"] * d0["stats"]["solves"]),
                                           d0["counters"]["all_reduces"] / (d0["stats"]["iters_per_solve"] * d0["stats"]["solves"]),
                                           round(d0["stats"]["iters_per_solve"] * d0["stats"]["solves"]), d0["stats"]["solves"],
                                           d0["stats"]["sharded_levels"], d0["eigs"][0]))


def _generated_slab_worker(rank, world, port, out):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "helmholtz-x_amd"), os.path.join(root, "tests")]
    os.environ["HX_QUIET"] = "1"
    os.environ["HX_NATIVE_COMM"] = "1"
    os.environ["HX_RCCL_LIB"] = os.path.join(root, "tests", "fake_rccl", "libfakerccl.so")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        import contextlib
        import io
        import bench
        res = {}
        with contextlib.redirect_stdout(io.StringIO()):
            for how in ("cut from the global mesh", "generated"):
                case = bench.build_case(bench.WORKLOADS["annulus-66k"], resident=False, slab_only=(how == "generated"))
                assert (case["mesh"] is None) == (how == "generated")
                hist = []
                E, _ = bench.hot_path(case, st_type="amg", nev=4, tol=1e-3, history=hist, sharded=True)
                res[how] = dict(hist=list(hist), eigs=[E.getEigenvalue(i) for i in range(2)], partition=case["partition"])
                E.destroy()
                case["shard"].destroy()
                dist.barrier()
        out[rank] = res
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_ranks_generate_their_own_slabs_three_ranks():
    """`bench.py --gpus N` on the analytic annulus: every rank generates its theta-slab instead of building the
    global mesh and cutting it.  Three ranks, 64,896 DOF: the fixed-point trajectory and the eigenvalues of the run on generated slabs
    equal those of the run on parts cut from the global mesh BITWISE (same local meshes, test_host_logic: same arithmetic); the
    partition record carries per-rank rows / ghosts / neighbours and the host set-up time."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "tests", "fake_rccl", "libfakerccl.so")):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(root, "tests", "fake_rccl")])
    world, port = 3, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_generated_slab_worker, args=(world, port, out), nprocs=world, join=True)
    assert len(out) == world
    for rank in range(world):
        a, b = out[rank]["cut from the global mesh"], out[rank]["generated"]
        assert a["hist"] == b["hist"] and a["eigs"] == b["eigs"], (a["hist"], b["hist"])
        pa, pb = a["partition"]["per_rank"], b["partition"]["per_rank"]
        assert [(q["rows"], q["ghosts"], q["neighbours"]) for q in pa] == [(q["rows"], q["ghosts"], q["neighbours"]) for q in pb]
        assert all("host_setup_s" in q for q in pb) and "generates its own slab" in b["partition"]["kind"]
    assert sum(q["rows"] for q in out[0]["generated"]["partition"]["per_rank"]) == 64896


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_three_ranks_replicated_direct_factor_library_issued_all_gathers():
    """`bench.py --gpus 3 --st-type lu` (64,896 DOF) with the exchanges issued by the library (hx_comm through the asynchronous librccl
    stand-in): the direct inner solve of the row-sharded path is the REPLICATED device factor, its right-hand sides travel by
    hx_comm_allgather (one group of point-to-point messages on the compute stream, counted with the exchanges); the eigenvalue equals
    the single-GPU direct run's to 1e-9 and no exchange leaves the compute stream."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "tests", "fake_rccl", "libfakerccl.so")):
        subprocess.check_call(["make", "-C", os.path.join(root, "tests", "fake_rccl")])
    env = dict(os.environ, HX_DIST_BACKEND="gloo", HX_NATIVE_COMM="1", HX_RCCL_LIB=os.path.join(root, "tests", "fake_rccl", "libfakerccl.so"))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--workload", "annulus-66k", "--st-type", "lu", "--no-cpu-baseline", "--steps", "1", "--warmup", "0"]
    r3 = subprocess.run(cmd + ["--gpus", "3"], capture_output=True, text=True, timeout=500, env=env)
    assert r3.returncode == 0, r3.stderr[-3000:]
    d3 = json.loads(r3.stdout.strip().splitlines()[-1])
    r1 = subprocess.run(cmd, capture_output=True, text=True, timeout=500, env=dict(os.environ))
    assert r1.returncode == 0, r1.stderr[-3000:]
    d1 = json.loads(r1.stdout.strip().splitlines()[-1])
    assert d3.get("error") is None and d3["n_gpus"] == 3 and "libhx" in d3["exchanges_issued_by"], d3
    st = d3["config"]["inner_stats"]
    assert st["replicated_over"] == 3 and st["rhs_all_gathers"] == st["solves"] > 0, st
    ex = d3["config"]["exchanges_per_step"]
    assert ex["communicator_stream_switches"] == 0 and ex["halo_exchanges"] >= st["rhs_all_gathers"], ex
    for a, b in zip(d3["config"]["omega"], d1["config"]["omega"]):
        assert abs(complex(*a) - complex(*b)) <= 1e-9 * abs(complex(*b)), (a, b)


def _rcb_worker(rank, world, port, out, golden_dir):
    """RCB partition of the reference's real (unstructured) MICCA mesh over `world` ranks: distribute, exchange, check."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "helmholtz-x_amd"), os.path.join(root, "tests")]
    os.environ["HX_QUIET"] = "1"
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        import contextlib
        import io
        from helmholtz_x.distributed import distribute_mesh, partition_quality, partition_rows, rcb_renumbering
        from helmholtz_x.fem import MeshTags
        from helmholtz_x.io_utils import XDMFReader
        from helmholtz_x.sharded import ShardContext
        with contextlib.redirect_stdout(io.StringIO()):
            mesh, sub, ft = XDMFReader(os.path.join(golden_dir, "meshes", "micca_full", "mesh")).getAll()
        n = mesh.num_vertices
        perm, ranges = rcb_renumbering(mesh, world)
        assert np.array_equal(np.sort(perm), np.arange(n)) and ranges[0][0] == 0 and ranges[-1][1] == n
        m2 = mesh.renumbered(perm)
        ft2 = MeshTags(m2, ft.dim, ft.indices, ft.values)
        sub2 = MeshTags(m2, sub.dim, sub.indices, sub.values)
        local, sub_l, ft_l, _ = distribute_mesh(m2, ranges, rank, sub2, ft2, [])
        part = local.part
        n_own, l2g = part["n_own"], np.asarray(part["local_to_global"])
        r0, r1 = ranges[rank]
        assert n_own == r1 - r0 and abs(n_own - n / world) <= 1
        ctx = ShardContext(n, ranges=ranges)
        ctx.set_ghosts(l2g[n_own:])
        v = torch.zeros(ctx.n_ext, dtype=torch.float64)
        v[:n_own] = torch.from_numpy(m2.geometry.x[r0:r1, 2].copy())          # the owners' z coordinates travel to the ghosts
        ctx.halo(v)
        assert np.array_equal(v[n_own:].numpy(), m2.geometry.x[l2g[n_own:], 2])
        owned_cells = torch.tensor([float(part["cell_owned"].sum())], dtype=torch.float64)
        dist.all_reduce(owned_cells)
        assert int(owned_cells[0]) == mesh.num_cells
        q_rcb = partition_quality(m2.topology.cells, ranges)
        assert q_rcb[rank]["ghosts"] == l2g.size - n_own                      # the plan's ghost block IS the partition's cut
        m3 = mesh.locality_renumbered("rcm")
        q_rows = partition_quality(m3.topology.cells, partition_rows(n, world))
        out[rank] = (q_rcb, q_rows, len(ctx.plan.recv_from))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_rcb_partition_of_the_real_micca_mesh_over_eight_ranks(golden_dir):
    """a geometric partitioner for unstructured meshes (METIS is not installed; the reference partitions through
    DOLFINx, io_utils.py:149-155).  Recursive coordinate bisection + reverse Cuthill-McKee inside the parts on the reference's
    real MICCA mesh, world 8 (gloo, CPU): balanced parts, every cell owned once, the exchange plan built from the local mesh
    delivers the owners' values to the ghosts -- and the cut is less than half of what contiguous strips of the global RCM
    numbering (the round-3 partition) give: 4.4 k against 10.6 k ghost rows in total, at 5 instead of 2 neighbours per rank
    (point-to-point messages to different peers travel over different xGMI links)."""
    world, port = 8, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_rcb_worker, args=(world, port, out, golden_dir), nprocs=world, join=True)
    assert len(out) == world
    q_rcb, q_rows, _ = out[0]
    g_rcb, g_rows = sum(d["ghosts"] for d in q_rcb), sum(d["ghosts"] for d in q_rows)
    assert g_rcb < 0.5 * g_rows, (g_rcb, g_rows)
    assert max(d["neighbours"] for d in q_rcb) <= 7               # (one xGMI link per peer on an 8-GPU node)
    for rank in range(world):
        assert out[rank][2] == q_rcb[rank]["neighbours"]
    assert max(d["rows"] for d in q_rcb) - min(d["rows"] for d in q_rcb) <= 1


def _outstanding_work_worker(rank, world, port, out, gpu):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "helmholtz-x_amd")]
    if gpu:
        os.environ["HX_NATIVE_COMM"] = "1"
        os.environ["HX_RCCL_LIB"] = os.path.join(root, "tests", "fake_rccl", "libfakerccl.so")
        os.environ["HX_FAKE_RCCL_TIMEOUT_S"] = "4"
        torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        from helmholtz_x import distributed as dd
        peer = 1 - rank
        send, recv = torch.full((4,), float(rank)), torch.empty(4)
        reqs = dd.issue([dist.P2POp(dist.isend, send, peer), dist.P2POp(dist.irecv, recv, peer)])
        res = {}
        if gpu:
            from helmholtz_x.sharded import ShardContext
            ctx = ShardContext(1000)
            assert ctx.comm is not None
            t = torch.ones(2, dtype=torch.float64, device="cuda")
            for name, call in (("all_reduce", lambda: ctx.all_reduce_(t)),):
                try:
                    call()
                    res[name] = "no error"
                except RuntimeError as e:
                    res[name] = str(e)
        else:
            try:
                dd.assert_no_outstanding_work("test")
                res["assert"] = "no error"
            except RuntimeError as e:
                res["assert"] = str(e)
        dd.wait_all(reqs)
        dd.assert_no_outstanding_work("after wait_all")          # the registry is empty again
        res["recv"] = recv.tolist()
        if gpu:
            ctx.all_reduce_(t)                                    # and the library's communicator may be used
            torch.cuda.synchronize()
            res["sum"] = t.tolist()
            # the stand-in's barrier has a deadline: rank 1 skips the next all-reduce; rank 0's operation must FAIL (not hang)
            if rank == 0:
                import time
                t0 = time.perf_counter()
                try:
                    ctx.all_reduce_(t)
                    torch.cuda.synchronize()
                    ctx.all_reduce_(t)                            # every later operation of the failed communicator reports it
                    res["timeout"] = "no error"
                except Exception as e:
                    res["timeout"] = "%s after %.1f s" % (type(e).__name__, time.perf_counter() - t0)
            dist.barrier()
        out[rank] = res
    finally:
        dist.destroy_process_group()


def test_outstanding_torch_request_is_caught_before_the_library_communicator_is_used():
    """Two NCCL communicators live in a rank under the nccl backend (torch's process group for the set-up, the library's hx_comm
    for the solves); NCCL does not order them.  Every point-to-point request of the set-up goes through distributed.issue /
    wait_all, and a request left un-waited makes the next entry into hx_comm raise.  CPU half: the registry itself (gloo)."""
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_outstanding_work_worker, args=(world, port, out, False), nprocs=world, join=True)
    for rank in range(world):
        assert "still outstanding" in out[rank]["assert"], out[rank]
        assert out[rank]["recv"] == [float(1 - rank)] * 4


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_outstanding_request_guard_and_stand_in_deadline_two_ranks_one_gpu():
    """GPU half (two ranks on the one GPU, exchanges issued by the library through tests/fake_rccl): hx_comm_allreduce refuses to
    start while a batch_isend_irecv request of the set-up is un-waited and works once it was waited for; and the stand-in's
    cross-process barrier gives up after its deadline when a peer never enqueues the matching operation (it used to
    spin for ever inside a host function, blocking the stream, hipDeviceSynchronize and the pool behind it)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "tests", "fake_rccl", "libfakerccl.so")):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(root, "tests", "fake_rccl")])
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_outstanding_work_worker, args=(world, port, out, True), nprocs=world, join=True)
    for rank in range(world):
        assert "still outstanding" in out[rank]["all_reduce"], out[rank]
        assert out[rank]["sum"] == [2.0, 2.0]
    assert out[0]["timeout"] != "no error" and "after" in out[0]["timeout"], out[0]


def _sharded_ladder_worker(rank, world, port, out):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "helmholtz-x_amd"), os.path.join(root, "tests")]
    os.environ["HX_QUIET"] = "1"
    os.environ["HX_NATIVE_COMM"] = "1"
    os.environ["HX_RCCL_LIB"] = os.path.join(root, "tests", "fake_rccl", "libfakerccl.so")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        import contextlib
        import io
        import bench
        from helmholtz_x.petsc import Options
        Options._db = {}
        case = bench.build_case((10, 192, 11))
        res = {}
        for mode, max_it in (("free", None), ("limited", "auto")):
            if max_it == "auto":                          # half of what the hardest solve of the unrestricted run needed
                Options()["st_ksp_max_it"] = max(3, res["free"]["stats"]["max_iters_in_a_solve"] // 2)
            hist = []
            with contextlib.redirect_stdout(io.StringIO()):
                E, _ = bench.hot_path(case, st_type="amg", nev=4, tol=1e-3, history=hist, sharded=True)
            inner = [v for v in E.inner_cache.values() if hasattr(v, "stats")][0]
            res[mode] = dict(eigs=[E.getEigenvalue(i) for i in range(2)], hist=list(hist), stats=inner.stats())
            E.destroy()
        res["counters"] = case["shard"].counters()
        out[rank] = res
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_escalation_ladder_on_the_row_sharded_path_two_ranks():
    """The ladder's decisions rest on all-reduced residual norms, so every rank takes the same step at the same solve: two ranks
    (library-issued exchanges through the stand-in), iteration limit of the first rung cut to half of what the hardest solve needs
    (-st_ksp_max_it) -> both ranks move to the wide rung after the same solve, rebuild their slab of the hierarchy, the recycling
    workspace and the level plans, and return the eigenvalues of the unrestricted run."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "tests", "fake_rccl", "libfakerccl.so")):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(root, "tests", "fake_rccl")])
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_sharded_ladder_worker, args=(world, port, out), nprocs=world, join=True)
    assert len(out) == world
    for rank in range(world):
        free, lim = out[rank]["free"], out[rank]["limited"]
        assert free["stats"]["rung"] == "default" and free["stats"]["ladder"] == []
        assert lim["stats"]["rung"] == "wide" and "stalled" in lim["stats"]["ladder"][0]["reason"], lim["stats"]
        assert len(lim["stats"]["sharded_levels"]) >= 1
        for a, b in zip(free["eigs"], lim["eigs"]):
            assert abs(a - b) / abs(a) < 1e-8, (a, b)
        assert out[rank]["counters"]["communicator_stream_switches"] == 0
    assert out[0]["limited"]["stats"]["ladder"] == out[1]["limited"]["stats"]["ladder"]       # same step, same solve, same reason


def _deep_plan_worker(rank, world, port, out):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "helmholtz-x_amd")]
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        from helmholtz_x.distributed import DeepHaloPlan, HaloPlan, partition_rows
        n = 40
        ranges = partition_rows(n, world, 1)
        r0, r1 = ranges[rank]
        # a periodic band graph of half-width 2: layer 1 = ids within 2 of my range, layer 2 = the next 2 on each side
        near = lambda lo, hi: np.unique(np.concatenate([np.arange(r0 - hi, r0 - lo + 1), np.arange(r1 - 1 + lo, r1 + hi)]) % n)
        mine = np.arange(r0, r1)
        g1 = np.setdiff1d(near(1, 2), mine)
        g2 = np.setdiff1d(np.setdiff1d(near(3, 4), mine), g1)
        g3 = np.setdiff1d(np.setdiff1d(np.setdiff1d(near(5, 6), mine), g1), g2)
        deep = DeepHaloPlan([HaloPlan(ranges, rank, g1), HaloPlan(ranges, rank, g2), HaloPlan(ranges, rank, g3)])
        n_own = r1 - r0
        g2 = np.concatenate([g2, g3])                      # (layers 2 and 3 are checked together below)
        x = torch.full((n_own + len(g1) + len(g2),), -1.0, dtype=torch.float64)
        x[:n_own] = torch.arange(r0, r1, dtype=torch.float64)
        deep.exchange(x, n_own)
        want = np.concatenate([mine, g1, g2]).astype(np.float64)
        z = torch.full((n_own + len(g1) + len(g2),), -1.0, dtype=torch.complex128)
        z[:n_own] = torch.arange(r0, r1, dtype=torch.float64) * (1 + 2j)
        deep.exchange(z, n_own)
        sp, sptr, sidx, rp_, roff, rcnt = deep.native_arrays()
        out[rank] = dict(ok=bool(np.array_equal(x.numpy(), want)), okz=bool(np.allclose(z.numpy(), want * (1 + 2j))),
                         sends=[(int(p), int(sptr[q + 1] - sptr[q])) for q, p in enumerate(sp)],
                         recvs=[(int(p), int(c), int(o)) for p, c, o in zip(rp_, rcnt, roff)], n=(n_own, len(g1), len(g2)),
                         exchanges=deep.n_exchanges)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_deep_halo_plan_two_layers_in_one_exchange(world):
    """DeepHaloPlan (the multi-layer exchange of the deep-halo V-cycle): three ghost layers of every rank filled by ONE batch of
    point-to-point messages, real and complex entries; the host arrays for hx_comm_set_plan list per peer one message per layer, in
    layer order, and what one rank sends to a peer is what that peer expects from it, message by message."""
    port = _free_port()
    out = mp.Manager().dict()
    mp.spawn(_deep_plan_worker, args=(world, port, out), nprocs=world, join=True)
    assert len(out) == world
    for rank in range(world):
        r = out[rank]
        assert r["ok"] and r["okz"] and r["exchanges"] == 2, r
        n_own, n1, n2 = r["n"]
        assert n1 > 0 and n2 > 0
        # receive ranges tile the two ghost blocks
        assert sorted(o for _, _, o in r["recvs"])[0] == n_own and sum(c for _, c, _ in r["recvs"]) == n1 + n2
        for peer in range(world):
            if peer == rank:
                continue
            mine = [c for p, c in r["sends"] if p == peer]                      # my messages to the peer, in issue order
            theirs = [c for p, c, _ in out[peer]["recvs"] if p == rank]          # what the peer expects from me, in issue order
            assert mine == theirs, (rank, peer, mine, theirs)
