"""CPU tests of the multi-GPU layer under the gloo backend (world size 2 and 3): node partition, halo maps,
grouped point-to-point exchange and the two-reduction distributed CG, with the oracle doing each rank's local
arithmetic.  The distributed result must reproduce the serial Eigen-order CG of the oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from polyfem_amd import distributed as D
from polyfem_amd import mesh as M


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pf_oracle as O
        from tests.dist_helpers import OracleOps, csc_tuple_to_csr
        slab = None
        if case == "slab_p2":
            # the benchmark's decomposition: n x n x (cells_per_rank * world) cells, one slab of node planes per rank
            kind, conn, gconn, xyz, n_nodes, L = D.cube_tets_slab(2, world, 0, world, 2)
            h = 0.5 / 2
            g = np.arange(n_nodes)
            mesh = M.Mesh(kind, M.GEOM_KIND[kind], conn.astype(np.int32), gconn.astype(np.int32), xyz, n_nodes,
                          np.stack([g % L, (g // L) % L, g // (L * L)], axis=1) * h)
            model, bs = O.LINEAR_ELASTICITY, 3
            slab = D.slab_part(2, 1, rank, world, 2)
        elif case == "morton_p2":
            base = M.cube_tets(4, 2, jitter=0.1)
            mesh, model, bs = D.renumber_nodes(base, D.morton_numbering(base.node_xyz)), O.LINEAR_ELASTICITY, 3
        elif case == "laplace_p1":
            mesh, model, bs = M.cube_tets(5, 1), O.LAPLACIAN, 1
        elif case == "elastic_q2":
            mesh, model, bs = M.cube_hexes(2, 2), O.LINEAR_ELASTICITY, 3
        elif case == "elastic_tri_p2":
            mesh, model, bs = M.square_tris(6, 2, jitter=0.1), O.LINEAR_ELASTICITY, 2
        else:
            mesh, model, bs = M.cube_tets(3, 2, jitter=0.1), O.LINEAR_ELASTICITY, 3
        lam, mu = O.lame(1e5, 0.3, M.KIND_DIM[mesh.kind])
        bounds = D.owner_ranges(mesh.n_nodes, world)
        part = D.partition(mesh.kind, mesh.conn, mesh.gconn, mesh.xyz, bounds, rank, mesh.node_xyz)
        if slab is not None:
            part, bounds, nn = slab
            assert nn == mesh.n_nodes
        l2g = part.local_to_global()
        nl = part.n_local
        # --- partition invariants
        assert part.n_owned == bounds[rank + 1] - bounds[rank]
        assert np.all(np.diff(part.halo_global) > 0) and np.all(part.halo_owner != rank)
        counts = [None] * world
        dist.all_gather_object(counts, int(part.elem_owned.sum()))
        assert sum(counts) == mesh.n_el, "every element's energy must be owned by exactly one rank"
        # --- local assembly of the local mesh == owned rows of the global matrix
        n_glob = mesh.n_nodes * bs
        Kg = csc_tuple_to_csr(O.assemble_linear(mesh, model, lam, mu), n_glob)
        Kl = csc_tuple_to_csr(O.assemble_linear(part.mesh, model, lam, mu), nl * bs)
        dof_l2g = (l2g[:, None] * bs + np.arange(bs)[None]).reshape(-1)
        no = part.n_owned * bs
        ref_rows = Kg[dof_l2g[:no]][:, dof_l2g]
        assert abs(Kl[:no] - ref_rows).max() <= 1e-12 * abs(Kg).max()
        # --- distributed CG == serial CG
        bd = M.boundary_dofs(mesh, bs, "all" if bs == 1 else "x0")
        fixed_g = np.zeros(n_glob, dtype=bool)
        fixed_g[bd] = True
        b_g = np.random.default_rng(3).standard_normal(n_glob)
        ops = OracleOps(Kl, no, fixed_g[dof_l2g])
        comm = D.Comm()
        cg = D.DistributedCG(ops, part, bs, comm, torch.device("cpu"))
        b = torch.from_numpy(b_g[dof_l2g].copy())
        x = torch.zeros(nl * bs, dtype=torch.float64)
        iters, err = cg.solve(b, x, 1, 1000, 1e-12, check_every=7)
        xs = [None] * world
        dist.all_gather_object(xs, (int(part.lo), x.numpy()[:no].copy()))
        # halo exchange check: after one more exchange the halo of x equals the owners' values
        cg.halo(x.view(nl, bs))
        if rank == 0:
            xg = np.zeros(n_glob)
            for lo, v in xs:
                xg[lo * bs: lo * bs + v.size] = v
            keep = np.setdiff1d(np.arange(n_glob), bd)
            red = O.full_to_reduced_matrix(O.assemble_linear(mesh, model, lam, mu), bd)
            xr, it_ref, err_ref = O.cg(red, b_g[keep], jacobi=True, max_iter=1000, tol=1e-12)
            ret["iters"], ret["it_ref"], ret["err"] = iters, it_ref, err
            ret["dx"] = float(np.abs(xg[keep] - xr).max() / np.abs(xr).max())
            ret["fixed_zero"] = bool(not xg[bd].any())
        allx = [None] * world
        dist.all_gather_object(allx, None)
        xg_all = np.zeros(n_glob)
        for lo, v in xs:
            xg_all[lo * bs: lo * bs + v.size] = v
        assert np.array_equal(x.numpy(), xg_all[dof_l2g]), "halo entries must equal the owners' values"
        ret[f"ok{rank}"] = True
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,case", [(2, "laplace_p1"), (2, "elastic_p2"), (3, "elastic_p2"), (2, "elastic_q2"), (4, "laplace_p1"),
                                        (8, "slab_p2"), (8, "morton_p2"), (3, "elastic_tri_p2")])
def test_distributed_cg_gloo(world, case):
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, case, ret), nprocs=world, join=True)
        assert all(ret.get(f"ok{r}") for r in range(world))
        assert abs(ret["iters"] - ret["it_ref"]) <= 2, (ret["iters"], ret["it_ref"])
        assert ret["err"] < 1e-11 and ret["dx"] < 1e-8 and ret["fixed_zero"]


def test_morton_blocks_have_smaller_halos_than_index_slabs():
    """Eight owner ranges of a Morton renumbering are compact blocks: fewer halo nodes in total than eight slabs of the
    lattice numbering, and the renumbered mesh is the same mesh (same sorted element node sets up to the renaming)."""
    mesh = M.cube_tets(8, 2)
    new_id = D.morton_numbering(mesh.node_xyz)
    assert np.array_equal(np.sort(new_id), np.arange(mesh.n_nodes))
    mm = D.renumber_nodes(mesh, new_id)
    assert np.array_equal(mm.conn, new_id[mesh.conn]) and np.allclose(mm.node_xyz[new_id], mesh.node_xyz)
    bounds = D.owner_ranges(mesh.n_nodes, 8)
    h_slab = D.halo_nodes_total(mesh.kind, mesh.conn, mesh.gconn, mesh.xyz, bounds)
    h_morton = D.halo_nodes_total(mm.kind, mm.conn, mm.gconn, mm.xyz, bounds)
    assert h_morton < 0.8 * h_slab, (h_morton, h_slab)


def test_slab_partition_matches_generic():
    """The benchmark's direct slab generator gives the same local parts as partitioning the full mesh."""
    n, cpr, world = 3, 2, 3
    nz = cpr * world
    kind, conn, gconn, xyz, n_nodes, L = D.cube_tets_slab(n, nz, 0, nz, 2)
    step_bounds = None
    for rank in range(world):
        part, bounds, nn = D.slab_part(n, cpr, rank, world, 2)
        assert nn == n_nodes
        ref = D.partition(kind, conn, gconn, xyz, bounds, rank)
        assert part.n_owned == ref.n_owned and np.array_equal(part.halo_global, ref.halo_global)
        assert part.mesh.n_el == ref.mesh.n_el
        a = np.sort(part.local_to_global()[part.mesh.conn], axis=1)
        b = np.sort(ref.local_to_global()[ref.mesh.conn], axis=1)
        assert np.array_equal(a[np.lexsort(a.T[::-1])], b[np.lexsort(b.T[::-1])])
        assert part.elem_owned.sum() == ref.elem_owned.sum()
        step_bounds = bounds
    assert step_bounds[-1] == n_nodes


def _newton_worker(rank, world, port, case, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pf_oracle as O
        from tests.dist_helpers import OracleNLOps
        mesh = M.cube_tets(3, 1, jitter=0.1)
        lam, mu = O.lame(2e4, 0.3)
        n_glob = mesh.n_nodes * 3
        bd = M.boundary_dofs(mesh, 3, "x0")
        rhs_g = np.zeros(n_glob)
        if case == "soft":
            rhs_g[2::3] = -50.0 / mesh.n_nodes      # full Newton steps
        else:
            rhs_g[0::3] = -2e4 / mesh.n_nodes       # compression: the line search has to backtrack
        bounds = D.owner_ranges(mesh.n_nodes, world)
        part = D.partition(mesh.kind, mesh.conn, mesh.gconn, mesh.xyz, bounds, rank, mesh.node_xyz)
        l2g = part.local_to_global()
        dof_l2g = (l2g[:, None] * 3 + np.arange(3)[None]).reshape(-1)
        fixed_g = np.zeros(n_glob, dtype=bool)
        fixed_g[bd] = True
        fixed_l = np.nonzero(fixed_g[dof_l2g])[0]
        ops = OracleNLOps(part, O.NEOHOOKEAN, lam, mu, fixed_g[dof_l2g])
        newton = D.DistributedNewton(ops, part, 3, D.Comm(), torch.device("cpu"), fixed_l)
        x = torch.zeros(part.n_local * 3, dtype=torch.float64)
        info = newton.solve(x, torch.from_numpy(rhs_g[dof_l2g].copy()), cg_check_every=9)
        no = part.n_owned * 3
        xs = [None] * world
        dist.all_gather_object(xs, (int(part.lo), x.numpy()[:no].copy()))
        xg = np.zeros(n_glob)
        for lo, v in xs:
            xg[lo * 3: lo * 3 + v.size] = v
        assert np.array_equal(x.numpy(), xg[dof_l2g]), "the halo of the iterate must be current on return"
        infos = [None] * world
        dist.all_gather_object(infos, {k: info[k] for k in ("status", "iterations", "line_search_iterations", "cg_iterations")})
        assert all(i == infos[0] for i in infos), "all ranks must take the same decisions"
        if rank == 0:
            xr, iref = O.newton_solve(mesh, O.NEOHOOKEAN, lam, mu, rhs_g, bd, np.zeros(n_glob))
            ret["info"], ret["iref"] = dict(info), iref
            ret["dx"] = float(np.abs(xg - xr).max() / np.abs(xr).max())
            ret["hessians"] = ops.hessians
        # not started: zero load at rest
        x.zero_()
        info = newton.solve(x, None)
        assert info["status"] == D.NOT_STARTED and info["iterations"] == 0
        # iteration limit -> the reference throws
        x.zero_()
        try:
            newton.solve(x, torch.from_numpy(rhs_g[dof_l2g].copy()), max_iterations=1, f_delta=0.0, grad_norm=1e-300)
            raised = False
        except D.NewtonFailure as e:
            raised = e.info["status"] == D.ITERATION_LIMIT
        assert raised
        ret[f"ok{rank}"] = True
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,case", [(2, "soft"), (2, "backtrack"), (3, "backtrack")])
def test_distributed_newton_gloo(world, case):
    """DistributedNewton over 2 and 3 gloo ranks (oracle arithmetic per rank) against the serial restatement of the
    reference's Newton solver: same status, iteration and line-search counts, same minimiser."""
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    mp.spawn(_newton_worker, args=(world, _free_port(), case, ret), nprocs=world, join=True)
    assert all(ret.get(f"ok{r}") for r in range(world))
    info, iref = ret["info"], ret["iref"]
    assert info["status"] == iref["status"] and info["status"] in (D.F_DELTA, D.GRAD_NORM)
    assert info["iterations"] == iref["iterations"]
    assert info["line_search_iterations"] == iref["line_search_iterations"]
    if case == "backtrack":
        assert info["line_search_iterations"] > 3 * info["iterations"]
    assert abs(info["cg_iterations"] - iref["cg_iterations"]) <= max(3 * info["iterations"], 0.03 * iref["cg_iterations"])
    assert abs(info["energy"] - iref["energy"]) <= 1e-9 * abs(iref["energy"])
    assert ret["dx"] <= 1e-8


def _euler_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pf_oracle as O
        from tests.dist_helpers import OracleNLOps
        mesh = M.cube_tets(3, 1, jitter=0.1)
        lam, mu = O.lame(2e4, 0.3)
        n_glob = mesh.n_nodes * 3
        bd = M.boundary_dofs(mesh, 3, "x0")
        rho, dt = 1000.0, 0.02
        rhs_g = O.assemble_rhs(mesh, 3, np.array([0.0, 3.0, -9.81]), rho)
        bounds = D.owner_ranges(mesh.n_nodes, world)
        part = D.partition(mesh.kind, mesh.conn, mesh.gconn, mesh.xyz, bounds, rank, mesh.node_xyz)
        l2g = part.local_to_global()
        dof_l2g = (l2g[:, None] * 3 + np.arange(3)[None]).reshape(-1)
        fixed_g = np.zeros(n_glob, dtype=bool)
        fixed_g[bd] = True
        ops = OracleNLOps(part, O.NEOHOOKEAN, lam, mu, fixed_g[dof_l2g])
        ops.set_mass(rho)
        newton = D.DistributedNewton(ops, part, 3, D.Comm(), torch.device("cpu"), np.nonzero(fixed_g[dof_l2g])[0])
        stepper = D.DistributedImplicitEuler(newton, dt)
        v0_g = np.where(fixed_g, 0.0, 0.05 * np.random.default_rng(9).standard_normal(n_glob))
        x = torch.zeros(part.n_local * 3, dtype=torch.float64)
        v = torch.from_numpy(v0_g[dof_l2g].copy())
        a = torch.zeros_like(x)
        rhs = torch.from_numpy(rhs_g[dof_l2g].copy())
        infos = [stepper.step(x, v, a, rhs, cg_check_every=11) for _ in range(2)]
        no = part.n_owned * 3
        out = [None] * world
        dist.all_gather_object(out, (int(part.lo), x.numpy()[:no].copy(), v.numpy()[:no].copy(), a.numpy()[:no].copy()))
        xg, vg, ag = np.zeros(n_glob), np.zeros(n_glob), np.zeros(n_glob)
        for lo, xv, vv, av in out:
            xg[lo * 3: lo * 3 + xv.size], vg[lo * 3: lo * 3 + vv.size], ag[lo * 3: lo * 3 + av.size] = xv, vv, av
        assert np.array_equal(x.numpy(), xg[dof_l2g]) and np.array_equal(v.numpy(), vg[dof_l2g])
        if rank == 0:
            mass = O.assemble_mass(mesh, 3, rho)
            xo, vo, irefs = np.zeros(n_glob), v0_g.copy(), []
            for _ in range(2):
                xo, vo, ao, iref = O.implicit_euler_step(mesh, O.NEOHOOKEAN, lam, mu, rhs_g, bd, mass, dt, xo, vo)
                irefs.append(iref)
            ret["ok_counts"] = all(i["iterations"] == r["iterations"] and i["status"] == r["status"]
                                   and i["line_search_iterations"] == r["line_search_iterations"] for i, r in zip(infos, irefs))
            ret["dx"] = float(np.abs(xg - xo).max() / np.abs(xo).max())
            ret["dv"] = float(np.abs(vg - vo).max() / np.abs(vo).max())
            ret["da"] = float(np.abs(ag - ao).max() / np.abs(ao).max())
        ret[f"ok{rank}"] = True
    finally:
        dist.destroy_process_group()


def test_distributed_implicit_euler_gloo():
    """Two implicit Euler steps over 2 gloo ranks (oracle arithmetic per rank) against the serial restatement."""
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    mp.spawn(_euler_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert all(ret.get(f"ok{r}") for r in range(world))
    assert ret["ok_counts"] and ret["dx"] <= 1e-8 and ret["dv"] <= 1e-7 and ret["da"] <= 1e-6


def test_hex_slab_partition_covers_the_global_mesh():
    """The hexahedral slab decomposition of the benchmark: every global element is energy-owned by exactly one rank,
    owned node ranges tile the lattice, and local connectivity maps back to the global lattice numbering."""
    from polyfem_amd import distributed as D
    from polyfem_amd import mesh as M
    n, world = 3, 2
    ref = M.cube_hexes(n, 2)  # one rank's worth; the global grid is n x n x (n*world)
    owned_total, seen = 0, []
    for rank in range(world):
        part, bounds, n_nodes = D.slab_part(n, n, rank, world, 2, "hex")
        assert part.mesh.kind == M.Q2HEX and part.mesh.conn.shape[1] == 27
        L = 2 * n + 1
        assert n_nodes == L * L * (2 * n * world + 1)
        assert part.lo == bounds[rank] and part.n_owned == bounds[rank + 1] - bounds[rank]
        l2g = part.local_to_global()
        gl = l2g[part.mesh.conn]
        own = part.elem_owned.astype(bool)
        owned_total += int(own.sum())
        seen.append(np.sort(gl[own], axis=1))
        # node coordinates follow the lattice
        assert np.allclose(part.mesh.node_xyz[:, 2], (l2g // (L * L)) * 0.5 / n)
    assert owned_total == n * n * n * world
    allc = np.concatenate(seen)
    assert np.unique(allc, axis=0).shape[0] == allc.shape[0]
    if world == 1:
        assert ref.n_el == owned_total


def _coarse_worker(rank, world, port, bs, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pf_oracle as O
        from tests.dist_helpers import OracleOps, csc_tuple_to_csr
        model = O.LINEAR_ELASTICITY if bs >= 2 else O.LAPLACIAN
        lam, mu = O.lame(1e5, 0.3)
        if bs == 2:
            # a strip of 2 x (4 * world) cells, P2 triangles, lattice numbering (x fastest): the ranks own bands in y
            strip = M.square_tris(4 * world, 2)
            keep_el = M.node_positions(strip)[strip.conn[:, :3]][:, :, 0].max(axis=1) <= 2.0 / (4 * world) + 1e-12
            conn = strip.conn[keep_el]
            used = np.unique(conn)
            new_id = np.full(strip.n_nodes, -1, dtype=np.int64)
            new_id[used] = np.arange(used.size)
            pos = M.node_positions(strip)[used]
            lam, mu = O.lame(1e5, 0.3, 2)
            n_nodes = int(used.size)
            bounds = D.owner_ranges(n_nodes, world)
            part = D.partition(strip.kind, new_id[conn].astype(np.int32), strip.gconn[keep_el], strip.xyz, bounds, rank, node_xyz=pos)
        else:
            part, bounds, n_nodes = D.slab_part(2, 2, rank, world, 2)  # a bar of 2 x 2 x (2 * world) cells, P2 tets
        nl, no = part.n_local, part.n_owned * bs
        l2g = part.local_to_global()
        dof_l2g = (l2g[:, None] * bs + np.arange(bs)[None]).reshape(-1)
        Kl = csc_tuple_to_csr(O.assemble_linear(part.mesh, model, lam, mu), nl * bs)
        fixed_nodes = np.nonzero(np.abs(part.mesh.node_xyz[:, -1]) < 1e-12)[0]  # clamped at the first rank's outer face (z = 0; y = 0 for the strip)
        fixed_dofs = np.sort((fixed_nodes[:, None] * bs + np.arange(bs)[None]).reshape(-1))
        fixed_l = np.zeros(nl * bs, dtype=bool)
        fixed_l[fixed_dofs] = True
        b_g = np.random.default_rng(9).standard_normal(n_nodes * bs)
        comm = D.Comm()
        cg = D.DistributedCG(OracleOps(Kl, no, fixed_l), part, bs, comm, torch.device("cpu"))
        b = torch.from_numpy(b_g[dof_l2g].copy())
        x0 = torch.zeros(nl * bs, dtype=torch.float64)
        it0, err0 = cg.solve(b, x0, 2, 500, 1e-10)
        cg.enable_coarse_space(M.node_positions(part.mesh), fixed_dofs)
        x1 = torch.zeros(nl * bs, dtype=torch.float64)
        it1, err1 = cg.solve(b, x1, 2, 500, 1e-10)
        # the coarse matrix is symmetric and the coarse correction reproduces a rigid-body mode of a floating rank
        c = cg.cs
        m = c["modes"]
        Einv = c["einv"]
        assert float((Einv - Einv.T).abs().max()) <= 1e-9 * float(Einv.abs().max())
        rows = [None] * world
        dist.all_gather_object(rows, (int(part.lo), x0.numpy()[:no].copy(), x1.numpy()[:no].copy(), Kl[:no], dof_l2g, fixed_l[:no]))
        if rank == 0:
            n = n_nodes * bs
            xg0, xg1, fx = np.zeros(n), np.zeros(n), np.zeros(n, dtype=bool)
            A = None
            import scipy.sparse as sp
            blocks = []
            for lo, v0, v1, K, l2g_r, f in rows:
                xg0[lo * bs: lo * bs + v0.size] = v0
                xg1[lo * bs: lo * bs + v1.size] = v1
                fx[lo * bs: lo * bs + f.size] = f
                coo = K.tocoo()
                blocks.append(sp.csr_matrix((coo.data, (coo.row + lo * bs, l2g_r[coo.col])), shape=(n, n)))
            A = sum(blocks[1:], blocks[0]).tocsr()
            Dm = sp.diags((~fx).astype(float))
            Am = (Dm @ A @ Dm + sp.diags(fx.astype(float))).tocsr()
            bm = np.where(fx, 0.0, b_g)
            import scipy.sparse.linalg as spla
            xr = spla.spsolve(Am.tocsc(), bm)
            ret["it0"], ret["it1"], ret["err0"], ret["err1"] = it0, it1, err0, err1
            ret["d0"] = float(np.abs(xg0 - xr).max() / np.abs(xr).max())
            ret["d1"] = float(np.abs(xg1 - xr).max() / np.abs(xr).max())
        ret[f"ok{rank}"] = True
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,bs", [(4, 3), (6, 3), (4, 1), (4, 2)])
def test_coarse_space_over_the_ranks(world, bs):
    """DistributedCG's preconditioned loop (precond = 2) with exact subdomain solves standing in for the per-rank AMG:
    block-Jacobi alone and with the coarse space of the ranks' rigid-body modes (constants for a scalar problem) both
    reach the direct solution of the whole system; the coarse space needs fewer iterations on a bar of slabs clamped at
    one end (the slabs away from the clamp float)."""
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_coarse_worker, args=(world, _free_port(), bs, ret), nprocs=world, join=True)
        assert all(ret.get(f"ok{r}") for r in range(world))
        it0, it1, err0, err1, d0, d1 = (ret[k] for k in ("it0", "it1", "err0", "err1", "d0", "d1"))
    assert err0 <= 1e-10 and err1 <= 1e-10 and d0 <= 1e-7 and d1 <= 1e-7, (err0, err1, d0, d1)
    assert it1 < it0, (it0, it1)
