"""GPU test of the multi-GPU path on ONE card: two ranks share cuda:0 and talk through gloo (device tensors are
staged through the host by polyfem_amd.distributed.Comm), so the real HIP step kernels, owned-row ranges, halo
exchange and scalar all-reduces are exercised end to end.  Reference: the single-engine solve of the whole mesh."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from polyfem_amd import distributed as D
        from polyfem_amd import engine as E
        from polyfem_amd import mesh as M
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        stream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(stream)
        n, cpr = 6, 3
        part, bounds, n_nodes = D.slab_part(n, cpr, rank, world, 2)
        lam, mu = 2e4 * 0.3 / (1.3 * 0.4), 2e4 / 2.6
        eng = E.Engine(0)
        eng.set_stream(stream.cuda_stream)
        eng.set_mesh(part.mesh)
        eng.set_material(E.NEOHOOKEAN, lam, mu)
        eng.build_pattern()
        eng.set_owned(part.n_owned, part.elem_owned)
        l2g = part.local_to_global()
        nl, bs = part.n_local, 3
        dof_l2g = (l2g[:, None] * bs + np.arange(bs)[None]).reshape(-1)
        rng = np.random.default_rng(7)
        u_g = rng.standard_normal(n_nodes * bs) * 1e-3
        b_g = rng.standard_normal(n_nodes * bs)
        u = torch.from_numpy(u_g[dof_l2g]).to(dev)
        eng.hessian_dev(u.data_ptr())
        e_loc = torch.zeros(1, dtype=torch.float64, device=dev)
        eng.energy_dev(u.data_ptr(), e_loc.data_ptr())
        g_loc = torch.zeros(nl * bs, dtype=torch.float64, device=dev)
        eng.gradient_dev(u.data_ptr(), g_loc.data_ptr())
        comm = D.Comm()
        comm.all_reduce_sum(e_loc)
        fixed = np.nonzero(np.abs(part.mesh.node_xyz[:, 0]) < 1e-12)[0]
        eng.set_dirichlet(np.sort((fixed[:, None] * bs + np.arange(bs)[None]).reshape(-1)).astype(np.int32))
        cg = D.DistributedCG(D.HipOps(eng), part, bs, comm, dev)
        b = torch.from_numpy(b_g[dof_l2g]).to(dev)
        x = torch.zeros(nl * bs, dtype=torch.float64, device=dev)
        iters, err = cg.solve(b, x, E.PRECOND_JACOBI, 1000, 1e-12, check_every=10)  # the loop inside the library (host transport over gloo)
        torch.cuda.synchronize()
        assert cg.lib is not None and cg.lib.libcomm.kind == "host"
        # the Python-driven loop over the step functions: same kernels, same all-reduced scalars -> the same iterates
        x_py = torch.zeros(nl * bs, dtype=torch.float64, device=dev)
        iters_py, err_py = cg.solve(b, x_py, E.PRECOND_JACOBI, 1000, 1e-12, check_every=10, use_library=False)
        torch.cuda.synchronize()
        no_ = part.n_owned * bs
        assert iters_py == iters and torch.equal(x_py[:no_], x[:no_]), (iters_py, iters)
        # the library returns x with a current halo; energy over the partition through the library equals the all-reduced one
        xh = x.clone()
        cg.lib.halo_exchange(xh)
        assert torch.equal(xh, x)
        e_lib = torch.zeros(1, dtype=torch.float64, device=dev)
        cg.lib.energy(u, e_lib)
        torch.cuda.synchronize()
        assert float(e_lib.item()) == float(e_loc.item())
        assert cg.lib.halo_bytes() == cg.halo.bytes_per_exchange(bs)
        no = part.n_owned * bs
        out = [None] * world
        dist.all_gather_object(out, (int(part.lo), x.cpu().numpy()[:no], g_loc.cpu().numpy()[:no]))
        if rank == 0:
            ret["iters"], ret["err"], ret["energy"] = iters, err, float(e_loc.item())
            xg = np.zeros(n_nodes * bs)
            gg = np.zeros(n_nodes * bs)
            for lo, xv, gv in out:
                xg[lo * bs: lo * bs + xv.size] = xv
                gg[lo * bs: lo * bs + gv.size] = gv
            ret["x"], ret["g"] = xg, gg
        eng.close()
        ret[f"ok{rank}"] = True
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_single_engine(oracle):
    import torch.multiprocessing as mp
    from polyfem_amd import distributed as D
    from polyfem_amd import engine as E
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        assert all(ret.get(f"ok{r}") for r in range(world))
        iters, err, energy, xg, gg = ret["iters"], ret["err"], ret["energy"], ret["x"], ret["g"]
    # single engine on the whole mesh (world = 1 slab == whole grid, same global numbering)
    part, bounds, n_nodes = D.slab_part(6, 6, 0, 1, 2)
    eng = E.Engine(0)
    eng.set_mesh(part.mesh)
    eng.set_material(E.NEOHOOKEAN, 2e4 * 0.3 / (1.3 * 0.4), 2e4 / 2.6)
    eng.build_pattern()
    rng = np.random.default_rng(7)
    u = rng.standard_normal(n_nodes * 3) * 1e-3
    b = rng.standard_normal(n_nodes * 3)
    eng.hessian(u, fetch=False)
    e_ref = eng.energy(u)
    g_ref = eng.gradient(u)
    fixed = np.nonzero(np.abs(part.mesh.node_xyz[:, 0]) < 1e-12)[0]
    eng.set_dirichlet(np.sort((fixed[:, None] * 3 + np.arange(3)[None]).reshape(-1)).astype(np.int32))
    x_ref, it_ref, err_ref = eng.cg(b, None, E.PRECOND_JACOBI, 1000, 1e-12)
    eng.close()
    assert abs(energy - e_ref) <= 1e-12 * abs(e_ref)
    assert np.abs(gg - g_ref).max() <= 1e-12 * np.abs(g_ref).max()
    assert abs(iters - it_ref) <= 2 and err < 1e-11
    assert np.abs(xg - x_ref).max() <= 1e-8 * np.abs(x_ref).max()
    # ... and the CPU oracle on the whole mesh (not only the library against itself): energy, gradient, and the Eigen-order
    # CG on the explicitly reduced Hessian
    mesh = part.mesh
    lam, mu = 2e4 * 0.3 / (1.3 * 0.4), 2e4 / 2.6
    e_o = oracle.assemble_energy(mesh, oracle.NEOHOOKEAN, lam, mu, u)
    g_o = oracle.assemble_gradient(mesh, oracle.NEOHOOKEAN, lam, mu, u)
    assert abs(energy - e_o) <= 1e-12 * abs(e_o)
    assert np.abs(gg - g_o).max() <= 1e-12 * np.abs(g_o).max()
    bd = np.sort((fixed[:, None] * 3 + np.arange(3)[None]).reshape(-1))
    red = oracle.full_to_reduced_matrix(oracle.assemble_hessian(mesh, oracle.NEOHOOKEAN, lam, mu, u, 8), bd)
    keep = np.setdiff1d(np.arange(n_nodes * 3), bd)
    x_o, it_o, _ = oracle.cg(red, b[keep], jacobi=True, max_iter=1000, tol=1e-12)
    assert abs(iters - it_o) <= 2, (iters, it_o)
    assert np.abs(xg[keep] - x_o).max() <= 1e-8 * np.abs(x_o).max() and not xg[bd].any()


def _newton_case(n_nodes, node_xyz):
    """Load and constraint of the Newton test in global numbering: x = 0 face clamped, compression along x."""
    fixed = np.nonzero(np.abs(node_xyz[:, 0]) < 1e-12)[0]
    rhs = np.zeros(n_nodes * 3)
    rhs[0::3] = -6e3 / n_nodes
    return fixed, rhs


def _newton_worker(rank, world, port, ret):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from polyfem_amd import distributed as D
        from polyfem_amd import engine as E
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        stream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(stream)
        part, bounds, n_nodes = D.slab_part(4, 2, rank, world, 2)
        eng = E.Engine(0)
        eng.set_stream(stream.cuda_stream)
        eng.set_mesh(part.mesh)
        eng.set_material(E.NEOHOOKEAN, 2e4 * 0.3 / (1.3 * 0.4), 2e4 / 2.6)
        eng.build_pattern()
        eng.set_owned(part.n_owned, part.elem_owned)
        l2g = part.local_to_global()
        dof_l2g = (l2g[:, None] * 3 + np.arange(3)[None]).reshape(-1)
        fixed_l, _ = _newton_case(part.n_local, part.mesh.node_xyz)
        fixed_dofs = np.sort((fixed_l[:, None] * 3 + np.arange(3)[None]).reshape(-1)).astype(np.int32)
        eng.set_dirichlet(fixed_dofs)
        rhs_g = np.zeros(n_nodes * 3)
        rhs_g[0::3] = -6e3 / n_nodes
        newton = D.DistributedNewton(D.HipOps(eng), part, 3, D.Comm(), dev, fixed_dofs)
        x = torch.zeros(part.n_local * 3, dtype=torch.float64, device=dev)
        rhs_l = torch.from_numpy(rhs_g[dof_l2g]).to(dev)
        info = newton.solve(x, rhs_l, cg_check_every=16)  # the whole solve inside the library (pfx_dist_newton_dev)
        torch.cuda.synchronize()
        assert newton.cg.lib is not None and info["objective_evaluations"] is None
        assert not eng.take_numeric_flag(), "the driver must leave the numeric flag cleared"
        # the loop driven from Python over the step functions takes the same decisions and ends at the same point
        x_py = torch.zeros_like(x)
        info_py = newton.solve(x_py, rhs_l, cg_check_every=16, use_library=False)
        torch.cuda.synchronize()
        for key in ("status", "iterations", "line_search_iterations", "descent_strategy"):
            assert info_py[key] == info[key], (key, info_py, info)
        assert abs(info_py["cg_iterations"] - info["cg_iterations"]) <= 2 * info["iterations"]
        assert float((x_py - x).abs().max()) <= 1e-10 * float(x.abs().max())
        # the same solve with the AMG hierarchy that spans the ranks as the linear solver
        x_amg = torch.zeros_like(x)
        info_amg = newton.solve(x_amg, rhs_l, cg_precond=E.PRECOND_AMG, cg_tol=1e-10)
        torch.cuda.synchronize()
        assert info_amg["status"] in (D.F_DELTA, D.GRAD_NORM) and info_amg["iterations"] == info["iterations"]
        assert info_amg["cg_iterations"] < info["cg_iterations"]
        assert float((x_amg - x).abs().max()) <= 1e-7 * float(x.abs().max())
        no = part.n_owned * 3
        out = [None] * world
        dist.all_gather_object(out, (int(part.lo), x.cpu().numpy()[:no]))
        xg = np.zeros(n_nodes * 3)
        for lo, xv in out:
            xg[lo * 3: lo * 3 + xv.size] = xv
        assert np.array_equal(x.cpu().numpy(), xg[dof_l2g]), "halo of the iterate must be current"
        if rank == 0:
            ret["info"], ret["x"] = dict(info), xg
        eng.close()
        ret[f"ok{rank}"] = True
    finally:
        dist.destroy_process_group()


def test_distributed_newton_two_ranks_match_single_engine():
    """DistributedNewton (HIP kernels per rank, gloo between two ranks sharing the card) against the single-GPU
    pfx_newton_solve on the whole mesh: same decisions, same minimiser."""
    import torch.multiprocessing as mp
    from polyfem_amd import distributed as D
    from polyfem_amd import engine as E
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_newton_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        assert all(ret.get(f"ok{r}") for r in range(world))
        info, xg = ret["info"], ret["x"]
    part, bounds, n_nodes = D.slab_part(4, 4, 0, 1, 2)
    eng = E.Engine(0)
    eng.set_mesh(part.mesh)
    eng.set_material(E.NEOHOOKEAN, 2e4 * 0.3 / (1.3 * 0.4), 2e4 / 2.6)
    eng.build_pattern()
    fixed, rhs = _newton_case(n_nodes, part.mesh.node_xyz)
    eng.set_dirichlet(np.sort((fixed[:, None] * 3 + np.arange(3)[None]).reshape(-1)).astype(np.int32))
    x_ref, iref = eng.newton_solve(np.zeros(n_nodes * 3), rhs)
    eng.close()
    assert info["status"] == iref["status"] and info["status"] in (D.F_DELTA, D.GRAD_NORM)
    assert info["iterations"] == iref["iterations"] and info["iterations"] >= 2
    assert info["line_search_iterations"] == iref["line_search_iterations"]
    assert abs(info["cg_iterations"] - iref["cg_iterations"]) <= max(3 * info["iterations"], 0.03 * iref["cg_iterations"])
    assert abs(info["energy"] - iref["energy"]) <= 1e-9 * abs(iref["energy"])
    assert np.abs(xg - x_ref).max() <= 1e-8 * np.abs(x_ref).max()


def _euler_worker(rank, world, port, ret):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from polyfem_amd import distributed as D
        from polyfem_amd import engine as E
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        stream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(stream)
        part, bounds, n_nodes = D.slab_part(4, 2, rank, world, 2)
        eng = E.Engine(0)
        eng.set_stream(stream.cuda_stream)
        eng.set_mesh(part.mesh)
        eng.set_material(E.NEOHOOKEAN, 2e4 * 0.3 / (1.3 * 0.4), 2e4 / 2.6)
        eng.build_pattern()
        eng.set_owned(part.n_owned, part.elem_owned)
        rho, dt = 1000.0, 0.02
        f = np.array([0.0, 3.0, -9.81])
        rhs_l = eng.assemble_rhs(f, rho)            # owned rows are complete
        eng.inertia_set_mass(rho)
        fixed_l, _ = _newton_case(part.n_local, part.mesh.node_xyz)
        fixed_dofs = np.sort((fixed_l[:, None] * 3 + np.arange(3)[None]).reshape(-1)).astype(np.int32)
        eng.set_dirichlet(fixed_dofs)
        l2g = part.local_to_global()
        dof_l2g = (l2g[:, None] * 3 + np.arange(3)[None]).reshape(-1)
        v_g = 0.05 * np.random.default_rng(9).standard_normal(n_nodes * 3)
        v_l = v_g[dof_l2g].copy()
        v_l[fixed_dofs] = 0.0
        newton = D.DistributedNewton(D.HipOps(eng), part, 3, D.Comm(), dev, fixed_dofs)
        stepper = D.DistributedImplicitEuler(newton, dt)
        x = torch.zeros(part.n_local * 3, dtype=torch.float64, device=dev)
        v = torch.from_numpy(v_l).to(dev)
        a = torch.zeros_like(x)
        rhs = torch.from_numpy(rhs_l).to(dev)
        x0, v0 = x.clone(), v.clone()
        infos = [dict(stepper.step(x, v, a, rhs, cg_check_every=16)) for _ in range(2)]  # inside the library
        torch.cuda.synchronize()
        assert newton.cg.lib is not None
        # the steps driven from Python (Newton loop over the step functions, integrator bookkeeping in torch) agree
        x_py, v_py, a_py = x0.clone(), v0.clone(), torch.zeros_like(x)
        infos_py = [dict(stepper.step(x_py, v_py, a_py, rhs, cg_check_every=16, use_library=False)) for _ in range(2)]
        torch.cuda.synchronize()
        for i_lib, i_py in zip(infos, infos_py):
            for key in ("status", "iterations", "line_search_iterations"):
                assert i_lib[key] == i_py[key], (key, i_lib, i_py)
        for lib_vec, py_vec in ((x, x_py), (v, v_py), (a, a_py)):
            assert float((lib_vec - py_vec).abs().max()) <= 1e-9 * float(lib_vec.abs().max())
        # one Newton solve with the inertia term through pfx_inertia_enable + pfx_dist_newton_dev
        xt = (x + dt * v).contiguous()
        x_a, x_b = x.clone(), x.clone()
        i_a = newton.solve(x_a, rhs, x_tilde=xt, weight=dt * dt)
        i_b = newton.solve(x_b, rhs, x_tilde=xt, weight=dt * dt, use_library=False)
        assert i_a["iterations"] == i_b["iterations"] and float((x_a - x_b).abs().max()) <= 1e-10 * float(x_b.abs().max())
        no = part.n_owned * 3
        out = [None] * world
        dist.all_gather_object(out, (int(part.lo), x.cpu().numpy()[:no], v.cpu().numpy()[:no], a.cpu().numpy()[:no]))
        if rank == 0:
            xg, vg, ag = np.zeros(n_nodes * 3), np.zeros(n_nodes * 3), np.zeros(n_nodes * 3)
            for lo, xv, vv, av in out:
                xg[lo * 3: lo * 3 + xv.size], vg[lo * 3: lo * 3 + vv.size], ag[lo * 3: lo * 3 + av.size] = xv, vv, av
            ret["infos"], ret["x"], ret["v"], ret["a"] = infos, xg, vg, ag
        eng.close()
        ret[f"ok{rank}"] = True
    finally:
        dist.destroy_process_group()


def test_distributed_implicit_euler_two_ranks_match_single_engine():
    """Two implicit Euler steps with the mesh split over two ranks (HIP kernels per rank, gloo in between) against
    pfx_implicit_euler_step on the whole mesh."""
    import torch.multiprocessing as mp
    from polyfem_amd import distributed as D
    from polyfem_amd import engine as E
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_euler_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        assert all(ret.get(f"ok{r}") for r in range(world))
        infos, xg, vg, ag = ret["infos"], ret["x"], ret["v"], ret["a"]
    part, bounds, n_nodes = D.slab_part(4, 4, 0, 1, 2)
    eng = E.Engine(0)
    eng.set_mesh(part.mesh)
    eng.set_material(E.NEOHOOKEAN, 2e4 * 0.3 / (1.3 * 0.4), 2e4 / 2.6)
    eng.build_pattern()
    rho, dt = 1000.0, 0.02
    rhs = eng.assemble_rhs(np.array([0.0, 3.0, -9.81]), rho)
    eng.inertia_set_mass(rho)
    fixed, _ = _newton_case(n_nodes, part.mesh.node_xyz)
    fixed_dofs = np.sort((fixed[:, None] * 3 + np.arange(3)[None]).reshape(-1)).astype(np.int32)
    eng.set_dirichlet(fixed_dofs)
    x = np.zeros(n_nodes * 3)
    v = 0.05 * np.random.default_rng(9).standard_normal(n_nodes * 3)
    v[fixed_dofs] = 0.0
    for step in range(2):
        x, v, a, iref = eng.implicit_euler_step(dt, x, v, rhs)
        assert infos[step]["status"] == iref["status"] and infos[step]["iterations"] == iref["iterations"]
        assert infos[step]["line_search_iterations"] == iref["line_search_iterations"]
    eng.close()
    assert np.abs(xg - x).max() <= 1e-8 * np.abs(x).max()
    assert np.abs(vg - v).max() <= 1e-7 * np.abs(v).max()
    assert np.abs(ag - a).max() <= 1e-6 * np.abs(a).max()


def _amg_case(n_nodes, node_xyz):
    """Right-hand side and constraint of the AMG test in global numbering: x = 0 face clamped, seeded load."""
    fixed = np.nonzero(np.abs(node_xyz[:, 0]) < 1e-12)[0]
    bd = np.sort((fixed[:, None] * 3 + np.arange(3)[None]).reshape(-1)).astype(np.int32)
    b = np.random.default_rng(11).standard_normal(n_nodes * 3)
    return bd, b


def _amg_worker(rank, world, port, ret):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from polyfem_amd import distributed as D
        from polyfem_amd import engine as E
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        stream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(stream)
        n, cpr = 8, 4
        part, bounds, n_nodes = D.slab_part(n, cpr, rank, world, 2)
        eng = E.Engine(0)
        eng.set_stream(stream.cuda_stream)
        eng.set_mesh(part.mesh)
        eng.set_material(E.LINEAR_ELASTICITY, 2e4 * 0.3 / (1.3 * 0.4), 2e4 / 2.6)
        eng.build_pattern()
        eng.set_owned(part.n_owned, part.elem_owned)
        eng.assemble_linear()
        l2g = part.local_to_global()
        bs = 3
        dof_l2g = (l2g[:, None] * bs + np.arange(bs)[None]).reshape(-1)
        fixed = np.nonzero(np.abs(part.mesh.node_xyz[:, 0]) < 1e-12)[0]
        eng.set_dirichlet(np.sort((fixed[:, None] * bs + np.arange(bs)[None]).reshape(-1)).astype(np.int32))
        _, b_g = _amg_case(n_nodes, np.zeros((1, 3)))
        comm = D.Comm()
        cg = D.DistributedCG(D.HipOps(eng), part, bs, comm, dev)
        b = torch.from_numpy(b_g[dof_l2g]).to(dev)
        x = torch.zeros(part.n_local * bs, dtype=torch.float64, device=dev)
        iters, err = cg.solve(b, x, E.PRECOND_AMG, 500, 1e-10)
        # a second solve with the hierarchy in place and a non-zero guess
        x2 = x.clone() * 0.5
        iters2, err2 = cg.solve(b, x2, E.PRECOND_AMG, 500, 1e-10)
        torch.cuda.synchronize()
        no = part.n_owned * bs
        out = [None] * world
        dist.all_gather_object(out, (int(part.lo), x.cpu().numpy()[:no], x2.cpu().numpy()[:no]))
        if rank == 0:
            xg, xg2 = np.zeros(n_nodes * bs), np.zeros(n_nodes * bs)
            for lo, xv, xv2 in out:
                xg[lo * bs: lo * bs + xv.size] = xv
                xg2[lo * bs: lo * bs + xv2.size] = xv2
            ret["iters"], ret["err"], ret["x"], ret["iters2"], ret["err2"], ret["x2"] = iters, err, xg, iters2, err2, xg2
            ret["levels"] = eng.amg_info()["n_levels"]
        cg.close()
        eng.close()
        ret[f"ok{rank}"] = True
    finally:
        dist.destroy_process_group()


def test_block_jacobi_amg_two_ranks_converges_to_the_single_engine_solution():
    """PFX_PRECOND_AMG over a partition through DistributedCG (the hierarchy that spans the ranks, inside the library): the
    solution is that of the single-engine AMG solve (both to 1e-10) at about its iteration count, a second solve reuses the
    hierarchy and honours the initial guess."""
    import torch.multiprocessing as mp
    from polyfem_amd import distributed as D
    from polyfem_amd import engine as E
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_amg_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        assert all(ret.get(f"ok{r}") for r in range(world))
        iters, err, xg, iters2, err2, xg2, levels = (ret[k] for k in ("iters", "err", "x", "iters2", "err2", "x2", "levels"))
    part, bounds, n_nodes = D.slab_part(8, 8, 0, 1, 2)
    eng = E.Engine(0)
    eng.set_mesh(part.mesh)
    eng.set_material(E.LINEAR_ELASTICITY, 2e4 * 0.3 / (1.3 * 0.4), 2e4 / 2.6)
    eng.build_pattern()
    eng.assemble_linear()
    bd, b = _amg_case(n_nodes, part.mesh.node_xyz)
    eng.set_dirichlet(bd)
    x_ref, it_ref, err_ref = eng.cg(b, None, E.PRECOND_AMG, 500, 1e-10)
    x_jac, it_jac, _ = eng.cg(b, None, E.PRECOND_JACOBI, 5000, 1e-10)
    eng.close()
    assert levels >= 2
    assert err < 1e-10 and err2 < 1e-10 and err_ref < 1e-10
    assert iters <= 2 * it_ref + 8, (iters, it_ref)
    assert iters < it_jac // 3, (iters, it_jac)
    assert iters2 <= iters
    scale = np.abs(x_ref).max()
    assert np.abs(xg - x_ref).max() <= 1e-7 * scale
    assert np.abs(xg2 - x_ref).max() <= 1e-7 * scale
    assert not xg[bd].any()


def _coarse_space_worker(rank, world, port, ret):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from polyfem_amd import distributed as D
        from polyfem_amd import engine as E
        from polyfem_amd import mesh as M
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        stream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(stream)
        n, cpr = 6, 6  # a bar of 6 x 6 x (6 * world) cells, clamped at one small face: the longer, the worse block-Jacobi
        part, bounds, n_nodes = D.slab_part(n, cpr, rank, world, 2)
        eng = E.Engine(0)
        eng.set_stream(stream.cuda_stream)
        eng.set_mesh(part.mesh)
        eng.set_material(E.LINEAR_ELASTICITY, 2e4 * 0.3 / (1.3 * 0.4), 2e4 / 2.6)
        eng.build_pattern()
        eng.set_owned(part.n_owned, part.elem_owned)
        eng.assemble_linear()
        l2g = part.local_to_global()
        bs = 3
        dof_l2g = (l2g[:, None] * bs + np.arange(bs)[None]).reshape(-1)
        fixed = np.nonzero(np.abs(part.mesh.node_xyz[:, 2]) < 1e-12)[0]  # z = 0: the first rank's outer face
        fixed_dofs = np.sort((fixed[:, None] * bs + np.arange(bs)[None]).reshape(-1)).astype(np.int32)
        eng.set_dirichlet(fixed_dofs)
        b_g = np.random.default_rng(21).standard_normal(n_nodes * bs)
        comm = D.Comm()
        cg = D.DistributedCG(D.HipOps(eng), part, bs, comm, dev)
        b = torch.from_numpy(b_g[dof_l2g]).to(dev)
        x0 = torch.zeros(part.n_local * bs, dtype=torch.float64, device=dev)
        it0, err0 = cg.solve(b, x0, E.PRECOND_AMG, 1000, 1e-10, use_library=False)  # one hierarchy per rank (block-Jacobi)
        cg.enable_coarse_space(M.node_positions(part.mesh), fixed_dofs)
        x1 = torch.zeros_like(x0)
        it1, err1 = cg.solve(b, x1, E.PRECOND_AMG, 1000, 1e-10, use_library=False)  # + the ranks' rigid-body modes
        x2 = torch.zeros_like(x0)
        it2, err2 = cg.solve(b, x2, E.PRECOND_AMG, 1000, 1e-10)  # ONE hierarchy that spans the ranks (pfx_dist_cg_dev)
        torch.cuda.synchronize()
        no = part.n_owned * bs
        diff = torch.tensor([float((x1[:no] - x0[:no]).abs().max()), float(x0[:no].abs().max()), float((x2[:no] - x0[:no]).abs().max())], dtype=torch.float64)
        dist.all_reduce(diff, op=dist.ReduceOp.MAX)
        if rank == 0:
            ret["it0"], ret["err0"], ret["it1"], ret["err1"], ret["diff"], ret["scale"] = it0, err0, it1, err1, float(diff[0]), float(diff[1])
            ret["it2"], ret["err2"], ret["diff2"] = it2, err2, float(diff[2])
        cg.close()
        eng.close()
        ret[f"ok{rank}"] = True
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_coarse_space_over_the_ranks_cuts_the_iterations(world):
    """The rigid-body modes of every rank's block as a coarse space beside the per-rank AMG hierarchies: the same
    solution, fewer iterations, and the more ranks the larger the gain (a bar cut into slabs and clamped at one end is the
    hard case for block-Jacobi: the far slabs float)."""
    import torch.multiprocessing as mp
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_coarse_space_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        assert all(ret.get(f"ok{r}") for r in range(world))
        it0, err0, it1, err1, diff, scale = (ret[k] for k in ("it0", "err0", "it1", "err1", "diff", "scale"))
        it2, err2, diff2 = ret["it2"], ret["err2"], ret["diff2"]
    # one context on the whole bar
    from polyfem_amd import distributed as D
    from polyfem_amd import engine as E
    part, bounds, n_nodes = D.slab_part(6, 6 * world, 0, 1, 2)
    eng = E.Engine(0)
    try:
        eng.set_mesh(part.mesh)
        eng.set_material(E.LINEAR_ELASTICITY, 2e4 * 0.3 / (1.3 * 0.4), 2e4 / 2.6)
        eng.build_pattern()
        eng.assemble_linear()
        fixed = np.nonzero(np.abs(part.mesh.node_xyz[:, 2]) < 1e-12)[0]
        eng.set_dirichlet(np.sort((fixed[:, None] * 3 + np.arange(3)[None]).reshape(-1)).astype(np.int32))
        b_g = np.random.default_rng(21).standard_normal(n_nodes * 3)
        _, it_one, _ = eng.cg(b_g, None, E.PRECOND_AMG, 1000, 1e-10)
    finally:
        eng.close()
    print(f"world {world}: block-Jacobi AMG {it0} iterations, with the coarse space {it1}, one hierarchy over the ranks {it2}, one context {it_one}")
    assert err0 < 1e-10 and err1 < 1e-10 and err2 < 1e-10
    assert diff <= 1e-6 * scale and diff2 <= 1e-6 * scale
    assert it1 < it0
    if world >= 4:
        assert it1 <= 0.8 * it0
    # the hierarchy that spans the ranks keeps the iteration count of the unpartitioned solve
    assert it2 <= 1.5 * it_one and it2 <= it1, (it2, it_one, it1)


@pytest.mark.parametrize("precond", [1, 2])
def test_single_rank_driver_with_slice_ordered_product(precond):
    """One rank, no halo, more than 65536 owned nodes: the solver copy's product then stays in slice order between the
    two half products of the distributed loop (interior rows; boundary rows: none on this rank), and the update kernel
    must read it there.  Reference: pfx_cg_dev on the same engine."""
    import torch
    from polyfem_amd import distributed as D
    from polyfem_amd import engine as E
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        part, bounds, n_nodes = D.slab_part(20, 20, 0, 1, 2)
        assert part.n_owned == n_nodes > 65536
        eng = E.Engine(0)
        try:
            eng.set_stream(stream.cuda_stream)
            eng.set_mesh(part.mesh)
            lam, mu = 1e5 * 0.3 / (1.3 * 0.4), 1e5 / 2.6
            eng.set_material(E.LINEAR_ELASTICITY, lam, mu)
            eng.build_pattern()
            eng.assemble_linear()
            fixed = np.nonzero(np.abs(part.mesh.node_xyz[:, 0]) < 1e-12)[0]
            eng.set_dirichlet(np.sort((fixed[:, None] * 3 + np.arange(3)[None]).reshape(-1)).astype(np.int32))
            n = n_nodes * 3
            b = torch.from_numpy(np.random.default_rng(3).standard_normal(n)).to(dev)
            tol, max_iter = (1e-10, 300) if precond == 2 else (1e-6, 1000)
            x_ref = torch.zeros(n, dtype=torch.float64, device=dev)
            it_ref, err_ref = eng.cg_dev(b.data_ptr(), x_ref.data_ptr(), precond, max_iter, tol)
            cg = D.DistributedCG(D.HipOps(eng), part, 3, D.Comm(), dev)
            x = torch.zeros(n, dtype=torch.float64, device=dev)
            iters, err = cg.solve(b, x, precond, max_iter, tol)
            torch.cuda.synchronize()
            assert abs(iters - it_ref) <= 1 and err <= tol, (iters, it_ref, err, err_ref)
            assert float((x - x_ref).abs().max()) <= 1e-6 * float(x_ref.abs().max())
            # a second solver object on the same engine, the first one dropped afterwards: the engine keeps the
            # workspace of the object that is alive
            cg2 = D.DistributedCG(D.HipOps(eng), part, 3, D.Comm(), dev)
            cg.close()
            x2 = torch.zeros(n, dtype=torch.float64, device=dev)
            iters2, err2 = cg2.solve(b, x2, precond, max_iter, tol)
            torch.cuda.synchronize()
            assert iters2 == iters and float((x2 - x).abs().max()) <= 1e-12 * float(x.abs().max())
            cg2.close()
        finally:
            eng.close()


def test_one_rank_rccl_communicator_runs_the_library_loop():
    """RCCL on the hardware that is there: a ONE-rank communicator made by the library itself (ncclGetUniqueId,
    ncclCommInitRank), the distributed CG loop with its all-reduces enqueued on the context's stream through ncclAllReduce.
    No neighbour exists, so ncclSend / ncclRecv stay unexecuted until a multi-GPU node runs the benchmark."""
    import ctypes as C
    import torch
    from polyfem_amd import distributed as D
    from polyfem_amd import engine as E
    L = E.load()
    ident = C.create_string_buffer(128)
    assert L.pfx_comm_rccl_unique_id(ident) == 0
    comm = C.c_void_p()
    assert L.pfx_comm_create_rccl(C.byref(comm), ident, 0, 1, 0) == 0
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        part, bounds, n_nodes = D.slab_part(6, 6, 0, 1, 2)
        eng = E.Engine(0)
        try:
            eng.set_stream(stream.cuda_stream)
            eng.set_mesh(part.mesh)
            eng.set_material(E.LINEAR_ELASTICITY, 1e5 * 0.3 / (1.3 * 0.4), 1e5 / 2.6)
            eng.build_pattern()
            eng.assemble_linear()
            fixed = np.nonzero(np.abs(part.mesh.node_xyz[:, 0]) < 1e-12)[0]
            eng.set_dirichlet(np.sort((fixed[:, None] * 3 + np.arange(3)[None]).reshape(-1)).astype(np.int32))
            n = n_nodes * 3
            b = torch.from_numpy(np.random.default_rng(3).standard_normal(n)).to(dev)
            x_ref = torch.zeros(n, dtype=torch.float64, device=dev)
            it_ref, err_ref = eng.cg_dev(b.data_ptr(), x_ref.data_ptr(), E.PRECOND_JACOBI, 1000, 1e-12)
            dist = C.c_void_p()
            assert L.pfx_dist_create(C.byref(dist), eng.h, comm, 0, None, None, None, None) == 0
            x = torch.zeros(n, dtype=torch.float64, device=dev)
            it, err = C.c_int(), C.c_double()
            rc = L.pfx_dist_cg_dev(dist, b.data_ptr(), x.data_ptr(), E.PRECOND_JACOBI, 1000, 1e-12, 25, C.byref(it), C.byref(err))
            assert rc == 0, L.pfx_dist_last_error(dist).decode()
            torch.cuda.synchronize()
            assert it.value == it_ref and torch.equal(x, x_ref)
            L.pfx_dist_destroy(dist)
        finally:
            eng.close()
            L.pfx_comm_destroy(comm)


def _span_worker(rank, world, port, ret, n, cpr):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from polyfem_amd import distributed as D
        from polyfem_amd import engine as E
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        stream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(stream)
        part, bounds, n_nodes = D.slab_part(n, cpr, rank, world, 2)
        lam, mu = 1e5 * 0.3 / (1.3 * 0.4), 1e5 / 2.6
        eng = E.Engine(0)
        eng.set_stream(stream.cuda_stream)
        eng.set_mesh(part.mesh)
        eng.set_material(E.LINEAR_ELASTICITY, lam, mu)
        eng.build_pattern()
        eng.set_owned(part.n_owned, part.elem_owned)
        eng.assemble_linear()
        eng.amg_set_params(coarsest_dofs=120, mixed_precision=0)  # double everywhere: compared with the mirror at 1e-10
        bs, nl = 3, part.n_local
        l2g = part.local_to_global()
        dof_l2g = (l2g[:, None] * bs + np.arange(bs)[None]).reshape(-1)
        fixed = np.nonzero(np.abs(part.mesh.node_xyz[:, 0]) < 1e-12)[0]
        fixed_dofs = np.sort((fixed[:, None] * bs + np.arange(bs)[None]).reshape(-1)).astype(np.int32)
        eng.set_dirichlet(fixed_dofs)
        comm = D.Comm()
        cg = D.DistributedCG(D.HipOps(eng), part, bs, comm, dev)
        rng = np.random.default_rng(11)
        r_g = rng.standard_normal(n_nodes * bs)
        b_g = rng.standard_normal(n_nodes * bs)
        r = torch.from_numpy(r_g[dof_l2g]).to(dev)
        r[torch.from_numpy(fixed_dofs.astype(np.int64)).to(dev)] = 0.0
        z = torch.zeros(nl * bs, dtype=torch.float64, device=dev)
        cg.lib.amg_apply(r, z)
        torch.cuda.synchronize()
        info = eng.amg_info()
        aggs = [eng.amg_aggregates(l) for l in range(info["n_levels"] - 1)]
        aggs[0] = aggs[0][:part.n_owned]
        b = torch.from_numpy(b_g[dof_l2g]).to(dev)
        x = torch.zeros(nl * bs, dtype=torch.float64, device=dev)
        iters, err = cg.solve(b, x, E.PRECOND_AMG, 500, 1e-10)
        # the per-rank hierarchies + the ranks' rigid-body modes, for comparison
        cg.enable_coarse_space(part.mesh.node_xyz, fixed_dofs)
        x2 = torch.zeros(nl * bs, dtype=torch.float64, device=dev)
        iters_bj, err_bj = cg.solve(b, x2, E.PRECOND_AMG, 500, 1e-10, use_library=False)
        torch.cuda.synchronize()
        no = part.n_owned * bs
        out = [None] * world
        dist.all_gather_object(out, dict(lo=int(part.lo), z=z.cpu().numpy()[:no], x=x.cpu().numpy()[:no], aggs=aggs, n_vertices=info["n_vertices"],
                                         n_aggregates=info["n_aggregates"], lambda_max=info["lambda_max"], n_levels=info["n_levels"]))
        if rank == 0:
            ret["ranks"], ret["iters"], ret["err"], ret["iters_bj"] = out, iters, err, iters_bj
        cg.close()
        eng.close()
        ret[f"ok{rank}"] = True
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,cpr", [(2, 6, 3), (3, 5, 2)])
def test_hierarchy_that_spans_the_ranks_is_the_serial_one_with_the_same_aggregates(oracle, world, n, cpr):
    """PFX_PRECOND_AMG over a partition (pfx_dist_cg_dev): the ranks share one card and talk through the host transport.
    The preconditioner must be the one a single context builds from the union of the ranks' aggregates: one application
    against the scipy mirror of the serial algorithm fed with those aggregates (1e-10), PCG iteration count within +-2 of
    the mirror's, and fewer iterations than the per-rank hierarchies need."""
    import scipy.sparse as sp
    import torch.multiprocessing as mp
    from polyfem_amd import distributed as D
    from polyfem_amd import engine as E
    from polyfem_amd import mesh as M
    from tests import amg_mirror as AM
    from tests.helpers import csc_to_scipy
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_span_worker, args=(world, _free_port(), ret, n, cpr), nprocs=world, join=True)
        assert all(ret.get(f"ok{r}") for r in range(world))
        ranks, iters, err, iters_bj = ret["ranks"], ret["iters"], ret["err"], ret["iters_bj"]
    part, bounds, n_nodes = D.slab_part(n, cpr * world, 0, 1, 2)
    mesh = part.mesh
    lam, mu = oracle.lame(1e5, 0.3)
    N = n_nodes * 3
    A = csc_to_scipy(oracle.assemble_linear(mesh, oracle.LINEAR_ELASTICITY, lam, mu), N).tocsr()
    fixed_nodes = np.nonzero(np.abs(mesh.node_xyz[:, 0]) < 1e-12)[0]
    fixed = np.zeros(N, dtype=bool)
    fixed[(fixed_nodes[:, None] * 3 + np.arange(3)[None]).reshape(-1)] = True
    Dm = sp.diags((~fixed).astype(float))
    Am = (Dm @ A @ Dm + sp.diags(fixed.astype(float))).tocsr()
    # the union of the ranks' aggregates in a global numbering: rank r's vertices of a level follow rank r-1's
    n_levels = ranks[0]["n_levels"]
    assert n_levels >= 3 and all(rk["n_levels"] == n_levels for rk in ranks)
    assert all(rk["lambda_max"] == ranks[0]["lambda_max"] for rk in ranks), "every rank must relax with the same spectral radius"
    aggs_global = []
    for l in range(n_levels - 1):
        off_agg = np.cumsum([0] + [rk["n_aggregates"][l] for rk in ranks])
        parts = []
        for r, rk in enumerate(ranks):
            a = np.asarray(rk["aggs"][l]).astype(np.int64)
            assert a.size == rk["n_vertices"][l] and (a < rk["n_aggregates"][l]).all()
            parts.append(np.where(a >= 0, a + off_agg[r], -1))
        aggs_global.append(np.concatenate(parts))
    assert aggs_global[0].size == n_nodes
    levels = AM.build(Am, M.node_positions(mesh), fixed, aggs_global, ranks[0]["lambda_max"], ncycle=2)
    rng = np.random.default_rng(11)
    r_g = rng.standard_normal(N)
    b_g = rng.standard_normal(N)
    r_g[fixed] = 0
    z_ref = AM.cycle(levels, 0, r_g)
    z = np.zeros(N)
    x = np.zeros(N)
    for rk in ranks:
        z[rk["lo"] * 3: rk["lo"] * 3 + rk["z"].size] = rk["z"]
        x[rk["lo"] * 3: rk["lo"] * 3 + rk["x"].size] = rk["x"]
    assert np.abs(z - z_ref).max() <= 1e-10 * np.abs(z_ref).max()
    bm = np.where(fixed, 0.0, b_g)
    x_ref, it_ref, _ = AM.pcg(Am, bm, lambda v: AM.cycle(levels, 0, v), 1e-10, 500)
    assert abs(iters - it_ref) <= 2 and err <= 1e-10, (iters, it_ref, err)
    assert np.linalg.norm(Am @ x - bm) <= 1.01e-10 * np.linalg.norm(bm)
    # one context on the whole mesh, its own aggregates: the partition costs (almost) nothing
    eng = E.Engine(0)
    try:
        eng.set_mesh(mesh)
        eng.set_material(E.LINEAR_ELASTICITY, lam, mu)
        eng.build_pattern()
        eng.assemble_linear()
        eng.amg_set_params(coarsest_dofs=120)
        eng.set_dirichlet(np.nonzero(fixed)[0].astype(np.int32))
        _, it_one, _ = eng.cg(b_g, None, E.PRECOND_AMG, 500, 1e-10)
    finally:
        eng.close()
    assert iters <= 1.3 * it_one + 2, (iters, it_one)
    assert iters <= iters_bj, (iters, iters_bj)


# ---- 2-D: a triangle mesh over two ranks (2 x 2 blocks) --------------------------------------------------------------


def _tri_case():
    from polyfem_amd import mesh as M
    mesh = M.square_tris(14, 2, jitter=0.1, seed=4)  # lattice numbering: contiguous node ranges are strips in y
    n = mesh.n_nodes
    pos = M.node_positions(mesh)
    fixed = np.nonzero(np.abs(pos[:, 0]) < 1e-12)[0]
    fixed_dofs = np.sort((fixed[:, None] * 2 + np.arange(2)[None]).reshape(-1)).astype(np.int32)
    rhs = np.zeros(2 * n)
    rhs[1::2] = -60.0 / n
    lam, mu = 2e4 * 0.3 / (1.0 - 0.09), 2e4 / 2.6
    return mesh, pos, fixed_dofs, rhs, lam, mu


def _tri_worker(rank, world, port, ret):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from polyfem_amd import distributed as D
        from polyfem_amd import engine as E
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        stream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(stream)
        mesh, pos, fixed_g, rhs_g, lam, mu = _tri_case()
        n_nodes, bs = mesh.n_nodes, 2
        bounds = D.owner_ranges(n_nodes, world)
        part = D.partition(mesh.kind, mesh.conn, mesh.gconn, mesh.xyz, bounds, rank, node_xyz=pos)
        eng = E.Engine(0)
        eng.set_stream(stream.cuda_stream)
        eng.set_mesh(part.mesh)
        eng.set_material(E.NEOHOOKEAN, lam, mu)
        eng.build_pattern()
        eng.set_owned(part.n_owned, part.elem_owned)
        l2g = part.local_to_global()
        nl = part.n_local
        dof_l2g = (l2g[:, None] * bs + np.arange(bs)[None]).reshape(-1)
        g2l = {int(g): i for i, g in enumerate(dof_l2g)}
        fixed_l = np.array(sorted(g2l[int(g)] for g in fixed_g if int(g) in g2l), dtype=np.int32)
        rng = np.random.default_rng(7)
        u_g = rng.standard_normal(n_nodes * bs) * 1e-3
        b_g = rng.standard_normal(n_nodes * bs)
        u = torch.from_numpy(u_g[dof_l2g]).to(dev)
        eng.hessian_dev(u.data_ptr())
        eng.set_dirichlet(fixed_l)
        comm = D.Comm()
        cg = D.DistributedCG(D.HipOps(eng), part, bs, comm, dev)
        b = torch.from_numpy(b_g[dof_l2g]).to(dev)
        x = torch.zeros(nl * bs, dtype=torch.float64, device=dev)
        iters, err = cg.solve(b, x, E.PRECOND_JACOBI, 3000, 1e-12, check_every=10)
        x_amg = torch.zeros_like(x)
        it_amg, err_amg = cg.solve(b, x_amg, E.PRECOND_AMG, 500, 1e-10)
        torch.cuda.synchronize()
        assert cg.lib is not None and cg.lib.halo_bytes() == cg.halo.bytes_per_exchange(bs)
        e_lib = torch.zeros(1, dtype=torch.float64, device=dev)
        cg.lib.energy(u, e_lib)
        newton = D.DistributedNewton(D.HipOps(eng), part, bs, comm, dev, fixed_l)
        xn = torch.zeros(nl * bs, dtype=torch.float64, device=dev)
        info = newton.solve(xn, torch.from_numpy(rhs_g[dof_l2g]).to(dev), cg_check_every=16)
        torch.cuda.synchronize()
        no = part.n_owned * bs
        out = [None] * world
        dist.all_gather_object(out, (int(part.lo), x.cpu().numpy()[:no], x_amg.cpu().numpy()[:no], xn.cpu().numpy()[:no]))
        if rank == 0:
            xg, xa, xnn = np.zeros(n_nodes * bs), np.zeros(n_nodes * bs), np.zeros(n_nodes * bs)
            for lo, xv, av, nv in out:
                xg[lo * bs: lo * bs + xv.size], xa[lo * bs: lo * bs + av.size], xnn[lo * bs: lo * bs + nv.size] = xv, av, nv
            ret["cg"] = (iters, err, it_amg, err_amg, float(e_lib.item()))
            ret["x"], ret["x_amg"], ret["x_newton"], ret["info"] = xg, xa, xnn, dict(info)
        eng.close()
        ret[f"ok{rank}"] = True
    finally:
        dist.destroy_process_group()


def test_triangle_mesh_over_two_ranks_matches_single_engine(oracle):
    """2 x 2 blocks through the partitioned path: energy, Jacobi-CG (Eigen order), AMG-CG with the hierarchy that spans the
    ranks and the Newton solve of two ranks against the single engine on the whole mesh, and the CG against the oracle."""
    import torch.multiprocessing as mp
    from polyfem_amd import engine as E
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_tri_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        assert all(ret.get(f"ok{r}") for r in range(world))
        (iters, err, it_amg, err_amg, energy), xg, xa, xn, info = ret["cg"], ret["x"], ret["x_amg"], ret["x_newton"], ret["info"]
    mesh, pos, fixed_dofs, rhs, lam, mu = _tri_case()
    n = 2 * mesh.n_nodes
    eng = E.Engine(0)
    eng.set_mesh(mesh)
    eng.set_material(E.NEOHOOKEAN, lam, mu)
    eng.build_pattern()
    rng = np.random.default_rng(7)
    u = rng.standard_normal(n) * 1e-3
    b = rng.standard_normal(n)
    eng.hessian(u, fetch=False)
    assert abs(energy - eng.energy(u)) <= 1e-12 * abs(energy)
    assert abs(energy - oracle.assemble_energy(mesh, oracle.NEOHOOKEAN, lam, mu, u)) <= 1e-12 * abs(energy)
    eng.set_dirichlet(fixed_dofs)
    x_ref, it_ref, _ = eng.cg(b, None, E.PRECOND_JACOBI, 3000, 1e-12)
    assert abs(iters - it_ref) <= 2 and err < 1e-11
    assert np.abs(xg - x_ref).max() <= 1e-8 * np.abs(x_ref).max()
    red = oracle.full_to_reduced_matrix(oracle.assemble_hessian(mesh, oracle.NEOHOOKEAN, lam, mu, u, 8), fixed_dofs)
    keep = np.setdiff1d(np.arange(n), fixed_dofs)
    x_o, it_o, _ = oracle.cg(red, b[keep], jacobi=True, max_iter=3000, tol=1e-12)
    assert abs(iters - it_o) <= 2, (iters, it_o)
    assert np.abs(xg[keep] - x_o).max() <= 1e-8 * np.abs(x_o).max() and not xg[fixed_dofs].any()
    assert err_amg <= 1e-10 and it_amg < iters / 3, (it_amg, iters)
    assert np.abs(xa - x_ref).max() <= 1e-6 * np.abs(x_ref).max()
    x_n, iref = eng.newton_solve(np.zeros(n), rhs)
    eng.close()
    assert info["status"] == iref["status"] and info["iterations"] == iref["iterations"] >= 2
    assert info["line_search_iterations"] == iref["line_search_iterations"]
    assert np.abs(xn - x_n).max() <= 1e-8 * np.abs(x_n).max()


# ---- Newmark and BDF over a partition ---------------------------------------------------------------------------------


def _transient_setup(eng, mesh, E):
    eng.set_mesh(mesh)
    eng.set_material(E.NEOHOOKEAN, 2e4 * 0.3 / (1.3 * 0.4), 2e4 / 2.6)
    eng.build_pattern()


def _integrator_worker(rank, world, port, kind, ret):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from polyfem_amd import distributed as D
        from polyfem_amd import engine as E
        from polyfem_amd import time_integrator as TI
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        stream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(stream)
        part, bounds, n_nodes = D.slab_part(4, 2, rank, world, 2)
        eng = E.Engine(0)
        eng.set_stream(stream.cuda_stream)
        _transient_setup(eng, part.mesh, E)
        eng.set_owned(part.n_owned, part.elem_owned)
        rho, dt = 1000.0, 0.02
        rhs_l = eng.assemble_rhs(np.array([0.0, 3.0, -9.81]), rho)
        eng.inertia_set_mass(rho)
        fixed_l, _ = _newton_case(part.n_local, part.mesh.node_xyz)
        fixed_dofs = np.sort((fixed_l[:, None] * 3 + np.arange(3)[None]).reshape(-1)).astype(np.int32)
        eng.set_dirichlet(fixed_dofs)
        l2g = part.local_to_global()
        dof_l2g = (l2g[:, None] * 3 + np.arange(3)[None]).reshape(-1)
        v_l = (0.05 * np.random.default_rng(9).standard_normal(n_nodes * 3))[dof_l2g].copy()
        v_l[fixed_dofs] = 0.0
        newton = D.DistributedNewton(D.HipOps(eng), part, 3, D.Comm(), dev, fixed_dofs)
        ti = TI.construct_time_integrator(kind)
        z = torch.zeros(part.n_local * 3, dtype=torch.float64, device=dev)
        ti.init(z, torch.from_numpy(v_l).to(dev), z.clone(), dt)
        solver = D.DistributedTransientSolver(newton, ti, torch.from_numpy(rhs_l).to(dev), cg_check_every=16)
        infos = []
        for _ in range(3):
            x, info = solver.step()
            infos.append(dict(info))
        torch.cuda.synchronize()
        no = part.n_owned * 3
        out = [None] * world
        dist.all_gather_object(out, (int(part.lo), x.cpu().numpy()[:no], ti.v_prev().cpu().numpy()[:no], ti.a_prev().cpu().numpy()[:no]))
        xg, vg, ag = np.zeros(n_nodes * 3), np.zeros(n_nodes * 3), np.zeros(n_nodes * 3)
        for lo, xv, vv, av in out:
            xg[lo * 3: lo * 3 + xv.size], vg[lo * 3: lo * 3 + vv.size], ag[lo * 3: lo * 3 + av.size] = xv, vv, av
        # the halos of the integrator's vectors are those of their owners without any exchange of their own
        assert np.abs(ti.v_prev().cpu().numpy() - vg[dof_l2g]).max() <= 1e-9 * np.abs(vg).max()
        if rank == 0:
            ret["infos"], ret["x"], ret["v"], ret["a"] = infos, xg, vg, ag
        eng.close()
        ret[f"ok{rank}"] = True
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["ImplicitNewmark", "BDF2"])
def test_newmark_and_bdf_over_two_ranks_match_single_engine(kind):
    """Three steps of ImplicitNewmark / BDF2 (time_integrator/ImplicitNewmark.cpp, BDF.cpp) with the mesh split over two
    ranks -- every Newton solve inside the library, the integrator's bookkeeping on local vectors -- against the
    single-engine TransientSolver on the whole mesh."""
    import torch
    import torch.multiprocessing as mp
    from polyfem_amd import distributed as D
    from polyfem_amd import engine as E
    from polyfem_amd import time_integrator as TI
    spec = {"type": "BDF", "steps": 2} if kind == "BDF2" else kind
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_integrator_worker, args=(world, _free_port(), spec, ret), nprocs=world, join=True)
        assert all(ret.get(f"ok{r}") for r in range(world))
        infos, xg, vg, ag = ret["infos"], ret["x"], ret["v"], ret["a"]
    part, bounds, n_nodes = D.slab_part(4, 4, 0, 1, 2)
    eng = E.Engine(0)
    _transient_setup(eng, part.mesh, E)
    rho, dt = 1000.0, 0.02
    rhs = eng.assemble_rhs(np.array([0.0, 3.0, -9.81]), rho)
    eng.inertia_set_mass(rho)
    fixed, _ = _newton_case(n_nodes, part.mesh.node_xyz)
    fixed_dofs = np.sort((fixed[:, None] * 3 + np.arange(3)[None]).reshape(-1)).astype(np.int32)
    eng.set_dirichlet(fixed_dofs)
    v = 0.05 * np.random.default_rng(9).standard_normal(n_nodes * 3)
    v[fixed_dofs] = 0.0
    dev = torch.device("cuda", 0)
    ti = TI.construct_time_integrator(spec)
    z = torch.zeros(n_nodes * 3, dtype=torch.float64, device=dev)
    ti.init(z, torch.from_numpy(v).to(dev), z.clone(), dt)
    solver = TI.TransientSolver(eng, ti, torch.from_numpy(rhs).to(dev))
    for step in range(3):
        x, iref = solver.step()
        assert infos[step]["status"] == iref["status"] and infos[step]["iterations"] == iref["iterations"], (step, infos[step], iref)
    x, vr, ar = x.cpu().numpy(), ti.v_prev().cpu().numpy(), ti.a_prev().cpu().numpy()
    eng.close()
    assert np.abs(xg - x).max() <= 1e-8 * np.abs(x).max()
    assert np.abs(vg - vr).max() <= 1e-7 * np.abs(vr).max()
    assert np.abs(ag - ar).max() <= 1e-6 * np.abs(ar).max()

