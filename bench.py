#!/usr/bin/env python3
"""Benchmark of the assembly + CG hot path on MI355X.

Default workload (the north-star configuration): NeoHookean hyperelasticity, P2 tetrahedra on the linear geometry of a
119^3 Kuhn-split grid PER GPU (10 110 954 elements, 41 M dofs, 3.5e9 non-zeros), E = 2e4, nu = 0.3, at a seeded
displacement.  One STEP = gradient + Hessian assembly into the resident pattern (NLAssembler::assemble_grad /
assemble_hessian) + one Jacobi-CG solve with the reference's defaults (tolerance 1e-12, max_iter 1000; it runs into
max_iter, exactly as the reference would).  `value` = elements through the whole step per second.  Inputs (mesh,
pattern, vectors) are resident in HBM before the timed region.

`--gpus N` (N > 1) starts N ranks itself when it was not started by torch.distributed.run: the parent spawns N child
processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) before it has touched the GPU, and waits.

Extra objects in the JSON line:
  roofline      dominant kernel of the step (the CG SpMV): the bytes its matrix format must move per launch / average
                launch duration (HIP events on the engine's stream) against the 8 TB/s HBM peak -- a fraction that cannot
                exceed one; the same launch priced on a scalar CSR's bytes is kept as csr_equivalent_*; traffic = HBM
                bytes per launch from the committed rocprofv3 PMC passes of this same workload and grid, quoted only when
                that recording was made with the library sources this run is built from (hash compared), else null
  assembly      elements/s and the same roofline accounting for the assembly kernels
  cpu_baseline  the CPU oracle (reference-algorithm restatement) on a bounded sample of the same workload
  linear_elasticity_1M   (N = 1 only) the same measurement for the LinearElasticity stiffness + CG configuration on the
                55^3 grid (998 250 P2 tets), a few steps
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0
FP64_PEAK_TFLOPS = 78.6  # MI355X vector FP64 (multiply-add), the rate the assembly kernels can use


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def spmv_csr_bytes(nnz, n_dofs):
    """What a scalar CSR product y = A x would have to move for the same matrix: 8-B value and 4-B column per stored
    entry, the 4-B row pointers, x read once and y written once.  Reported as csr_equivalent_*; the node-block format
    never moves 8 of every 9 of those column indices, so a rate on these bytes can exceed the peak and is no roofline
    fraction."""
    return 12.0 * nnz + 4.0 * (n_dofs + 1) + 16.0 * n_dofs


def spmv_bytes(nnz, n_blocks, n_nodes, n_dofs):
    """Algorithmic bytes of one product y = A x in the node-block format the solver holds (padding excluded), the figure
    the roofline fraction is quoted on: 8 B per value, ONE 4-B column per 3x3 block, the 8-B slice bases, x read once
    and y written once.  Every one of these bytes has to cross the HBM interface once per launch (the matrix is far
    larger than the caches), so bytes / time is bounded by the peak."""
    return 8.0 * nnz + 4.0 * n_blocks + 8.0 * (n_nodes + 1) + 16.0 * n_dofs


def assembly_bytes(n_el, n_loc, n_gloc, n_gnodes, n_nodes, nnz, nonlinear):
    """Compulsory traffic of one assembly (indices 4 B, values 8 B): conn + gconn + geometry (+ displacement) read
    once, every matrix entry written once."""
    b = 4.0 * n_loc * n_el + 4.0 * n_gloc * n_el + 24.0 * n_gnodes + 8.0 * nnz
    if nonlinear:
        b += 24.0 * n_nodes
    return b


def assembly_flops(n_el, n_loc, n_pts, nonlinear):
    """FP64 operations of one row-owner assembly: every pair (i >= j) of an element is evaluated once (its mirror image
    is a transposed store).  Per quadrature point (counted from the kernels' source, a multiply-add = 2): linear
    elasticity 94 (separate multiplies and adds, the path is compiled without contraction), NeoHookean 93 (the products
    that do not depend on the pair come ready in the element state; 150 while the functor formed them itself)."""
    return float(n_el) * (n_loc * (n_loc + 1) // 2) * n_pts * (93.0 if nonlinear else 94.0)


def cpu_baseline(args, workload, hexes, oracle, M):
    """Oracle on a bounded sample of the same workload, on this host's cores: the assembly runs on `cores` threads (static
    chunks standing in for the reference's TBB loop); the CG is single-threaded like the reference's (Eigen's
    ConjugateGradient on a column-major matrix multiplies on one thread) and timed on a few iterations, scaled."""
    n = max(args.cpu_n // 2, 2) if hexes else args.cpu_n
    mesh = M.cube_hexes(n, 2) if hexes else M.cube_tets(n, 2)
    # every core this process may use: its affinity mask, cut by the share the machine grants a job (OMP_NUM_THREADS is
    # set to that share on the GPU box, where os.cpu_count() shows the whole machine's cores)
    host_cores = os.cpu_count() or 1
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else host_cores
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    cores = max(1, min(avail, share) if share > 0 else avail)
    om = oracle.OMesh(mesh)
    t0 = time.perf_counter()
    if workload == "neohookean":
        lam, mu = oracle.lame(2e4, 0.3)
        x = M.seeded_displacement(mesh, 3, 1e-2 / n)
        K = oracle.assemble_hessian(om, oracle.NEOHOOKEAN, lam, mu, x, cores)
        oracle.assemble_gradient(om, oracle.NEOHOOKEAN, lam, mu, x, cores)
    else:
        lam, mu = oracle.lame(1e5, 0.3)
        K = oracle.assemble_linear(om, oracle.LINEAR_ELASTICITY, lam, mu, cores)
    t_asm = time.perf_counter() - t0
    bd = M.boundary_dofs(mesh, 3, "x0")
    Kr = oracle.full_to_reduced_matrix(K, bd)
    nr = Kr[0].size - 1
    b = np.random.default_rng(1).standard_normal(nr)
    iters = args.cpu_cg_iters
    t0 = time.perf_counter()
    _, it, _ = oracle.cg(Kr, b, jacobi=True, max_iter=iters, tol=1e-12)
    t_cg = (time.perf_counter() - t0) * (args.cg_max_iter / max(it, 1))
    return {
        "value": mesh.n_el / (t_asm + t_cg), "unit": "elements/s", "cores": cores, "kind": "port",
        "host_cores": host_cores, "threads_assembly": cores, "threads_cg": 1,
        "thread_limit": (f"OMP_NUM_THREADS={share} (this job's share of the host)" if share > 0 and share < avail else "affinity mask of the process"),
        "sample": f"{'Q2 hexes' if hexes else 'P2 tets'} {n}^3 grid ({mesh.n_el} elements, {nr} free dofs, {Kr[1].size} non-zeros: out of "
                  f"cache): {'gradient + Hessian' if workload == 'neohookean' else 'stiffness'} assembly with {cores} threads "
                  f"{t_asm:.3f}s + {it} single-thread CG iterations scaled to {args.cg_max_iter} ({t_cg:.3f}s)",
        "assembly_elements_per_s": mesh.n_el / t_asm, "assembly_threads": cores, "cg_threads": 1,
    }


def run_workload(args, workload, grid, element, steps, warmup, world, rank, local_rank, dev, E, M, D, torch, dist):
    """Sets the workload up on this rank, times `steps` steps and returns the result dictionary (rank 0; None elsewhere)."""
    nonlinear = workload in ("neohookean", "newton", "transient")
    hexes = element == "q2hex"
    t_setup = time.perf_counter()
    # weak scaling: the global grid is n x n x (n*world) cells; every rank owns the node planes of n cell layers
    # (plus one layer of its lower neighbour's elements, which it assembles redundantly instead of communicating)
    cpr = args.cells_per_rank or grid
    part, bounds, n_nodes_global = D.slab_part(grid, cpr, rank, world, 2, "hex" if hexes else "tet")
    mesh = part.mesh
    if rank == 0:
        log(f"{workload} grid {grid}: mesh of {mesh.conn.shape[0]} elements built on the host in {time.perf_counter() - t_setup:.1f}s")
    eng = E.Engine(local_rank)
    eng.set_mesh(mesh)
    if nonlinear:
        lam = (2e4 * 0.3) / ((1 + 0.3) * (1 - 0.6))
        mu = 2e4 / (2 * 1.3)
        eng.set_material(E.NEOHOOKEAN, lam, mu)
    else:
        lam = (1e5 * 0.3) / ((1 + 0.3) * (1 - 0.6))
        mu = 1e5 / (2 * 1.3)
        eng.set_material(E.LINEAR_ELASTICITY, lam, mu)
    t0 = time.perf_counter()
    eng.build_pattern()
    eng.synchronize()
    t_pattern = time.perf_counter() - t0
    inf = eng.info()
    eng.set_owned(part.n_owned, part.elem_owned)
    fixed_nodes = np.nonzero(np.abs(mesh.node_xyz[:, 0]) < 1e-12)[0]
    fixed_dofs = np.sort((fixed_nodes[:, None] * 3 + np.arange(3)[None]).reshape(-1)).astype(np.int32)
    eng.set_dirichlet(fixed_dofs)

    n = inf.n_dofs
    # torch owns the vectors; the engine enqueues on the same (non-default) stream, so no cross-stream syncs
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    eng.set_stream(stream.cuda_stream)
    l2g = part.local_to_global()

    def global_field(seed, amp):
        # value of a globally defined pseudo-random field at this rank's nodes (same on every rank for shared nodes)
        h = (l2g[:, None] * 3 + np.arange(3)[None]).reshape(-1).astype(np.uint64)
        z = (h + np.uint64(seed)) * np.uint64(0x9E3779B97F4A7C15)
        z ^= z >> np.uint64(31)
        z *= np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(29)
        return ((z >> np.uint64(11)).astype(np.float64) / float(1 << 53) - 0.5) * 2.0 * amp

    u = torch.from_numpy(global_field(12345, 1e-2 / grid)).to(dev)
    rhs = torch.from_numpy(global_field(1, 1.0)).to(dev)
    x = torch.zeros(n, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()

    comm = D.Comm()
    dcg = D.DistributedCG(D.HipOps(eng), part, 3, comm, dev) if world > 1 and workload != "newton" else None
    halo_bytes = dcg.halo.bytes_per_exchange(3) if dcg is not None else 0

    newton_info = {}
    load = torch.zeros(n, dtype=torch.float64, device=dev)
    load[2::3] = -50.0 / (n_nodes_global / world)
    dnewton = (D.DistributedNewton(D.HipOps(eng), part, 3, comm, dev, fixed_dofs)
               if world > 1 and workload == "newton" else None)
    if dnewton is not None:
        halo_bytes = dnewton.cg.halo.bytes_per_exchange(3)

    if workload == "transient":
        rho = 1000.0
        eng.inertia_set_mass(rho)
        gravity = torch.from_numpy(eng.assemble_rhs(np.array([0.0, 0.0, -9.81]), rho)).to(dev)
        vel = torch.zeros(n, dtype=torch.float64, device=dev)
        acc = torch.zeros(n, dtype=torch.float64, device=dev)

    def step():
        if workload == "transient":
            # consecutive time steps of one simulation: x and vel carry over
            newton_info.update(eng.implicit_euler_step_dev(args.dt, x.data_ptr(), vel.data_ptr(), acc.data_ptr(), gravity.data_ptr(),
                                                           cg_max_iter=args.cg_max_iter, cg_tol=args.cg_tol, cg_precond=args.precond))
            return newton_info["cg_iterations"], newton_info["grad_norm"]
        if dnewton is not None:
            x.zero_()
            newton_info.update(dnewton.solve(x, load, cg_max_iter=args.cg_max_iter, cg_tol=args.cg_tol, cg_check_every=50, cg_precond=args.precond))
            return newton_info["cg_iterations"], newton_info["grad_norm"]
        if workload == "newton":
            x.zero_()
            newton_info.update(eng.newton_solve_dev(x.data_ptr(), load.data_ptr(), cg_max_iter=args.cg_max_iter, cg_tol=args.cg_tol,
                                                    cg_precond=args.precond))
            return newton_info["cg_iterations"], newton_info["grad_norm"]
        if nonlinear:
            eng.hessian_dev(u.data_ptr())
            eng.gradient_dev(u.data_ptr(), x.data_ptr())
        else:
            eng.assemble_linear()
        x.zero_()
        if dcg is not None:
            return dcg.solve(rhs, x, args.precond, args.cg_max_iter, args.cg_tol, check_every=50)
        return eng.cg_dev(rhs.data_ptr(), x.data_ptr(), args.precond, args.cg_max_iter, args.cg_tol)

    if rank == 0:
        log(f"set-up done in {time.perf_counter() - t_setup:.1f}s (pattern {t_pattern:.2f}s, nnz {inf.nnz}); {warmup} warm-up + {steps} timed steps")
    t_first = None  # the first step also allocates the solver's arrays and searches a placement for the value array
    for k in range(warmup):
        t_w = time.perf_counter()
        step()
        eng.synchronize()
        if k == 0:
            t_first = time.perf_counter() - t_w
        if rank == 0:
            log(f"warm-up step {k + 1}/{warmup}")
    eng.synchronize()
    eng.timing_reset()
    # HIP events around every 16th launch of each kernel family: timing every launch of the sub-millisecond CG kernels
    # costs 2.5 % of the step
    timed_every = int(os.environ.get("PFX_BENCH_TIMED_EVERY", "16"))
    eng.timing_enable(timed_every)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    it = err = None
    t_last = t0
    for k in range(steps):
        it, err = step()
        if rank == 0 and time.perf_counter() - t_last > 20.0:  # a progress line now and then (stderr; no synchronisation)
            t_last = time.perf_counter()
            log(f"timed step {k + 1}/{steps} enqueued")
    eng.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    eng.timing_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    spmv_ms, spmv_n = eng.timing_get("spmv")
    spmv_b_ms, _ = eng.timing_get("spmv_boundary")  # N > 1: the rows that wait for the halo exchange are a second launch
    spmv_ms += spmv_b_ms
    loc_ms, loc_n = eng.timing_get("local")
    gat_ms, gat_n = eng.timing_get("gather")
    lg_ms, _ = eng.timing_get("local_grad")
    gv_ms, _ = eng.timing_get("gather_vec")
    upd_ms, _ = eng.timing_get("cg_update")
    dir_ms, _ = eng.timing_get("cg_dir")
    conv_ms, _ = eng.timing_get("sell_convert")
    pre_ms, pre_n = eng.timing_get("precond")

    sb = spmv_bytes(inf.nnz, inf.n_blocks, inf.n_nodes, n)
    sb_csr = spmv_csr_bytes(inf.nnz, n)
    ab = assembly_bytes(inf.n_el, inf.n_loc, 8 if hexes else 4, mesh.xyz.shape[0], inf.n_nodes, inf.nnz, nonlinear)
    spmv_gbs = sb / (spmv_ms * 1e-3) / 1e9 if spmv_ms > 0 else 0.0
    af = assembly_flops(inf.n_el, inf.n_loc, inf.n_pts, nonlinear)
    asm_ms = loc_ms + gat_ms
    asm_gbs = ab / (asm_ms * 1e-3) / 1e9 if asm_ms > 0 else 0.0

    # HBM bytes per SpMV launch from the committed PMC passes (FETCH_SIZE / WRITE_SIZE, one rocprofv3 run each, corrected
    # as the gfx950 notes prescribe: 2 * FETCH_SIZE + WRITE_SIZE) of this same workload and grid.  A recording carries the
    # hash of the library sources it was made with; it is quoted only if this run is built from the same sources.  Null
    # when no recording matches or the run is partitioned.
    traffic, traffic_src = None, None
    if world == 1 and workload in ("neohookean", "linear_elasticity"):
        from polyfem_amd.buildinfo import source_sha256
        here = source_sha256()
        for rnd in ("r3", "r2"):
            name = f"{rnd}_pmc_traffic_{workload}_{element}_grid{grid}.json"
            tpath = os.path.join(ROOT, "profiles", name)
            if not os.path.exists(tpath):
                continue
            recs = json.load(open(tpath))
            rec = next((v for k, v in recs.items() if k.startswith("k_sell_spmv<3, true")), None)  # the CG's product (with the p.Ap partials)
            made_with = recs.get("_meta", {}).get("source_sha256")
            if rec and made_with == here:
                traffic, traffic_src = rec["hbm_bytes"], os.path.relpath(tpath, ROOT)
            elif rec:
                traffic_src = f"{os.path.relpath(tpath, ROOT)} was recorded with other library sources: not quoted"
            break

    n_el_global = (1 if hexes else 6) * grid * grid * cpr * world
    owned_el = int(part.elem_owned.sum())
    dims = f"{grid}^3" if cpr == grid else f"{grid} x {grid} x {cpr}"
    shape = f"Q2 hexes, {dims} grid per GPU" if hexes else f"P2 tets, {dims} Kuhn grid per GPU"
    backend = (dist.get_backend() if world > 1 else None)
    out = None
    if rank == 0:
        pname = {E.PRECOND_IDENTITY: "none", E.PRECOND_JACOBI: "Jacobi", E.PRECOND_AMG: "AMG"}.get(args.precond, str(args.precond))
        out = {
            "metric": "elements/s Hessian assembly + CG SpMV GB/s, P2 elasticity tets, 1-8 GPUs",
            "value": n_el_global * steps / elapsed,
            "unit": "elements/s",
            "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": 1e3 * elapsed / steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": ("NeoHookean gradient + Hessian" if nonlinear else "LinearElasticity stiffness")
                            + f" assembly + {pname}-CG (tol {args.cg_tol:g}, max_iter {args.cg_max_iter}), {shape}",
                "elements": n_el_global, "elements_per_gpu_owned": owned_el, "elements_per_gpu_assembled": inf.n_el,
                "dofs_per_gpu": part.n_owned * 3, "nnz_per_gpu": inf.nnz, "cg_iters": it, "cg_rel_residual": err,
                "preconditioner": pname,
                "parallelism": (f"row/element partition over {world} GPUs (z slabs), halo exchange + all-reduce over torch.distributed "
                                f"backend '{backend}'" + (" (RCCL over xGMI)" if backend == "nccl" else " -- NOT RCCL: single-card rehearsal of the N > 1 path"))
                               if world > 1 else "single GPU",
                "halo_bytes_per_exchange_rank0": halo_bytes,
                # who runs the distributed iteration: the library's own loop (pfx_dist_cg_dev) over its RCCL communicator, or over
                # a host-staged transport when the process group is gloo (single-card rehearsal)
                "distributed_iteration": (f"inside the library (pfx_dist_cg_dev), communicator: {dcg.lib.libcomm.kind}" if dcg is not None and dcg.lib is not None
                                          else f"inside the library (pfx_dist_newton_dev), communicator: {dnewton.cg.lib.libcomm.kind}" if dnewton is not None and dnewton.cg.lib is not None
                                          else None),
                "bytes_device": eng.info().bytes_device,  # everything the context holds now: pattern, values, solver copy, vectors
            },
            "roofline": {"kernel": "k_sell_spmv (CG matrix-vector product)", "bound": "hbm", "achieved": spmv_gbs,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": spmv_gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_src, "avg_ms": spmv_ms, "timed_launches": spmv_n, "timed_every": timed_every,
                         # bytes of the node-block format (8 B per value, one 4-B column per 3x3 block, x and y once): what
                         # `achieved` = algorithmic_bytes / avg_ms is priced on
                         "algorithmic_bytes": sb,
                         # the measured HBM traffic of the same launch as a rate and as a fraction of the peak
                         "hbm_rate": traffic / (spmv_ms * 1e-3) / 1e9 if traffic and spmv_ms > 0 else None,
                         "moved_frac": traffic / (spmv_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if traffic and spmv_ms > 0 else None,
                         # for comparison with scalar-CSR products: the bytes such a product would move for this matrix
                         # (12 B per entry) over the same time; may exceed the peak, since 8 of 9 column indices do not exist here
                         "csr_equivalent_bytes": sb_csr,
                         "csr_equivalent_rate": sb_csr / (spmv_ms * 1e-3) / 1e9 if spmv_ms > 0 else 0.0},
            "assembly": {"elements_per_s": world * owned_el / (asm_ms * 1e-3) if asm_ms > 0 else 0.0, "local_ms": loc_ms,
                         "gather_ms": gat_ms, "solver_copy_ms": conv_ms, "gradient_local_ms": lg_ms, "gradient_gather_ms": gv_ms,
                         "roofline": {"bound": "hbm", "achieved": asm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": asm_gbs / HBM_PEAK_GBS, "algorithmic_bytes": ab},
                         "roofline_fp64": {"bound": "fp64 valu", "achieved": af / (asm_ms * 1e-3) / 1e12 if asm_ms > 0 else 0.0,
                                           "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                           "frac": af / (asm_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS if asm_ms > 0 else 0.0,
                                           "flops": af}},
            "cg": {"update_ms": upd_ms, "direction_ms": dir_ms, "spmv_ms": spmv_ms, "precond_ms": pre_ms,
                   "ms_per_iter": spmv_ms + upd_ms + dir_ms + pre_ms,
                   # fused vector kernels on their compulsory bytes: update reads p, A p, 1/diag, x, r and writes x, r
                   # (56 B per dof); direction reads r, 1/diag, p and writes p (32 B per dof)
                   "update_gbs": 56.0 * n / (upd_ms * 1e-3) / 1e9 if upd_ms > 0 else 0.0,
                   "direction_gbs": 32.0 * n / (dir_ms * 1e-3) / 1e9 if dir_ms > 0 else 0.0},
            "pattern_build_s": t_pattern,
            "first_step_s": t_first,
            "placement_tries": int(os.environ.get("PFX_PLACEMENT_TRIES", "6")),
        }
        if workload == "transient":
            out["config"]["workload"] = (f"NeoHookean implicit Euler time step (dt {args.dt:g}, Newton + backtracking, {pname}-CG tol {args.cg_tol:g} "
                                         f"max_iter {args.cg_max_iter}), {shape}")
            out["newton"] = dict(newton_info)
        if workload == "newton":
            out["config"]["workload"] = (f"NeoHookean Newton solve (backtracking line search, {pname}-CG tol {args.cg_tol:g} max_iter "
                                         f"{args.cg_max_iter} per step), {shape}")
            out["newton"] = dict(newton_info)
    # the same system solved to convergence with the AMG preconditioner (non-default option; elastic workloads); with
    # several ranks: one hierarchy per rank on its owned block (block-Jacobi over the ranks)
    if workload in ("neohookean", "linear_elasticity") and not args.no_amg:
        watchdog = None
        if world > 1:
            # A rank that fails inside this optional leg would leave the others waiting in a collective, and the line
            # measured above would never be printed: past the limit every rank gives up, rank 0 reports what it has.
            import threading
            limit = float(os.environ.get("PFX_BENCH_AMG_LIMIT_S", "300"))

            def give_up():
                log(f"AMG leg: no result within {limit:g} s, reporting the line without it")
                if out is not None:
                    out["amg_solve"] = {"error": f"no result within {limit:g} s"}
                    print(json.dumps(out), flush=True)
                os._exit(0)

            watchdog = threading.Timer(limit, give_up)
            watchdog.daemon = True
            watchdog.start()
        try:
            eng.timing_reset()
            eng.timing_enable(1)
            if args.amg_ncycle != 2 or args.amg_degree != 1:
                eng.amg_set_params(ncycle=args.amg_ncycle, relax_degree=args.amg_degree)
            x.zero_()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            if dcg is not None and dcg.lib is not None:
                # one hierarchy that spans the ranks: built (collectively) by its first application
                eng.cgs_set_workspace(0, 0)
                dcg.lib.amg_apply(rhs, x)
                x.zero_()
            else:
                eng.amg_setup()
            eng.synchronize()
            t_setup_amg = time.perf_counter() - t0
            t_tune = 0.0
            if world == 1 and not args.no_amg_tune:
                # a simulation that solves many times on this pattern asks for the placement search of the hierarchy's
                # single-precision copies once (pfx_tune_placement); charged to the set-up below
                t0 = time.perf_counter()
                eng.tune_placement()
                eng.synchronize()
                t_tune = time.perf_counter() - t0
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            if dcg is not None:
                it_a, err_a = dcg.solve(rhs, x, E.PRECOND_AMG, args.cg_max_iter, args.amg_tol)
            else:
                it_a, err_a = eng.cg_dev(rhs.data_ptr(), x.data_ptr(), E.PRECOND_AMG, args.cg_max_iter, args.amg_tol)
            eng.synchronize()
            t_solve = time.perf_counter() - t0
            if world > 1:
                tt = torch.tensor([t_solve, t_setup_amg], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                t_solve, t_setup_amg = (float(v) for v in tt.tolist())
            eng.timing_enable(False)
            if out is not None:
                ai = eng.amg_info()
                pre_ms, pre_n = eng.timing_get("precond")
                out["amg_solve"] = {"preconditioner": "aggregation AMG (PFX_PRECOND_AMG), " + ("V" if args.amg_ncycle == 1 else "W") + f" cycle, Chebyshev degree {args.amg_degree}, coarse corrections x 1.6"
                                                      + (f", ONE hierarchy that spans the {world} ranks (aggregates local to a rank, Galerkin products across the partition boundaries, "
                                                         "coarse levels row-distributed, coarsest matrix gathered)" if world > 1 else ""),
                                    "tolerance": args.amg_tol, "cg_iters": it_a, "cg_rel_residual": err_a, "solve_ms": 1e3 * t_solve, "setup_ms": 1e3 * t_setup_amg,
                                    "placement_search_ms": 1e3 * t_tune,
                                    "symbolic_s": ai["symbolic_s"], "numeric_s": ai["numeric_s"], "levels": ai["n_levels"], "vertices": ai["n_vertices"],
                                    "operator_complexity": ai["operator_complexity"], "precond_apply_ms": pre_ms,
                                    "elements_per_s_assembly_plus_converged_solve": n_el_global / (1e-3 * (asm_ms + lg_ms + gv_ms + conv_ms) + t_solve),
                                    # the same with the whole hierarchy set-up (symbolic: once per pattern; numeric: once per matrix) charged to this one solve
                                    "elements_per_s_assembly_setup_and_converged_solve": n_el_global / (1e-3 * (asm_ms + lg_ms + gv_ms + conv_ms) + t_setup_amg + t_tune + t_solve)}
                log(f"AMG-PCG: {it_a} iterations, rel. residual {err_a:.2e}, solve {t_solve:.3f}s, setup {t_setup_amg:.3f}s")
        except Exception as exc:  # the reported line must not depend on this optional leg (all ranks fail alike: same data)
            log(f"AMG leg failed: {exc!r}")
            if out is not None:
                out["amg_solve"] = {"error": repr(exc)}
        finally:
            if watchdog is not None:
                watchdog.cancel()
    if dcg is not None:
        dcg.close()
    if dnewton is not None:
        dnewton.cg.close()
    eng.close()
    del u, rhs, x, load
    torch.cuda.empty_cache()
    return out


def dry_run(args):
    """Per-rank device memory of a `--gpus N` run, measured on one GPU: the ranks' real meshes and patterns are built one after
    the other (no assembly, no communication).  The AMG hierarchy cannot be built without matrix values; its share is
    estimated from the operator complexity measured on this library's hierarchies (1.25: the coarse matrices and their
    solver copies add a quarter of the fine level's, the transfer lists about as much as the fine level's column indices)."""
    import torch

    from polyfem_amd import distributed as D
    from polyfem_amd import engine as E

    hexes = args.element == "q2hex"
    world, grid = args.gpus, args.n
    cpr = args.cells_per_rank or grid
    total = torch.cuda.get_device_properties(0).total_memory
    ranks = sorted(set([0, world // 2, world - 1]))
    rows = []
    for r in ranks:
        part, bounds, n_nodes_global = D.slab_part(grid, cpr, r, world, 2, "hex" if hexes else "tet")
        eng = E.Engine(0)
        eng.set_mesh(part.mesh)
        eng.set_material(E.NEOHOOKEAN if args.workload != "linear_elasticity" else E.LINEAR_ELASTICITY, 1.0, 1.0)
        eng.build_pattern()
        eng.set_owned(part.n_owned, part.elem_owned)
        inf = eng.info()
        b_pattern = inf.bytes_device
        n = inf.n_dofs
        x = torch.zeros(n, dtype=torch.float64, device="cuda")
        b = torch.zeros(n, dtype=torch.float64, device="cuda")
        eng.cg_dev(b.data_ptr(), x.data_ptr(), E.PRECOND_JACOBI, 0, 1e-12)  # zero iterations: allocates the solver copy and the vectors
        eng.synchronize()
        b_solver = eng.info().bytes_device
        values = 8.0 * inf.nnz
        amg_est = 0.25 * (2.0 * values) + 2.0 * 4.0 * inf.n_blocks
        caller = 8.0 * n * 8  # vectors a driver keeps beside the context (u, rhs, x, gradient, direction, work copies)
        held = b_solver + amg_est + caller
        rows.append({"rank": r, "elements_assembled": inf.n_el, "elements_owned": int(part.elem_owned.sum()), "nodes_local": inf.n_nodes,
                     "nodes_owned": part.n_owned, "halo_nodes": int(part.halo_global.size), "nnz": inf.nnz, "node_blocks": inf.n_blocks,
                     "bytes_after_pattern": b_pattern, "bytes_with_solver_copy_and_cg_vectors": b_solver,
                     "bytes_amg_hierarchy_estimate": amg_est, "bytes_caller_vectors_estimate": caller,
                     "bytes_total_estimate": held, "fraction_of_device_memory": held / total})
        log(f"rank {r}: {inf.n_el} elements, nnz {inf.nnz}, {held / 2**30:.1f} GiB of {total / 2**30:.0f} GiB")
        eng.close()
        del x, b
        torch.cuda.empty_cache()
    worst = max(rw["fraction_of_device_memory"] for rw in rows)
    print(json.dumps({"dry_run": True, "n_gpus": world, "element": args.element, "grid": [grid, grid, cpr * world],
                      "elements_global": (1 if hexes else 6) * grid * grid * cpr * world, "device_memory_bytes": total,
                      "ranks": rows, "fits": worst < 0.9, "largest_fraction": worst}), flush=True)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n):
    """Parent of a `--gpus N` run that was not started by a launcher: N children of this same command line, one per
    rank.  The parent never initialises the GPU; it waits for the children and exits with the first non-zero code."""
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   LOCAL_WORLD_SIZE=str(n))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in pending:  # one rank failed: the others would wait for it forever
                    q.terminate()
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-amg-tune", action="store_true", help="AMG leg: do not ask for the placement search of the hierarchy's single-precision copies")
    ap.add_argument("--grid", dest="n", type=int, default=None,
                    help="grid cells per side and GPU (default 119 -> 10 110 954 P2 tets; 55 -> 998 250; q2hex: 100)")
    ap.add_argument("--element", default="p2tet", choices=["p2tet", "q2hex"],
                    help="q2hex: n^3 Q2 hexahedra per GPU on trilinear geometry (--grid 100 is the Q2 configuration of the scope table)")
    ap.add_argument("--workload", default="neohookean", choices=["linear_elasticity", "neohookean", "newton", "transient"],
                    help="newton: one step = a full NeoHookean Newton solve (gravity-like load, clamped x = 0 face); "
                         "transient: one step = one implicit Euler time step of the same body falling under gravity (single GPU)")
    ap.add_argument("--dt", type=float, default=1e-3, help="time step of the transient workload")
    ap.add_argument("--cg-max-iter", type=int, default=1000)
    ap.add_argument("--cg-tol", type=float, default=1e-12)
    ap.add_argument("--precond", type=int, default=1, help="0 none, 1 Jacobi (the reference's default for Eigen::ConjugateGradient), 2 aggregation AMG")
    ap.add_argument("--no-amg", action="store_true", help="skip the converged AMG-preconditioned solve that is reported beside the default line")
    ap.add_argument("--amg-tol", type=float, default=1e-10)
    ap.add_argument("--amg-ncycle", type=int, default=2)
    ap.add_argument("--amg-degree", type=int, default=1)
    ap.add_argument("--cpu-n", type=int, default=40, help="grid of the CPU baseline sample (40 -> 384 000 P2 tets, 1.6 M dofs)")
    ap.add_argument("--cpu-cg-iters", type=int, default=15)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the LinearElasticity 55^3 measurement that rides along at N = 1")
    ap.add_argument("--cells-per-rank", type=int, default=0,
                    help="cell layers (z) per GPU when they differ from --grid: the global grid is grid x grid x (cells-per-rank * gpus)")
    ap.add_argument("--dry-run", action="store_true",
                    help="memory accounting of a --gpus N run on ONE GPU, without assembling anything: for the first, a middle and the last rank "
                         "of the partition, builds that rank's mesh and pattern, allocates the solver copy and the CG vectors, and prints what the "
                         "rank holds against the GPU's memory")
    ap.add_argument("--spawn-check", action="store_true", help="every rank reports its rank/world and exits (no GPU work): checks the launcher")
    args = ap.parse_args()
    if args.n is None:
        args.n = 100 if args.element == "q2hex" else (119 if args.workload == "neohookean" else 55)

    if args.dry_run:
        return dry_run(args)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE={world} of the launcher")
    if args.spawn_check:
        if rank == 0:
            print(json.dumps({"spawn_check": True, "n_gpus": world, "rank": rank}), flush=True)
        return

    import torch
    import torch.distributed as dist

    from polyfem_amd import engine as E
    from polyfem_amd import mesh as M

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("PFX_BENCH_BACKEND", "nccl")  # "gloo": single-card rehearsal of the N>1 path
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from polyfem_amd import distributed as D

    if args.workload == "transient" and world > 1:
        raise SystemExit("the transient workload of bench.py is single-GPU (DistributedImplicitEuler covers the partitioned case)")
    out = run_workload(args, args.workload, args.n, args.element, args.steps, args.warmup, world, rank, local_rank, dev, E, M, D, torch, dist)
    if rank == 0:
        # the CPU column and the second configuration are reported with the single-GPU line only
        if world == 1 and args.workload == "neohookean" and args.element == "p2tet" and not args.no_secondary and args.n != 55:
            log("secondary configuration: LinearElasticity stiffness + CG, 55^3 grid")
            sec = run_workload(args, "linear_elasticity", 55, "p2tet", min(args.steps, 3), 1, 1, 0, local_rank, dev, E, M, D, torch, dist)
            out["linear_elasticity_1M"] = {k: sec[k] for k in ("value", "unit", "ms_per_step", "steps", "config", "roofline", "assembly", "cg", "amg_solve") if k in sec}
        if not args.no_cpu_baseline and world == 1 and args.workload not in ("newton", "transient"):
            from oracle import pf_oracle
            pf_oracle.build()
            log("CPU baseline (oracle) ...")
            out["cpu_baseline"] = cpu_baseline(args, args.workload, args.element == "q2hex", pf_oracle, M)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
