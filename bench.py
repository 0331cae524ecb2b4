#!/usr/bin/env python3
"""Benchmark of the assembly + CG hot path on MI355X.

Default workload: linear elasticity, P2 tetrahedra on the linear geometry of a 55^3
Kuhn-split unit cube (998 250 elements, 4.1 M dofs), E = 1e5, nu = 0.3.  One STEP = stiffness assembly into the
resident pattern + one Jacobi-CG solve with the reference's defaults (tolerance 1e-12, max_iter 1000; it runs into
max_iter, exactly as the reference would).  `value` = elements through the whole step per second.  Inputs (mesh,
pattern, right-hand side) are resident in HBM before the timed region.

Extra objects in the JSON line:
  roofline      dominant kernel of the step (the CG SpMV): algorithmic bytes / average launch duration (HIP events on
                the engine's stream) against the 8 TB/s HBM peak
  assembly      elements/s and the same roofline accounting for the two assembly kernels
  cpu_baseline  the CPU oracle (reference-algorithm restatement) on a bounded sample of the same workload
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0


def spmv_bytes(nnz, n_dofs):
    """Algorithmic bytes of one product y = A x, the figure the roofline fraction is quoted on: the compulsory traffic
    of a scalar CSR product -- 8-B value and 4-B column per stored entry, the 4-B row pointers, x read once and y written
    once."""
    return 12.0 * nnz + 4.0 * (n_dofs + 1) + 16.0 * n_dofs


def spmv_stored_bytes(nnz, n_blocks, n_nodes, n_dofs):
    """Bytes the node-block format of this library has to move for the same product (padding excluded): 8 B per value,
    ONE 4-B column per 3x3 block, the 8-B slice bases, x read once and y written once.  Smaller than spmv_bytes: the
    block format drops 8/9 of the column indices."""
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
    elasticity 94 (separate multiplies and adds, the path is compiled without contraction), NeoHookean 150."""
    return float(n_el) * (n_loc * (n_loc + 1) // 2) * n_pts * (150.0 if nonlinear else 94.0)


FP64_PEAK_TFLOPS = 78.6  # MI355X vector FP64 (multiply-add), the rate the assembly kernels can use


def cpu_baseline(args, oracle, M):
    """Oracle on a bounded sample of the same workload, on this host's cores."""
    hexes = args.element == "q2hex"
    n = max(args.cpu_n // 2, 2) if hexes else args.cpu_n
    mesh = M.cube_hexes(n, 2) if hexes else M.cube_tets(n, 2)
    lam, mu = oracle.lame(1e5, 0.3)
    cores = min(os.cpu_count() or 1, 16)
    om = oracle.OMesh(mesh)
    t0 = time.perf_counter()
    if args.workload == "neohookean":
        x = M.seeded_displacement(mesh, 3, 1e-2 / n)
        K = oracle.assemble_hessian(om, oracle.NEOHOOKEAN, lam, mu, x, cores)
    else:
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
        "sample": f"{'Q2 hexes' if hexes else 'P2 tets'} {n}^3 grid ({mesh.n_el} elements): assembly with {cores} threads {t_asm:.3f}s + "
                  f"{it} single-thread CG iterations scaled to {args.cg_max_iter} ({t_cg:.3f}s)",
        "assembly_elements_per_s": mesh.n_el / t_asm,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--grid", dest="n", type=int, default=55, help="grid cells per side (55 -> 998 250 P2 tets)")
    ap.add_argument("--element", default="p2tet", choices=["p2tet", "q2hex"],
                    help="q2hex: n^3 Q2 hexahedra per GPU on trilinear geometry (--grid 100 is the Q2 configuration of the scope table)")
    ap.add_argument("--workload", default="linear_elasticity", choices=["linear_elasticity", "neohookean", "newton", "transient"],
                    help="newton: one step = a full NeoHookean Newton solve (gravity-like load, clamped x = 0 face); "
                         "transient: one step = one implicit Euler time step of the same body falling under gravity (single GPU)")
    ap.add_argument("--dt", type=float, default=1e-3, help="time step of the transient workload")
    ap.add_argument("--cg-max-iter", type=int, default=1000)
    ap.add_argument("--cg-tol", type=float, default=1e-12)
    ap.add_argument("--cpu-n", type=int, default=14)
    ap.add_argument("--cpu-cg-iters", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from polyfem_amd import engine as E
    from polyfem_amd import mesh as M

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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

    nonlinear = args.workload in ("neohookean", "newton", "transient")
    if args.workload == "transient" and world > 1:
        raise SystemExit("the transient workload of bench.py is single-GPU (DistributedImplicitEuler covers the partitioned case)")
    # weak scaling: the global grid is n x n x (n*world) cells; every rank owns the node planes of n cell layers
    # (plus one layer of its lower neighbour's elements, which it assembles redundantly instead of communicating)
    hexes = args.element == "q2hex"
    part, bounds, n_nodes_global = D.slab_part(args.n, args.n, rank, world, 2, "hex" if hexes else "tet")
    mesh = part.mesh
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
    nnz = eng.build_pattern()
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

    u = torch.from_numpy(global_field(12345, 1e-2 / args.n)).to(dev)
    rhs = torch.from_numpy(global_field(1, 1.0)).to(dev)
    x = torch.zeros(n, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()

    comm = D.Comm()
    dcg = D.DistributedCG(D.HipOps(eng), part, 3, comm, dev) if world > 1 and args.workload != "newton" else None

    newton_info = {}
    load = torch.zeros(n, dtype=torch.float64, device=dev)
    load[2::3] = -50.0 / (n_nodes_global / world)
    dnewton = (D.DistributedNewton(D.HipOps(eng), part, 3, comm, dev, fixed_dofs)
               if world > 1 and args.workload == "newton" else None)

    if args.workload == "transient":
        rho = 1000.0
        eng.inertia_set_mass(rho)
        gravity = torch.from_numpy(eng.assemble_rhs(np.array([0.0, 0.0, -9.81]), rho)).to(dev)
        vel = torch.zeros(n, dtype=torch.float64, device=dev)
        acc = torch.zeros(n, dtype=torch.float64, device=dev)

    def step():
        if args.workload == "transient":
            # consecutive time steps of one simulation: x and vel carry over
            newton_info.update(eng.implicit_euler_step_dev(args.dt, x.data_ptr(), vel.data_ptr(), acc.data_ptr(), gravity.data_ptr(),
                                                           cg_max_iter=args.cg_max_iter, cg_tol=args.cg_tol))
            return newton_info["cg_iterations"], newton_info["grad_norm"]
        if dnewton is not None:
            x.zero_()
            newton_info.update(dnewton.solve(x, load, cg_max_iter=args.cg_max_iter, cg_tol=args.cg_tol, cg_check_every=50))
            return newton_info["cg_iterations"], newton_info["grad_norm"]
        if args.workload == "newton":
            x.zero_()
            newton_info.update(eng.newton_solve_dev(x.data_ptr(), load.data_ptr(), cg_max_iter=args.cg_max_iter, cg_tol=args.cg_tol))
            return newton_info["cg_iterations"], newton_info["grad_norm"]
        if nonlinear:
            eng.hessian_dev(u.data_ptr())
            eng.gradient_dev(u.data_ptr(), x.data_ptr())
        else:
            eng.assemble_linear()
        x.zero_()
        if dcg is not None:
            return dcg.solve(rhs, x, E.PRECOND_JACOBI, args.cg_max_iter, args.cg_tol, check_every=50)
        return eng.cg_dev(rhs.data_ptr(), x.data_ptr(), E.PRECOND_JACOBI, args.cg_max_iter, args.cg_tol)

    for _ in range(args.warmup):
        step()
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
    for _ in range(args.steps):
        it, err = step()
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
    upd_ms, _ = eng.timing_get("cg_update")
    dir_ms, _ = eng.timing_get("cg_dir")
    conv_ms, _ = eng.timing_get("sell_convert")

    sb = spmv_bytes(inf.nnz, n)
    sb_stored = spmv_stored_bytes(inf.nnz, inf.n_blocks, inf.n_nodes, n)
    ab = assembly_bytes(inf.n_el, inf.n_loc, 8 if hexes else 4, mesh.xyz.shape[0], inf.n_nodes, inf.nnz, nonlinear)
    spmv_gbs = sb / (spmv_ms * 1e-3) / 1e9 if spmv_ms > 0 else 0.0
    af = assembly_flops(inf.n_el, inf.n_loc, inf.n_pts, nonlinear)
    asm_ms = loc_ms + gat_ms
    asm_gbs = ab / (asm_ms * 1e-3) / 1e9 if asm_ms > 0 else 0.0

    # HBM bytes per SpMV launch from the committed PMC passes (FETCH_SIZE / WRITE_SIZE, one rocprofv3 run each) of this
    # same workload; null when no recording matches the grid or the run is partitioned.
    traffic, traffic_src = None, None
    tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f"r1_pmc_traffic_grid{args.n}.json")
    if world == 1 and not nonlinear and not hexes and os.path.exists(tpath):
        rec = json.load(open(tpath)).get("k_sell_spmv<3, true>")
        if rec:
            traffic, traffic_src = rec["hbm_bytes"], os.path.relpath(tpath, os.path.dirname(os.path.abspath(__file__)))

    n_el_global = (1 if hexes else 6) * args.n * args.n * args.n * world
    owned_el = int(part.elem_owned.sum())
    shape = f"Q2 hexes, {args.n}^3 grid" if hexes else f"P2 tets, {args.n}^3 Kuhn grid"
    if rank == 0:
        out = {
            "metric": "elements/s Hessian assembly + CG SpMV GB/s, P2 elasticity tets, 1-8 GPUs",
            "value": n_el_global * args.steps / elapsed,
            "unit": "elements/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": ("NeoHookean Hessian+gradient" if nonlinear else "LinearElasticity stiffness")
                            + f" assembly + Jacobi-CG (tol {args.cg_tol:g}, max_iter {args.cg_max_iter}), {shape}",
                "elements": n_el_global, "elements_per_gpu_owned": owned_el, "elements_per_gpu_assembled": inf.n_el,
                "dofs_per_gpu": part.n_owned * 3, "nnz_per_gpu": inf.nnz, "cg_iters": it, "cg_rel_residual": err,
                "parallelism": f"row/element partition over {world} GPUs, RCCL halo exchange + all-reduce"
                               if world > 1 else "single GPU",
            },
            "roofline": {"kernel": "k_sell_spmv (CG matrix-vector product)", "bound": "hbm", "achieved": spmv_gbs,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": spmv_gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_src, "avg_ms": spmv_ms, "timed_launches": spmv_n, "timed_every": timed_every, "algorithmic_bytes": sb,
                         # the same launch priced on what the block format stores, and on the measured HBM traffic
                         "stored_bytes": sb_stored,
                         "stored_rate": sb_stored / (spmv_ms * 1e-3) / 1e9 if spmv_ms > 0 else 0.0,
                         "stored_frac": sb_stored / (spmv_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if spmv_ms > 0 else 0.0,
                         "hbm_rate": traffic / (spmv_ms * 1e-3) / 1e9 if traffic and spmv_ms > 0 else None},
            "assembly": {"elements_per_s": world * owned_el / (asm_ms * 1e-3) if asm_ms > 0 else 0.0, "local_ms": loc_ms,
                         "gather_ms": gat_ms, "solver_copy_ms": conv_ms,
                         "roofline": {"bound": "hbm", "achieved": asm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": asm_gbs / HBM_PEAK_GBS, "algorithmic_bytes": ab},
                         "roofline_fp64": {"bound": "fp64 valu", "achieved": af / (asm_ms * 1e-3) / 1e12 if asm_ms > 0 else 0.0,
                                           "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                           "frac": af / (asm_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS if asm_ms > 0 else 0.0,
                                           "flops": af}},
            "cg": {"update_ms": upd_ms, "direction_ms": dir_ms, "spmv_ms": spmv_ms,
                   "ms_per_iter": spmv_ms + upd_ms + dir_ms},
            "pattern_build_s": t_pattern,
        }
        if args.workload == "transient":
            out["config"]["workload"] = (f"NeoHookean implicit Euler time step (dt {args.dt:g}, Newton + backtracking, Jacobi-CG tol {args.cg_tol:g} "
                                         f"max_iter {args.cg_max_iter}), {shape}")
            out["newton"] = dict(newton_info)
        if args.workload == "newton":
            out["config"]["workload"] = (f"NeoHookean Newton solve (backtracking line search, Jacobi-CG tol {args.cg_tol:g} max_iter "
                                         f"{args.cg_max_iter} per step), {shape}")
            out["newton"] = dict(newton_info)
        if not args.no_cpu_baseline and world == 1 and args.workload not in ("newton", "transient"):  # the CPU column is reported with the single-GPU line only
            from oracle import pf_oracle
            pf_oracle.build()
            out["cpu_baseline"] = cpu_baseline(args, pf_oracle, M)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
