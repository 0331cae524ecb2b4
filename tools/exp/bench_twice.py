"""Runs bench.run_workload for the LinearElasticity 55^3 configuration several times in one process (does the product's
speed depend on what ran before?): python tools/exp/bench_twice.py [first_grid]"""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
import bench
from polyfem_amd import engine as E, mesh as M, distributed as D
args = argparse.Namespace(cg_max_iter=1000, cg_tol=1e-12, precond=1, no_amg=True, amg_tol=1e-10, amg_ncycle=2, amg_degree=1, dt=1e-3, cpu_n=40, cpu_cg_iters=15)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
if first:
    args.no_amg = not os.environ.get("FIRST_AMG")
    out = bench.run_workload(args, "neohookean", first, "p2tet", int(os.environ.get("FIRST_STEPS", "1")), 1, 1, 0, 0, dev, E, M, D, torch, dist)
    args.no_amg = True
    print("first: neohookean grid", first, out["cg"]["spmv_ms"], flush=True)
for k in range(3):
    out = bench.run_workload(args, "linear_elasticity", 55, "p2tet", 3, 1, 1, 0, 0, dev, E, M, D, torch, dist)
    print("LE 55 run", k, "ms_per_step", out["ms_per_step"], "spmv_ms", out["cg"]["spmv_ms"], flush=True)
