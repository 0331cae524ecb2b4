#!/usr/bin/env python3
"""Kernel micro-benchmarks on the GPU box: per-kernel average duration (HIP events) and algorithmic GB/s.
usage: python tools/bench_kernels.py [--n 55] [--model elastic|neo|laplace] [--reps 20] [--numbering lattice]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyfem_amd import engine as E  # noqa: E402
from polyfem_amd import mesh as M  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=55)
ap.add_argument("--order", type=int, default=2)
ap.add_argument("--model", default="elastic")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--numbering", default="lattice")
ap.add_argument("--hex", action="store_true")
a = ap.parse_args()

mesh = M.cube_hexes(a.n, a.order, a.numbering) if a.hex else M.cube_tets(a.n, a.order, a.numbering)
eng = E.Engine(0)
eng.set_mesh(mesh)
model = {"elastic": E.LINEAR_ELASTICITY, "neo": E.NEOHOOKEAN, "laplace": E.LAPLACIAN}[a.model]
eng.set_material(model, 57692.3, 38461.5)
nnz = eng.build_pattern()
inf = eng.info()
n = inf.n_dofs
u = M.seeded_displacement(mesh, inf.n_dofs // inf.n_nodes, 1e-2 / a.n)
x = np.random.default_rng(0).standard_normal(n)
# warmup
if model == E.NEOHOOKEAN:
    eng.hessian(u, fetch=False)
    eng.gradient(u)
    eng.energy(u)
else:
    eng.assemble_linear()
eng.spmv(x)
eng.timing_reset()
eng.timing_enable(True)
for _ in range(a.reps):
    if model == E.NEOHOOKEAN:
        eng.hessian(u, fetch=False)
        eng.gradient(u)
        eng.energy(u)
    else:
        eng.assemble_linear()
    eng.spmv(x)
eng.timing_enable(False)
sb = 8.0 * inf.nnz + 4.0 * inf.n_blocks + 8.0 * (inf.n_nodes + 1) + 16.0 * n
ab = 4.0 * inf.n_loc * inf.n_el + 4.0 * mesh.gconn.shape[1] * inf.n_el + 24.0 * mesh.xyz.shape[0] + 8.0 * inf.nnz
print(f"mesh n={a.n} el={inf.n_el} nodes={inf.n_nodes} dofs={n} nnz={inf.nnz} blocks={inf.n_blocks} dev_bytes={inf.bytes_device/1e9:.2f}GB "
      f"env LPN={os.environ.get('PFX_SPMV_LPN','-')}")
tot_asm = 0.0
for fam in ("local", "gather", "local_grad", "gather_vec", "local_energy", "spmv"):
    ms, cnt = eng.timing_get(fam)
    if cnt == 0:
        continue
    extra = ""
    if fam == "spmv":
        extra = f"  {sb/ms/1e6:8.1f} GB/s algorithmic ({sb/ms/1e6/8000*100:.1f}% of 8 TB/s)"
    if fam in ("local", "gather"):
        tot_asm += ms
    print(f"  {fam:14s} {ms:9.4f} ms  x{cnt}{extra}")
if tot_asm > 0:
    print(f"  assembly total {tot_asm:9.4f} ms -> {inf.n_el/tot_asm/1e3:.2f} M elements/s, {ab/tot_asm/1e6:.1f} GB/s algorithmic")
eng.close()
