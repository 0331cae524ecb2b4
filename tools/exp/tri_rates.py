"""Rates of the 2-D path on one GPU: NeoHookean gradient + Hessian assembly and Jacobi-CG on a grid of P2 triangles.
usage: python tools/exp/tri_rates.py [cells per side, default 1500] [order, default 2]"""
import sys
import time

import numpy as np
import torch

from polyfem_amd import engine as E
from polyfem_amd import mesh as M

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
order = int(sys.argv[2]) if len(sys.argv) > 2 else 2
t0 = time.perf_counter()
mesh = M.square_tris(n, order)
print(f"mesh: {mesh.n_el} triangles, {mesh.n_nodes} nodes ({time.perf_counter() - t0:.1f} s on the host)", flush=True)
eng = E.Engine(0)
eng.set_mesh(mesh)
lam, mu = 2e4 * 0.3 / (1 - 0.09), 2e4 / 2.6
eng.set_material(E.NEOHOOKEAN, lam, mu)
t0 = time.perf_counter()
nnz = eng.build_pattern()
eng.synchronize()
print(f"pattern: {nnz} non-zeros, {time.perf_counter() - t0:.2f} s", flush=True)
nd = 2 * mesh.n_nodes
eng.set_dirichlet(M.boundary_dofs(mesh, 2, "x0"))
u = torch.from_numpy(M.seeded_displacement(mesh, 2, 1e-4 / n)).cuda()
g = torch.zeros(nd, dtype=torch.float64, device="cuda")
b = torch.from_numpy(np.random.default_rng(0).standard_normal(nd)).cuda()
x = torch.zeros(nd, dtype=torch.float64, device="cuda")
for rep in range(2):
    eng.timing_reset()
    eng.timing_enable(1 if rep else 0)
    eng.gradient_dev(u.data_ptr(), g.data_ptr())
    eng.hessian_dev(u.data_ptr())
    x.zero_()
    it, err = eng.cg_dev(b.data_ptr(), x.data_ptr(), E.PRECOND_JACOBI, 400, 1e-30)
    eng.synchronize()
eng.timing_enable(False)
inf = eng.info()
names = ("local", "gather", "local_grad", "gather_vec", "sell_convert", "spmv", "cg_update", "cg_dir")
t = {k: eng.timing_get(k) for k in names}
for k in names:
    print(f"  {k:13s} {t[k][0]:9.4f} ms  x {t[k][1]}")
blocks = inf.n_blocks
fmt_bytes = blocks * (4 * 8 + 4) + 2 * nd * 8  # 2 x 2 values + column index per block, x read and y written once
print(f"assembly: {mesh.n_el / ((t['local'][0] + t['gather'][0]) * 1e-3) / 1e6:.1f} M elements/s; "
      f"SpMV {fmt_bytes / 1e9:.2f} GB of format bytes -> {fmt_bytes / (t['spmv'][0] * 1e-3) / 1e12:.2f} TB/s")
eng.close()
