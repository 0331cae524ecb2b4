#!/usr/bin/env python3
"""SpMV timing on device vectors: python tools/exp/spmv_time.py [grid] [reps]  (PFX_LIBRARY selects a build)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from polyfem_amd import engine as E, distributed as D
n = int(sys.argv[1]) if len(sys.argv) > 1 else 55
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
pre = float(os.environ.get("PRE_GB", "0"))
if pre > 0:  # touch and release a large range first (does the history of the allocator change the product's speed?)
    big = torch.empty(int(pre * 1e9), dtype=torch.uint8, device="cuda"); big.fill_(1); torch.cuda.synchronize(); del big; torch.cuda.empty_cache()
part, _, _ = D.slab_part(n, n, 0, 1, 2, "tet")
eng = E.Engine(0); eng.set_mesh(part.mesh); eng.set_material(E.LINEAR_ELASTICITY, 57692.3, 38461.5); eng.build_pattern(); eng.assemble_linear()
inf = eng.info()
x = torch.randn(inf.n_dofs, dtype=torch.float64, device="cuda")
y = torch.zeros_like(x)
for _ in range(3):
    eng.spmv_dev(x.data_ptr(), y.data_ptr())
eng.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    eng.spmv_dev(x.data_ptr(), y.data_ptr())
eng.synchronize()
ms = (time.perf_counter() - t0) / reps * 1e3
stored = 8.0 * inf.nnz + 4.0 * inf.n_blocks + 8.0 * (inf.n_nodes + 1) + 16.0 * inf.n_dofs
print(f"grid {n}: spmv {ms:.4f} ms, stored bytes {stored/1e9:.3f} GB -> {stored/ms/1e6:.0f} GB/s; checksum {float(y.abs().sum()):.6e}")
eng.close()
