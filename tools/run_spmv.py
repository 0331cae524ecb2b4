#!/usr/bin/env python3
"""Minimal driver for profiling: assemble once, then run the SpMV a few times."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyfem_amd import engine as E, mesh as M
n = int(sys.argv[1]) if len(sys.argv) > 1 else 55
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
mesh = M.cube_tets(n, 2)
eng = E.Engine(0); eng.set_mesh(mesh); eng.set_material(E.LINEAR_ELASTICITY, 57692.3, 38461.5); eng.build_pattern(); eng.assemble_linear()
x = np.random.default_rng(0).standard_normal(eng.info().n_dofs)
for _ in range(reps):
    eng.spmv(x)
eng.close()
