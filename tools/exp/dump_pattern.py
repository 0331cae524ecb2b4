#!/usr/bin/env python3
"""Writes the node-block pattern of a P2 Kuhn mesh to a binary file for the kernel experiments."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from polyfem_amd import engine as E, mesh as M
n = int(sys.argv[1]); out = sys.argv[2]
numbering = sys.argv[3] if len(sys.argv) > 3 else "lattice"
mesh = M.cube_tets(n, 2, numbering)
eng = E.Engine(0); eng.set_mesh(mesh); eng.set_material(E.LAPLACIAN, 1, 1); eng.build_pattern()
rp, ci = eng.get_pattern()
with open(out, "wb") as f:
    np.array([mesh.n_nodes, ci.size], dtype=np.int64).tofile(f)
    rp.astype(np.int64).tofile(f)
    ci.astype(np.int32).tofile(f)
print("dumped", mesh.n_nodes, ci.size)
