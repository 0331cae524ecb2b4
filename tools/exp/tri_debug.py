import sys; sys.path.insert(0, '.')
import numpy as np, torch
from polyfem_amd import engine as E, mesh as M
from oracle import pf_oracle as O
eng = E.Engine(0)
m = M.square_tris(2, 1)
eng.set_mesh(m); eng.set_material(E.LAPLACIAN, 1.0, 1.0); eng.build_pattern(); eng.assemble_linear()
rp, ci, v = eng.get_linear_csr()
o = O.assemble_linear(m, O.LAPLACIAN, 1.0, 1.0)
print("gpu", rp, ci, v)
print("ora", o[0], o[1], o[2])
prp, pci = eng.get_pattern(); print("pattern", prp, pci); print("values", eng.get_values())
