import sys, time
import numpy as np, torch
sys.path.insert(0, ".")  # run from the repository root
from polyfem_amd import engine as E, mesh as M
from oracle import pf_oracle as O
mesh = M.cube_hexes(40, 2)
eng = E.Engine(0); eng.set_mesh(mesh); eng.set_material(E.NEOHOOKEAN, 1e4, 8e3); eng.build_pattern()
u = torch.from_numpy(M.seeded_displacement(mesh, 3, 1e-3)).cuda(); e = torch.zeros(1, dtype=torch.float64, device="cuda")
eng.energy_dev(u.data_ptr(), e.data_ptr()); eng.synchronize()
eng.timing_reset(); eng.timing_enable(1)
for _ in range(5): eng.energy_dev(u.data_ptr(), e.data_ptr())
eng.synchronize(); eng.timing_enable(0)
print("q2 64000 elements: energy kernel %.3f ms, energy %.15e" % (eng.timing_get("local_energy")[0], e.item()))
small = M.cube_hexes(3, 2); eng.set_mesh(small); eng.set_material(E.NEOHOOKEAN, 1e4, 8e3); eng.build_pattern()
x = M.seeded_displacement(small, 3, 1e-2)
print("parity", abs(eng.energy(x) - O.assemble_energy(small, O.NEOHOOKEAN, 1e4, 8e3, x)) / abs(eng.energy(x)))
