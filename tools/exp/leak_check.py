"""Creates and destroys engines through the whole call sequence and reports the free device memory before / after."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from polyfem_amd import engine as E, mesh as M

torch.cuda.init()
mesh = M.cube_tets(12, 2)
n = mesh.n_nodes * 3
bnd = M.boundary_dofs(mesh, 3, "x0")
rhs = np.zeros(n); rhs[2::3] = -1.0 / mesh.n_nodes
free0 = None
for rep in range(12):
    eng = E.Engine(0)
    eng.set_mesh(mesh)
    eng.set_material(E.LINEAR_ELASTICITY, 5e4, 4e4)
    eng.build_pattern(); eng.assemble_linear(); eng.get_linear_csr()
    eng.set_elasticity_tensor(E.elasticity_tensor(lam=5e4, mu=4e4)); eng.assemble_linear()
    eng.set_material(E.NEOHOOKEAN, 1e4, 8e3)
    eng.set_element_models([E.NEOHOOKEAN, E.SAINT_VENANT] * (mesh.n_el // 2))
    eng.set_dirichlet(bnd)
    eng.newton_solve(np.zeros(n), rhs)
    eng.hessian(np.zeros(n), project_to_psd=True)
    eng.inertia_set_mass(1000.0)
    eng.implicit_euler_step(1e-3, np.zeros(n), np.zeros(n), rhs)
    eng.al_solve(np.zeros(n), rhs, bnd, np.zeros(bnd.size), force_al=True)
    eng.set_mesh(M.cube_hexes(4, 2)); eng.set_material(E.LAPLACIAN, 1, 1); eng.build_pattern(); eng.assemble_linear()
    eng.close()
    free, total = torch.cuda.mem_get_info()
    if rep == 1:
        free0 = free
    print(rep, "free MiB", free >> 20)
print("drift after warm-up (MiB):", (free0 - free) / 2**20)
