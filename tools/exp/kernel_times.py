"""Times the element kernels and the row assembly for one element kind: usage kernel_times.py {tet1,tet2,hex1,hex2} n [reps]"""
import sys
import torch
sys.path.insert(0, ".")
from polyfem_amd import engine as E
from polyfem_amd import mesh as M

kind, n = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
mesh = {"tet1": lambda: M.cube_tets(n, 1), "tet2": lambda: M.cube_tets(n, 2), "hex1": lambda: M.cube_hexes(n, 1), "hex2": lambda: M.cube_hexes(n, 2)}[kind]()
eng = E.Engine(0)
eng.set_mesh(mesh)
eng.set_material(E.LINEAR_ELASTICITY, 1e5 * 0.3 / (1.3 * 0.4), 1e5 / 2.6)
eng.build_pattern()
print(kind, n, "elements", mesh.n_el, "nodes", mesh.n_nodes, "nnz", eng.info().nnz, flush=True)
eng.assemble_linear()
eng.timing_reset(); eng.timing_enable(1)
for _ in range(reps):
    eng.assemble_linear()
eng.synchronize(); eng.timing_enable(0)
print("linear stiffness: local %.3f ms rows %.3f ms" % (eng.timing_get("local")[0], eng.timing_get("gather")[0]), flush=True)
eng.set_material(E.NEOHOOKEAN, 2e4 * 0.3 / (1.3 * 0.4), 2e4 / 2.6)
u = torch.from_numpy(M.seeded_displacement(mesh, 3, 1e-2 / n)).cuda()
g = torch.zeros_like(u)
e = torch.zeros(1, dtype=torch.float64, device="cuda")
eng.energy_dev(u.data_ptr(), e.data_ptr()); eng.gradient_dev(u.data_ptr(), g.data_ptr()); eng.hessian_dev(u.data_ptr())
eng.timing_reset(); eng.timing_enable(1)
for _ in range(reps):
    eng.energy_dev(u.data_ptr(), e.data_ptr())
    eng.gradient_dev(u.data_ptr(), g.data_ptr())
    eng.hessian_dev(u.data_ptr())
eng.synchronize(); eng.timing_enable(0)
print("neohookean: " + " ".join("%s %.3f ms" % (f, eng.timing_get(f)[0]) for f in ("local_energy", "local_grad", "gather_vec", "local", "gather")), flush=True)
print("energy %.15e gsum %.15e" % (e.item(), g.abs().sum().item()))
eng.close()
