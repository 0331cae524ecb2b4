"""Times the element kernels (energy / gradient / Hessian state) of the nonlinear interface on the 55^3 P2 grid."""
import sys
import torch
sys.path.insert(0, ".")
from polyfem_amd import engine as E
from polyfem_amd import mesh as M

n = int(sys.argv[1]) if len(sys.argv) > 1 else 55
mesh = M.cube_tets(n, 2)
eng = E.Engine(0)
eng.set_mesh(mesh)
eng.set_material(E.NEOHOOKEAN, 2e4 * 0.3 / (1.3 * 0.4), 2e4 / 2.6)
eng.build_pattern()
u = torch.from_numpy(M.seeded_displacement(mesh, 3, 1e-2 / n)).cuda()
g = torch.zeros_like(u)
e = torch.zeros(1, dtype=torch.float64, device="cuda")
for rep in range(2):
    eng.timing_reset()
    eng.timing_enable(1)
    for _ in range(5):
        eng.energy_dev(u.data_ptr(), e.data_ptr())
        eng.gradient_dev(u.data_ptr(), g.data_ptr())
        eng.hessian_dev(u.data_ptr())
    eng.synchronize()
    eng.timing_enable(0)
print(" ".join("%s %.3f ms" % (f, eng.timing_get(f)[0]) for f in ("local_energy", "local_grad", "gather_vec", "local", "gather")), "energy %.15e" % e.item(), "gsum %.15e" % g.abs().sum().item())
eng.close()
