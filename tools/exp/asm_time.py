"""Times the row-assembly kernels (linear stiffness and NeoHookean Hessian) on the n^3 P2 grid; a few launches, for PMC passes."""
import sys
import torch
sys.path.insert(0, ".")
from polyfem_amd import engine as E
from polyfem_amd import mesh as M

n = int(sys.argv[1]) if len(sys.argv) > 1 else 55
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
mesh = M.cube_tets(n, 2, numbering=(sys.argv[3] if len(sys.argv) > 3 else "lattice"))
eng = E.Engine(0)
eng.set_mesh(mesh)
eng.set_material(E.LINEAR_ELASTICITY, 1e5 * 0.3 / (1.3 * 0.4), 1e5 / 2.6)
eng.build_pattern()
u = torch.from_numpy(M.seeded_displacement(mesh, 3, 1e-2 / n)).cuda()
eng.assemble_linear()
eng.timing_reset(); eng.timing_enable(1)
for _ in range(reps):
    eng.assemble_linear()
eng.synchronize(); eng.timing_enable(0)
print("linear: local %.3f ms rows %.3f ms" % (eng.timing_get("local")[0], eng.timing_get("gather")[0]))
eng.set_material(E.NEOHOOKEAN, 2e4 * 0.3 / (1.3 * 0.4), 2e4 / 2.6)
eng.hessian_dev(u.data_ptr())
eng.timing_reset(); eng.timing_enable(1)
for _ in range(reps):
    eng.hessian_dev(u.data_ptr())
eng.synchronize(); eng.timing_enable(0)
print("neohookean: local %.3f ms rows %.3f ms" % (eng.timing_get("local")[0], eng.timing_get("gather")[0]))
y = torch.zeros_like(u)
eng.spmv_dev(u.data_ptr(), y.data_ptr()); eng.synchronize()
print("checksum %.17e" % y.double().abs().sum().item())
eng.close()
