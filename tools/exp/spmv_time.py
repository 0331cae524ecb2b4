"""Times the SELL SpMV (with the fused dot) on the n^3 P2 grid: usage spmv_time.py n [reps]"""
import sys
import torch
sys.path.insert(0, ".")
from polyfem_amd import engine as E
from polyfem_amd import mesh as M

n = int(sys.argv[1]) if len(sys.argv) > 1 else 55
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
mesh = M.cube_tets(n, 2, numbering=(sys.argv[3] if len(sys.argv) > 3 else "lattice"))
eng = E.Engine(0)
eng.set_mesh(mesh)
eng.set_material(E.LINEAR_ELASTICITY, 1e5 * 0.3 / (1.3 * 0.4), 1e5 / 2.6)
eng.build_pattern()
eng.assemble_linear()
x = torch.from_numpy(M.seeded_displacement(mesh, 3, 1.0)).cuda()
y = torch.zeros_like(x)
for _ in range(3):
    eng.spmv_dev(x.data_ptr(), y.data_ptr())
eng.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
eng.timing_reset(); eng.timing_enable(1)
for _ in range(reps):
    eng.spmv_dev(x.data_ptr(), y.data_ptr())
eng.synchronize(); eng.timing_enable(0)
print("spmv %.4f ms  checksum %.17e" % (eng.timing_get("spmv")[0], y.double().abs().sum().item()))
eng.close()
