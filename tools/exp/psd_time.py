"""Times the project_to_psd Hessian assembly (element projection + row assembly) on the 55^3 P2 grid."""
import sys
import numpy as np
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
amp = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-2
u = torch.from_numpy(M.seeded_displacement(mesh, 3, amp / n)).cuda()
for psd in (False, True):
    eng.hessian_dev(u.data_ptr(), psd)
    eng.synchronize()
    eng.timing_reset()
    eng.timing_enable(1)
    for _ in range(3):
        eng.hessian_dev(u.data_ptr(), psd)
    eng.synchronize()
    eng.timing_enable(0)
    print("psd" if psd else "plain", "local %.3f ms" % eng.timing_get("local")[0], "rows %.3f ms" % eng.timing_get("gather_psd" if psd else "gather")[0], flush=True)
v = eng.get_values()
print("checksum %.17g" % float(np.abs(v).sum()))
eng.close()
