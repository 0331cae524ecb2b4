import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from polyfem_amd import engine as E, mesh as M
n = int(sys.argv[1]) if len(sys.argv) > 1 else 55
mesh = M.cube_tets(n, 2)
eng = E.Engine(0); eng.set_mesh(mesh)
eng.set_material(E.NEOHOOKEAN, 2e4 * 0.3 / (1.3 * 0.4), 2e4 / 2.6); eng.build_pattern()
nd = mesh.n_nodes * 3
x = np.zeros(nd); x[0::3] = -0.2 * mesh.node_xyz[:, 0]; x[1::3] = 0.15 * mesh.node_xyz[:, 1]
u = torch.from_numpy(x).cuda()
eng.timing_enable(1)
for psd in (False, True, True):
    eng.synchronize(); t = time.time(); eng.hessian_dev(u.data_ptr(), psd); eng.synchronize(); print("psd", psd, "%.2f ms" % ((time.time() - t) * 1e3))
for f in ("local", "gather", "gather_psd"):
    print(f, eng.timing_get(f))
