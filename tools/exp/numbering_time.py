"""SpMV and row-assembly time for several node numberings of the same mesh: lattice, PolyFEM's, random, random + Morton."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from polyfem_amd import engine as E
from polyfem_amd import mesh as M
from polyfem_amd import distributed as D

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
base = M.cube_tets(n, 2)
rng = np.random.default_rng(0)
rand = D.renumber_nodes(base, rng.permutation(base.n_nodes).astype(np.int64))
cases = {"lattice": base, "polyfem": M.cube_tets(n, 2, numbering="polyfem"), "random": rand,
         "random+morton": D.renumber_nodes(rand, D.morton_numbering(rand.node_xyz))}
for name, mesh in cases.items():
    eng = E.Engine(0)
    eng.set_mesh(mesh)
    eng.set_material(E.NEOHOOKEAN, 2e4 * 0.3 / (1.3 * 0.4), 2e4 / 2.6)
    eng.build_pattern()
    u = torch.from_numpy(M.seeded_displacement(mesh, 3, 1e-2 / n)).cuda()
    y = torch.zeros_like(u)
    eng.hessian_dev(u.data_ptr())
    eng.spmv_dev(u.data_ptr(), y.data_ptr())
    eng.timing_reset(); eng.timing_enable(1)
    for _ in range(5):
        eng.hessian_dev(u.data_ptr())
        eng.spmv_dev(u.data_ptr(), y.data_ptr())
    eng.synchronize(); eng.timing_enable(0)
    print(f"{name:14s} elements {mesh.n_el}: rows {eng.timing_get('gather')[0]:.3f} ms  local {eng.timing_get('local')[0]:.3f} ms  copy {eng.timing_get('sell_convert')[0]:.3f} ms  spmv {eng.timing_get('spmv')[0]:.3f} ms", flush=True)
    eng.close()
