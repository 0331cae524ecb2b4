import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from polyfem_amd import engine as E, distributed as D
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 55
part, _, _ = D.slab_part(grid, grid, 0, 1, 2, "tet")
mesh = part.mesh
eng = E.Engine(0)
eng.set_mesh(mesh)
lam, mu = 1e5 * 0.3 / (1.3 * 0.4), 1e5 / 2.6
eng.set_material(E.LINEAR_ELASTICITY, lam, mu)
eng.build_pattern()
fixed_nodes = np.nonzero(np.abs(mesh.node_xyz[:, 0]) < 1e-12)[0]
eng.set_dirichlet(np.sort((fixed_nodes[:, None] * 3 + np.arange(3)[None]).reshape(-1)).astype(np.int32))
eng.assemble_linear()
n = mesh.n_nodes * 3
b = torch.from_numpy(np.random.default_rng(1).standard_normal(n)).cuda()
x = torch.zeros(n, dtype=torch.float64, device="cuda")
import os
kw = {}
if os.environ.get('AMG_NCYCLE'): kw['ncycle'] = int(os.environ['AMG_NCYCLE'])
if os.environ.get('AMG_DEGREE'): kw['relax_degree'] = int(os.environ['AMG_DEGREE'])
if os.environ.get('AMG_PASSES'): kw['aggregate_passes'] = int(os.environ['AMG_PASSES'])
if os.environ.get('AMG_OMEGA'): kw['coarse_scale'] = float(os.environ['AMG_OMEGA'])
if kw: eng.amg_set_params(**kw)
t0 = time.time(); eng.amg_setup(); eng.synchronize(); t1 = time.time()
info = eng.amg_info()
print("setup wall", t1 - t0, "symbolic", info["symbolic_s"], "numeric", info["numeric_s"], "levels", info["n_levels"], "cx", info["operator_complexity"])
print("verts", info["n_vertices"][:info["n_levels"]], "aggs", info["n_aggregates"][:info["n_levels"]])
t0 = time.time(); it, err = eng.cg_dev(b.data_ptr(), x.data_ptr(), E.PRECOND_AMG, 1000, 1e-10); eng.synchronize(); t1 = time.time()
print("grid", grid, "elements", mesh.n_el, "AMG iters", it, "err", err, "solve s", t1 - t0)
a0 = eng.amg_aggregates(0)
sz = np.bincount(a0[a0 >= 0])
print("agg sizes min/mean/max", sz.min(), sz.mean(), sz.max(), "hist", np.bincount(sz)[:24])
eng.close()
