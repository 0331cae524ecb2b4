import sys, time, numpy as np
sys.path.insert(0, '.')
from polyfem_amd import engine as E, mesh as M
n = int(sys.argv[1])
t = time.time(); mesh = M.cube_hexes(n, 2); print("mesh", time.time() - t, mesh.n_el, mesh.n_nodes, flush=True)
eng = E.Engine(0); eng.set_mesh(mesh)
lam, mu = 1e5 * 0.3 / (1.3 * 0.4), 1e5 / 2.6
eng.set_material(E.LINEAR_ELASTICITY, lam, mu)
t = time.time(); nnz = eng.build_pattern(); eng.synchronize(); print("pattern", time.time() - t, nnz, flush=True)
eng.timing_enable(1)
t = time.time(); eng.assemble_linear(); eng.synchronize(); print("assemble", time.time() - t, flush=True)
N = mesh.n_nodes * 3
x = np.random.default_rng(0).standard_normal(N)
t = time.time(); y = eng.spmv(x); print("spmv(host)", time.time() - t, flush=True)
for f in ("local", "gather", "sell_convert", "spmv"):
    print(f, eng.timing_get(f))
print(eng.info().bytes_device / 1e9, "GB")
