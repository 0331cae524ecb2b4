"""AMG-preconditioned CG as bench.py sets the problem up (body force, x = 0 face clamped), power-iteration counts: amg_try2.py kind n"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from polyfem_amd import engine as E
from polyfem_amd import mesh as M

kind, n = sys.argv[1], int(sys.argv[2])
mesh = {"tet2": lambda: M.cube_tets(n, 2), "hex2": lambda: M.cube_hexes(n, 2)}[kind]()
eng = E.Engine(0)
eng.set_mesh(mesh)
eng.set_material(E.LINEAR_ELASTICITY, 1e5 * 0.3 / (1.3 * 0.4), 1e5 / 2.6)
eng.build_pattern()
eng.assemble_linear()
bnd = M.boundary_dofs(mesh, 3, "x0")
eng.set_dirichlet(bnd)
N = mesh.n_nodes * 3
b = np.zeros(N); b[2::3] = -1.0 / mesh.n_nodes; b[bnd] = 0
print(kind, n, "dofs", N, flush=True)
for ps in [dict(), dict(power_iters=30), dict(power_iters=60), dict(power_iters=30, relax_degree=2)]:
    eng.amg_set_params(**ps)
    t = time.perf_counter(); eng.amg_setup(); ts = time.perf_counter() - t
    t = time.perf_counter(); x, it, err = eng.cg(b, precond=E.PRECOND_AMG, max_iter=300, tol=1e-10); tc = time.perf_counter() - t
    inf = eng.amg_info()
    print(ps, "its", it, "err %.2e" % err, "setup %.2fs solve %.2fs" % (ts, tc), "lmax", ["%.2f" % v for v in inf["lambda_max"]], flush=True)
eng.close()
