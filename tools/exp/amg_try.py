"""AMG-preconditioned CG on a clamped cube for several parameter sets: usage amg_try.py {tet2,hex2,hex1,tet1} n"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from polyfem_amd import engine as E
from polyfem_amd import mesh as M

kind, n = sys.argv[1], int(sys.argv[2])
mesh = {"tet1": lambda: M.cube_tets(n, 1), "tet2": lambda: M.cube_tets(n, 2), "hex1": lambda: M.cube_hexes(n, 1), "hex2": lambda: M.cube_hexes(n, 2)}[kind]()
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
x, it, err = eng.cg(b, precond=E.PRECOND_JACOBI, max_iter=3000, tol=1e-10)
print("jacobi its", it, "err %.2e" % err, flush=True)
sets = [dict(), dict(ncycle=1), dict(relax_degree=2), dict(relax_degree=3), dict(aggregate_passes=3), dict(aggregate_passes=3, relax_degree=2),
        dict(relax_degree=2, ncycle=1), dict(relax_lower=0.3), dict(relax_lower=0.03, relax_degree=3), dict(power_iters=30)]
for ps in sets:
    eng.amg_set_params(**ps)
    t = time.perf_counter(); eng.amg_setup(); ts = time.perf_counter() - t
    t = time.perf_counter(); x, it, err = eng.cg(b, precond=E.PRECOND_AMG, max_iter=500, tol=1e-10); tc = time.perf_counter() - t
    inf = eng.amg_info()
    print(ps, "its", it, "err %.2e" % err, "setup %.2fs solve %.2fs" % (ts, tc), "levels", inf["n_vertices"], "lmax", ["%.2f" % v for v in inf["lambda_max"]], flush=True)
eng.close()
