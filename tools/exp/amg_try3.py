"""AMG robustness probes: jittered P2 mesh, stretched mesh, material contrast, NeoHookean Hessian at a deformed state."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from polyfem_amd import engine as E
from polyfem_amd import mesh as M

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cases = {
    "aligned": lambda: M.cube_tets(n, 2),
    "jitter0.12": lambda: M.cube_tets(n, 2, jitter=0.12, seed=1),
    "stretched 1:4:0.25": lambda: M.cube_tets(n, 2, scale=(1.0, 4.0, 0.25)),
    "polyfem numbering": lambda: M.cube_tets(n, 2, numbering="polyfem"),
    "q2": lambda: M.cube_hexes(n // 2, 2),
    "p1": lambda: M.cube_tets(n, 1),
}
lam, mu = 1e5 * 0.3 / (1.3 * 0.4), 1e5 / 2.6
for name, mk in cases.items():
    mesh = mk()
    eng = E.Engine(0)
    eng.set_mesh(mesh)
    N = mesh.n_nodes * 3
    bnd = M.boundary_dofs(mesh, 3, "x0")
    b = np.zeros(N); b[2::3] = -1.0 / mesh.n_nodes; b[bnd] = 0
    for mat in ("uniform", "contrast1e3"):
        if mat == "uniform":
            eng.set_material(E.LINEAR_ELASTICITY, lam, mu)
        elif mat == "contrast1e3":
            f = np.where(np.arange(mesh.n_el) % 7 < 3, 1e3, 1.0)
            eng.set_material(E.LINEAR_ELASTICITY, lam * f, mu * f)
        else:
            eng.set_material(E.NEOHOOKEAN, lam, mu)
        eng.build_pattern()
        if mat == "neohookean":
            eng.hessian(M.seeded_displacement(mesh, 3, 2e-2 / n), fetch=False)
        else:
            eng.assemble_linear()
        eng.set_dirichlet(bnd)
        xj, itj, errj = eng.cg(b, precond=E.PRECOND_JACOBI, max_iter=2000, tol=1e-10)
        t = time.perf_counter()
        x, it, err = eng.cg(b, precond=E.PRECOND_AMG, max_iter=500, tol=1e-10)
        print(f"{name:20s} {mat:12s} dofs {N}: jacobi {itj} ({errj:.1e})  amg {it} ({err:.1e}) {time.perf_counter() - t:.2f}s  lmax {eng.amg_info()['lambda_max'][0]:.2f}", flush=True)
        eng.set_dirichlet(np.zeros(0, dtype=np.int32))
    eng.close()
