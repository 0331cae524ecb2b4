"""Times pfx_solver_analyze / pfx_solver_factorize(_dev) / CG on a matrix handed in by value: the elasticity stiffness
matrix of a P2 tetrahedral cube, fetched from an assembly context and given to a second, mesh-less solver context.
usage: python tools/exp/extmat_time.py [grid]"""
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
eng.assemble_linear()
n = mesh.n_nodes * 3
row_ptr, col_idx = eng.get_pattern()   # symmetric structure: the CSR arrays are the CSC arrays
vals = eng.get_values()
b = torch.from_numpy(np.random.default_rng(1).standard_normal(n)).cuda()
x0 = torch.zeros(n, dtype=torch.float64, device="cuda")
t0 = time.time(); it0, err0 = eng.cg_dev(b.data_ptr(), x0.data_ptr(), E.PRECOND_JACOBI, 200, 1e-12); eng.synchronize(); t_own = time.time() - t0
eng.close()

# no constraints: the matrix is singular but its diagonal is positive, and 200 iterations on the same right-hand side compare the two contexts
own = E.Engine(0)
t0 = time.time(); own.solver_analyze(n, row_ptr, col_idx); own.synchronize(); t_an = time.time() - t0
bs, nb, stored = own.solver_layout()
t0 = time.time(); own.solver_factorize(vals); own.synchronize(); t_fh = time.time() - t0
dv = torch.from_numpy(vals).cuda()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.time(); own.solver_factorize_dev(dv.data_ptr()); own.synchronize(); ts.append(time.time() - t0)
x1 = torch.zeros(n, dtype=torch.float64, device="cuda")
t0 = time.time(); it1, err1 = own.cg_dev(b.data_ptr(), x1.data_ptr(), E.PRECOND_JACOBI, 200, 1e-12); own.synchronize(); t_ext = time.time() - t0
print(f"grid {grid} elements {mesh.n_el} n {n} entries {col_idx.size} layout bs {bs} blocks {nb} stored {stored}")
print(f"analyze {t_an*1e3:.1f} ms (host arrays, {(row_ptr.nbytes + col_idx.nbytes)/1e9:.2f} GB over PCIe)  factorize from host values {t_fh*1e3:.1f} ms"
      f"  factorize from device values {min(ts)*1e3:.2f} ms")
print(f"200 Jacobi-CG iterations: assembly context {t_own*1e3:.1f} ms ({it0} its, err {err0:.3e}); handed-in matrix {t_ext*1e3:.1f} ms ({it1} its, err {err1:.3e})"
      f"  max |x_own - x_ext| {float((x0 - x1).abs().max()):.3e}")
own.close()
