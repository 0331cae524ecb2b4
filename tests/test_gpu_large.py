"""GPU tests at BASELINE.json's full sizes, through size-independent properties (the oracle cannot run these sizes in
seconds): symmetry, null space, energy/stiffness consistency, linearity of the operator, NeoHookean at rest ==
linear elasticity, CG residual, run-to-run bitwise determinism."""
import numpy as np
import pytest
import scipy.sparse as sp

from polyfem_amd import engine as E
from polyfem_amd import mesh as M

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big():
    mesh = M.cube_tets(55, 2)  # 998 250 P2 tets, 4.1 M dofs (config 2 / 3)
    eng = E.Engine(0)
    eng.set_mesh(mesh)
    yield mesh, eng
    eng.close()


def test_config2_linear_elasticity_properties(big):
    mesh, eng = big
    lam, mu = 1e5 * 0.3 / (1.3 * 0.4), 1e5 / 2.6
    eng.set_material(E.LINEAR_ELASTICITY, lam, mu)
    nnz = eng.build_pattern()
    assert mesh.n_el == 998250 and nnz == 348165189
    eng.assemble_linear()
    n = mesh.n_nodes * 3
    rng = np.random.default_rng(0)
    x, y = rng.standard_normal(n), rng.standard_normal(n)
    Ax, Ay = eng.spmv(x), eng.spmv(y)
    # linearity and symmetry of the operator
    assert np.abs(eng.spmv(2.0 * x - 3.0 * y) - (2.0 * Ax - 3.0 * Ay)).max() <= 1e-11 * np.abs(Ax).max()
    assert abs(y @ Ax - x @ Ay) <= 1e-11 * abs(y @ Ax)
    # rigid translations and an infinitesimal rotation are in the null space
    scale = np.abs(Ax).max()
    for c in range(3):
        t = np.zeros(n)
        t[c::3] = 1.0
        assert np.abs(eng.spmv(t)).max() <= 1e-10 * scale
    rot = np.zeros(n)
    rot[0::3], rot[1::3] = -mesh.node_xyz[:, 1], mesh.node_xyz[:, 0]
    assert np.abs(eng.spmv(rot)).max() <= 1e-9 * scale
    # positive semi-definite: x^T K x = 2 * energy > 0
    assert x @ Ax > 0
    # the reference's value-dependent pattern drops exact zeros only
    rp, ci, v = eng.get_linear_csr()
    assert rp[-1] == v.size < nnz and np.all(np.diff(rp) > 0)
    K = sp.csr_matrix((v, ci, rp), shape=(n, n))
    assert np.abs(K @ x - Ax).max() <= 1e-11 * scale
    # CG on the reduced system: residual bound and determinism
    eng.set_dirichlet(M.boundary_dofs(mesh, 3, "x0"))
    b = rng.standard_normal(n)
    x1, it1, err1 = eng.cg(b, None, E.PRECOND_JACOBI, 200, 1e-12)
    x2, it2, err2 = eng.cg(b, None, E.PRECOND_JACOBI, 200, 1e-12)
    assert it1 == it2 == 200 and np.array_equal(x1, x2) and err1 == err2
    free = np.ones(n, dtype=bool)
    free[M.boundary_dofs(mesh, 3, "x0")] = False
    r = (b - eng.spmv(x1))[free]
    assert abs(np.linalg.norm(r) / np.linalg.norm(b[free]) - err1) <= 1e-6 * err1
    eng.set_dirichlet(np.zeros(0, dtype=np.int32))


def test_config3_neohookean_properties(big):
    mesh, eng = big
    lam, mu = 2e4 * 0.3 / (1.3 * 0.4), 2e4 / 2.6
    eng.set_material(E.NEOHOOKEAN, lam, mu)
    eng.build_pattern()
    n = mesh.n_nodes * 3
    zero = np.zeros(n)
    assert abs(eng.energy(zero)) < 1e-6 and np.abs(eng.gradient(zero)).max() < 1e-9
    h0 = eng.hessian(zero)
    # at rest the NeoHookean Hessian is the linear-elasticity stiffness with the same Lame parameters
    eng.set_material(E.LINEAR_ELASTICITY, lam, mu)
    eng.assemble_linear()
    k = eng.get_values()
    assert np.abs(h0 - k).max() <= 1e-11 * np.abs(k).max()
    eng.set_material(E.NEOHOOKEAN, lam, mu)
    u = M.seeded_displacement(mesh, 3, 1e-2 / 55)
    e = eng.energy(u)
    g = eng.gradient(u)
    assert e > 0
    d = g / np.abs(g).max()  # along the gradient: no cancellation in g.d over the 4 M dofs
    hh = 1e-6
    fd = (eng.energy(u + hh * d) - eng.energy(u - hh * d)) / (2 * hh)
    assert abs(fd - g @ d) <= 1e-5 * abs(g @ d)
    eng.hessian(u, fetch=False)
    Hd = eng.spmv(d)
    gfd = (eng.gradient(u + hh * d) - eng.gradient(u - hh * d)) / (2 * hh)
    assert np.linalg.norm(gfd - Hd) <= 1e-5 * np.linalg.norm(Hd)
    # pattern reuse: a second assembly of the same state is bitwise identical
    v1 = eng.hessian(u)
    v2 = eng.hessian(u)
    assert np.array_equal(v1, v2)
