"""GPU parity tests of the 2-D element kinds (P1 / P2 triangles) through the C ABI, against the CPU oracle.

The reference instantiates its NeoHookean kernels for (3,2) and (6,2) (`NeoHookeanElasticity.cpp:47-115`) and its two
assembler property tests run on 2-D triangle meshes: `hessian_lin` (`tests/test_assembler.cpp:12-64`) and the
finite-difference test of the elastic form (`tests/test_form_derivatives.cpp:84-143, 200-227`).  Their mesh files
(`plane_hole.obj`, `circle36.obj`) are not part of the reference tree, so both run here on seeded triangle meshes:
parity unpinned beyond the oracle, the properties themselves are oracle-independent.

Bars as for the 3-D kinds: pattern bit-exact, values / energies / gradients 1e-12, CG +-2 iterations and 1e-10."""
import numpy as np
import pytest

from polyfem_amd import engine as E
from polyfem_amd import mesh as M
from tests.helpers import csc_to_scipy, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-12

MESHES = {
    "p1_aligned": lambda: M.square_tris(6, 1),
    "p1_jitter": lambda: M.square_tris(7, 1, jitter=0.25, seed=1),
    "p2_aligned": lambda: M.square_tris(5, 2),
    "p2_jitter_polyfem": lambda: M.square_tris(6, 2, numbering="polyfem", jitter=0.2, seed=2),
    "p2_stretched": lambda: M.square_tris(4, 2, scale=(2.0, 0.5)),
    "p2_one_cell": lambda: M.square_tris(1, 2),
}


@pytest.fixture(scope="module")
def eng():
    e = E.Engine(0)
    yield e
    e.close()


def _setup(eng, mesh, model, lam, mu):
    eng.set_mesh(mesh)
    eng.set_material(model, lam, mu)
    return eng.build_pattern()


@pytest.mark.parametrize("name", list(MESHES))
@pytest.mark.parametrize("model", ["Laplacian", "LinearElasticity"])
def test_linear_assembly_pattern_bit_exact_and_values(eng, oracle, name, model):
    mesh = MESHES[name]()
    mid = E.MODEL_BY_NAME[model]
    lam, mu = oracle.lame(1e5, 0.3, 2)
    _setup(eng, mesh, mid, lam, mu)
    eng.assemble_linear()
    rp, ci, v = eng.get_linear_csr()
    outer, inner, vals = oracle.assemble_linear(mesh, mid, lam, mu)
    assert rp.size == mesh.n_nodes * (1 if mid == E.LAPLACIAN else 2) + 1
    assert np.array_equal(rp, outer.astype(np.int64)), "row_ptr differs from the oracle"
    assert np.array_equal(ci, inner), "col_idx differs from the oracle"
    assert rel_err(v, vals) <= TOL


@pytest.mark.parametrize("name", list(MESHES))
@pytest.mark.parametrize("model", ["NeoHookean", "LinearElasticity"])
def test_energy_gradient_hessian(eng, oracle, name, model):
    mesh = MESHES[name]()
    mid = E.MODEL_BY_NAME[model]
    lam, mu = oracle.lame(2e4, 0.3, 2)
    nnz = _setup(eng, mesh, mid, lam, mu)
    h = 1.0 / round((mesh.n_el / 2) ** 0.5)
    for amp in (0.0, 2e-2 * h):
        x = M.seeded_displacement(mesh, 2, amp)
        e_ref = oracle.assemble_energy(mesh, mid, lam, mu, x)
        g_ref = oracle.assemble_gradient(mesh, mid, lam, mu, x)
        outer, inner, vals = oracle.assemble_hessian(mesh, mid, lam, mu, x)
        e, g, v = eng.energy(x), eng.gradient(x), eng.hessian(x)
        rp, ci = eng.get_pattern()
        assert np.array_equal(rp, outer.astype(np.int64)) and np.array_equal(ci, inner), "Hessian pattern not bit-exact"
        assert nnz == inner.size and g.size == 2 * mesh.n_nodes
        if amp == 0.0:
            assert abs(e) <= 1e-9 and np.abs(g).max() <= 1e-9 * np.abs(vals).max()
        else:
            assert abs(e - e_ref) <= TOL * abs(e_ref)
            assert rel_err(g, g_ref) <= TOL
        assert rel_err(v, vals) <= TOL


@pytest.mark.parametrize("order", [1, 2])
def test_hessian_lin_property(eng, oracle, order):
    """tests/test_assembler.cpp:12-64: the Hessian of the LinearElasticity energy, assembled through the nonlinear
    interface with the pattern reused, equals the stiffness matrix of the linear assembler to 1e-8 at zero and at nine
    random displacements."""
    mesh = M.square_tris(8, order, jitter=0.2, seed=3)
    lam, mu = oracle.lame(1e5, 0.3, 2)
    _setup(eng, mesh, E.LINEAR_ELASTICITY, lam, mu)
    eng.assemble_linear()
    n = 2 * mesh.n_nodes
    rp, ci, v = eng.get_linear_csr()
    import scipy.sparse as sp
    K = sp.csr_matrix((v, ci, rp), shape=(n, n))
    rng = np.random.default_rng(0)
    prp, pci = eng.get_pattern()
    for rand in range(10):
        x = np.zeros(n) if rand == 0 else rng.uniform(-1, 1, n)
        H = sp.csr_matrix((eng.hessian(x), pci, prp), shape=(n, n))
        assert abs(H - K).max() <= 1e-8 * max(1.0, abs(K).max())


@pytest.mark.parametrize("order", [1, 2])
def test_elastic_form_derivatives(eng, oracle, order):
    """tests/test_form_derivatives.cpp:84-143, 200-227: finite differences of the NeoHookean energy against its gradient,
    and of the gradient against the Hessian, at x = 0 and nine random x / 100."""
    mesh = M.square_tris(6, order, jitter=0.2, seed=4)
    lam, mu = oracle.lame(2e4, 0.3, 2)
    _setup(eng, mesh, E.NEOHOOKEAN, lam, mu)
    n = 2 * mesh.n_nodes
    rng = np.random.default_rng(1)
    prp, pci = eng.get_pattern()
    import scipy.sparse as sp
    step = 1e-6
    for rand in range(10):
        x = np.zeros(n) if rand == 0 else rng.uniform(-1, 1, n) / 100
        g = eng.gradient(x)
        H = sp.csr_matrix((eng.hessian(x), pci, prp), shape=(n, n))
        d = rng.standard_normal(n)
        d /= np.linalg.norm(d)
        fd_g = (eng.energy(x + step * d) - eng.energy(x - step * d)) / (2 * step)
        assert abs(fd_g - g @ d) <= 1e-6 * max(1.0, np.linalg.norm(g))
        fd_h = (eng.gradient(x + step * d) - eng.gradient(x - step * d)) / (2 * step)
        assert np.linalg.norm(fd_h - H @ d) <= 1e-6 * np.linalg.norm(H @ d)


@pytest.mark.parametrize("case", ["laplace_p1", "elastic_p2", "neohookean_p2"])
def test_reduced_matrix_and_cg(eng, oracle, case):
    if case == "laplace_p1":
        mesh, model, bs = M.square_tris(12, 1), E.LAPLACIAN, 1
    elif case == "elastic_p2":
        mesh, model, bs = M.square_tris(8, 2), E.LINEAR_ELASTICITY, 2
    else:
        mesh, model, bs = M.square_tris(8, 2, jitter=0.1), E.NEOHOOKEAN, 2
    lam, mu = oracle.lame(1e5, 0.3, 2)
    _setup(eng, mesh, model, lam, mu)
    n = mesh.n_nodes * bs
    if model == E.NEOHOOKEAN:
        x = M.seeded_displacement(mesh, 2, 1e-3)
        eng.hessian(x, fetch=False)
        full = oracle.assemble_hessian(mesh, oracle.NEOHOOKEAN, lam, mu, x)
    else:
        eng.assemble_linear()
        full = oracle.assemble_linear(mesh, model, lam, mu)
    bd = M.boundary_dofs(mesh, bs, "x0" if bs == 2 else "all")
    eng.set_dirichlet(bd)
    red = oracle.full_to_reduced_matrix(full, bd)
    rp, ci, v = eng.get_reduced_csr()
    assert np.array_equal(rp, red[0].astype(np.int64)) and np.array_equal(ci, red[1])
    assert rel_err(v, red[2]) <= TOL
    keep = np.setdiff1d(np.arange(n), bd)
    b_full = np.random.default_rng(5).standard_normal(n)
    xr, it_ref, _ = oracle.cg(red, b_full[keep], jacobi=True, max_iter=2000, tol=1e-12)
    xg, it, _ = eng.cg(b_full, None, E.PRECOND_JACOBI, 2000, 1e-12)
    assert abs(it - it_ref) <= 2, (it, it_ref)
    assert not xg[bd].any()
    A = csc_to_scipy(red, keep.size)
    assert np.linalg.norm(A @ xg[keep] - b_full[keep]) <= 1e-10 * np.linalg.norm(b_full[keep])
    assert np.linalg.norm(xg[keep] - xr) <= 1e-8 * np.linalg.norm(xr)
    # the product itself
    xv = np.random.default_rng(6).standard_normal(n)
    eng.set_dirichlet(np.zeros(0, dtype=np.int32))
    assert rel_err(eng.spmv(xv), csc_to_scipy(full, n) @ xv) <= 1e-12


def test_large_mesh_takes_the_slice_ordered_product(eng, oracle):
    """More than 65536 nodes: the CG keeps its products in slice order (the path the small meshes above never take)."""
    mesh = M.square_tris(140, 2)  # 281^2 = 78 961 nodes
    lam, mu = oracle.lame(1e5, 0.3, 2)
    _setup(eng, mesh, E.NEOHOOKEAN, lam, mu)
    n = 2 * mesh.n_nodes
    x = M.seeded_displacement(mesh, 2, 1e-5)
    assert abs(eng.energy(x) - oracle.assemble_energy(mesh, oracle.NEOHOOKEAN, lam, mu, x, n_threads=8)) <= TOL * abs(eng.energy(x))
    assert rel_err(eng.gradient(x), oracle.assemble_gradient(mesh, oracle.NEOHOOKEAN, lam, mu, x, n_threads=8)) <= TOL
    v = eng.hessian(x)
    full = oracle.assemble_hessian(mesh, oracle.NEOHOOKEAN, lam, mu, x, n_threads=8)
    assert rel_err(v, full[2]) <= TOL
    bd = M.boundary_dofs(mesh, 2, "x0")
    eng.set_dirichlet(bd)
    red = oracle.full_to_reduced_matrix(full, bd)
    keep = np.setdiff1d(np.arange(n), bd)
    b = np.random.default_rng(7).standard_normal(n)
    xr, it_ref, _ = oracle.cg(red, b[keep], jacobi=True, max_iter=300, tol=1e-12)
    xg, it, _ = eng.cg(b, None, E.PRECOND_JACOBI, 300, 1e-12)
    eng.set_dirichlet(np.zeros(0, dtype=np.int32))
    assert it == it_ref == 300  # the iteration limit: the iterates after the same number of steps
    assert np.linalg.norm(xg[keep] - xr) <= 1e-8 * np.linalg.norm(xr)


@pytest.mark.parametrize("case", ["neohookean_p2", "neohookean_p1_large_load", "linear_elasticity_p2"])
def test_newton_solve_matches_oracle(eng, oracle, case):
    if case == "neohookean_p2":
        mesh, model, load = M.square_tris(6, 2, jitter=0.1), E.NEOHOOKEAN, -50.0
    elif case == "neohookean_p1_large_load":
        mesh, model, load = M.square_tris(8, 1, jitter=0.1), E.NEOHOOKEAN, -1500.0
    else:
        mesh, model, load = M.square_tris(4, 2), E.LINEAR_ELASTICITY, -50.0
    lam, mu = oracle.lame(2e4, 0.3, 2)
    _setup(eng, mesh, model, lam, mu)
    n = mesh.n_nodes * 2
    bnd = M.boundary_dofs(mesh, 2, "x0")
    eng.set_dirichlet(bnd)
    rhs = np.zeros(n)
    rhs[1::2] = load / mesh.n_nodes
    x_ref, info_ref = oracle.newton_solve(mesh, model, lam, mu, rhs, bnd, np.zeros(n), n_threads=4)
    x, info = eng.newton_solve(np.zeros(n), rhs)
    eng.set_dirichlet(np.zeros(0, dtype=np.int32))
    assert info["status"] == info_ref["status"] and info["status"] in (2, 3)
    assert info["iterations"] == info_ref["iterations"] >= 2
    assert abs(info["line_search_iterations"] - info_ref["line_search_iterations"]) <= 1
    assert abs(info["cg_iterations"] - info_ref["cg_iterations"]) <= max(2 * info["iterations"], 0.03 * info_ref["cg_iterations"])
    assert np.abs(x - x_ref).max() <= 1e-9 * np.abs(x_ref).max()
    assert abs(info["energy"] - info_ref["energy"]) <= 1e-9 * abs(info_ref["energy"])
    assert not x[bnd].any()


@pytest.mark.parametrize("name", ["p1_jitter", "p2_jitter_polyfem"])
@pytest.mark.parametrize("size", [1, 2])
def test_mass_matrix_and_body_force(eng, oracle, name, size):
    mesh = MESHES[name]()
    lam, mu = oracle.lame(1e5, 0.3, 2)
    _setup(eng, mesh, E.LAPLACIAN if size == 1 else E.LINEAR_ELASTICITY, lam, mu)
    rng = np.random.default_rng(8)
    for rho in (1.7, rng.uniform(0.5, 2.0, mesh.n_el)):
        eng.assemble_mass(rho)
        rp, ci, v = eng.get_linear_csr()
        ref = oracle.assemble_mass(mesh, size, rho)
        assert np.array_equal(rp, ref[0].astype(np.int64)) and np.array_equal(ci, ref[1])
        assert rel_err(v, ref[2]) <= TOL
        force = np.array([0.3, -9.81])[:size]
        assert rel_err(eng.assemble_rhs(force, rho), oracle.assemble_rhs(mesh, size, force, rho)) <= TOL
    pts = eng.quadrature_points()
    assert pts.shape[2] == 2 and rel_err(pts, oracle.quadrature_points(mesh)) <= TOL
    f_pts = np.stack([np.sin(pts[..., 0]), np.cos(pts[..., 1])], axis=-1)[..., :size].copy()
    assert rel_err(eng.assemble_rhs_pointwise(f_pts, 1.3), oracle.assemble_rhs_pointwise(mesh, size, f_pts, 1.3)) <= TOL
    assert rel_err(eng.mass_quadrature_points(), oracle.mass_quadrature_points(mesh)) <= TOL


def test_per_element_and_per_point_materials(eng, oracle):
    mesh = MESHES["p2_jitter_polyfem"]()
    rng = np.random.default_rng(9)
    lam0, mu0 = oracle.lame(1e5, 0.3, 2)
    for shape in ((mesh.n_el,), (mesh.n_el, 4)):
        lam, mu = lam0 * rng.uniform(0.5, 2, shape), mu0 * rng.uniform(0.5, 2, shape)
        _setup(eng, mesh, E.LINEAR_ELASTICITY, lam, mu)
        eng.assemble_linear()
        rp, ci, v = eng.get_linear_csr()
        ref = oracle.assemble_linear(mesh, oracle.LINEAR_ELASTICITY, lam, mu)
        assert np.array_equal(ci, ref[1]) and rel_err(v, ref[2]) <= TOL
        _setup(eng, mesh, E.NEOHOOKEAN, lam, mu)
        x = M.seeded_displacement(mesh, 2, 2e-3)
        assert abs(eng.energy(x) - oracle.assemble_energy(mesh, oracle.NEOHOOKEAN, lam, mu, x)) <= TOL * abs(eng.energy(x))
        assert rel_err(eng.gradient(x), oracle.assemble_gradient(mesh, oracle.NEOHOOKEAN, lam, mu, x)) <= TOL
        assert rel_err(eng.hessian(x), oracle.assemble_hessian(mesh, oracle.NEOHOOKEAN, lam, mu, x)[2]) <= TOL


@pytest.mark.parametrize("name", ["p1_jitter", "p2_jitter_polyfem", "p2_stretched"])
def test_hessian_project_to_psd(eng, oracle, name):
    """assemble_hessian(project_to_psd = true) on triangles: (n_loc * 2)^2 element Hessians projected onto the PSD cone
    (Assembler.cpp:554-555); tolerance 1e-10 of the largest entry (different eigen-solvers, same projection)."""
    mesh = MESHES[name]()
    lam, mu = oracle.lame(2e4, 0.3, 2)
    nnz = _setup(eng, mesh, E.NEOHOOKEAN, lam, mu)
    n = mesh.n_nodes * 2
    x = np.zeros(n)
    x[0::2] = -0.35 * mesh.node_xyz[:, 0]
    x[1::2] = 0.3 * mesh.node_xyz[:, 1]
    x += M.seeded_displacement(mesh, 2, 2e-3)
    v = eng.hessian(x)
    vp = eng.hessian(x, project_to_psd=True)
    ref = oracle.assemble_hessian_psd(mesh, oracle.NEOHOOKEAN, lam, mu, x)[2]
    plain = oracle.assemble_hessian(mesh, oracle.NEOHOOKEAN, lam, mu, x)[2]
    assert vp.size == nnz == ref.size
    assert np.abs(plain - ref).max() > 1e-3 * np.abs(plain).max(), "test state is not indefinite anywhere"
    assert rel_err(v, plain) <= TOL
    assert np.abs(vp - ref).max() <= 1e-10 * np.abs(ref).max()
    zero = np.zeros(n)
    assert np.abs(eng.hessian(zero, project_to_psd=True) - eng.hessian(zero)).max() <= 1e-10 * np.abs(v).max()


def test_newton_with_backtracking_and_projected_fallback(eng, oracle):
    """A load large enough that plain Newton steps fail: the line search halves steps and the solver falls back to the
    projected Hessian (descent strategy 1), as on the 3-D meshes."""
    mesh = M.square_tris(6, 1, jitter=0.1)
    lam, mu = oracle.lame(2e4, 0.3, 2)
    _setup(eng, mesh, E.NEOHOOKEAN, lam, mu)
    n = mesh.n_nodes * 2
    bnd = M.boundary_dofs(mesh, 2, "x0")
    eng.set_dirichlet(bnd)
    rhs = np.zeros(n)
    rhs[0::2] = -2.5e4 / mesh.n_nodes
    x_ref, info_ref = oracle.newton_solve(mesh, oracle.NEOHOOKEAN, lam, mu, rhs, bnd, np.zeros(n), n_threads=4)
    x, info = eng.newton_solve(np.zeros(n), rhs)
    eng.set_dirichlet(np.zeros(0, dtype=np.int32))
    assert info_ref["line_search_iterations"] > 3 * info_ref["iterations"]
    assert info["iterations"] == info_ref["iterations"] and info["status"] == info_ref["status"]
    assert abs(info["line_search_iterations"] - info_ref["line_search_iterations"]) <= 0.1 * info_ref["line_search_iterations"]
    assert np.abs(x - x_ref).max() <= 1e-7 * np.abs(x_ref).max()


@pytest.mark.parametrize("model", ["NeoHookean", "LinearElasticity"])
def test_implicit_euler_steps_match_oracle(eng, oracle, model):
    """Transient steps (InertiaForm + weighted elastic / body forms, ImplicitEuler) on a triangle mesh against the oracle."""
    mesh = MESHES["p2_jitter_polyfem"]()
    lam, mu = oracle.lame(2e4, 0.3, 2)
    mdl = E.MODEL_BY_NAME[model]
    _setup(eng, mesh, mdl, lam, mu)
    n = mesh.n_nodes * 2
    bnd = M.boundary_dofs(mesh, 2, "x0")
    eng.set_dirichlet(bnd)
    rho, dt = 1000.0, 0.02
    rho_e = np.where(np.arange(mesh.n_el) < mesh.n_el // 2, rho, 2.5 * rho)
    mass = oracle.assemble_mass(mesh, 2, rho_e)
    force = np.array([3.0, -9.81])
    rhs = oracle.assemble_rhs(mesh, 2, force, rho_e)
    eng.inertia_set_mass(rho_e)
    free = np.ones(n, dtype=bool)
    free[bnd] = False
    x = np.zeros(n)
    v = np.where(free, 0.05 * np.random.default_rng(9).standard_normal(n), 0.0)
    xo, vo = x.copy(), v.copy()
    try:
        for step in range(3):
            x, v, a, info = eng.implicit_euler_step(dt, x, v, rhs)
            xo, vo, ao, iref = oracle.implicit_euler_step(mesh, mdl, lam, mu, rhs, bnd, mass, dt, xo, vo)
            assert info["status"] == iref["status"] and info["status"] in (2, 3)
            assert info["iterations"] == iref["iterations"] >= 1
            assert info["line_search_iterations"] == iref["line_search_iterations"]
            assert abs(info["energy"] - iref["energy"]) <= 1e-9 * abs(iref["energy"])
            assert np.abs(x - xo).max() <= 1e-8 * np.abs(xo).max()
            assert np.abs(v - vo).max() <= 1e-7 * np.abs(vo).max()
            assert np.abs(a - ao).max() <= 1e-6 * np.abs(ao).max()
            assert not x[bnd].any()
        # lumped mass: row sums of the mass matrix, once per component
        lumped = eng.lumped_mass()
        ref = np.asarray(csc_to_scipy(mass, n).sum(axis=1)).reshape(-1)
        assert rel_err(lumped, ref) <= 1e-12
    finally:
        eng.inertia_disable()
        eng.set_dirichlet(np.zeros(0, dtype=np.int32))


def test_augmented_lagrangian_dirichlet(eng, oracle):
    """ALForm / ALSolver on a triangle mesh: one penalised solve against the oracle (the static penalty is its inertia
    term with mass w * D and predictor = target), then the loop against the direct constrained solve."""
    import scipy.sparse as sp
    mesh = M.square_tris(6, 2, jitter=0.1)
    lam, mu = oracle.lame(2e4, 0.3, 2)
    _setup(eng, mesh, E.NEOHOOKEAN, lam, mu)
    n = mesh.n_nodes * 2
    bnd = M.boundary_dofs(mesh, 2, "x0")
    rhs = np.zeros(n)
    rhs[1::2] = -40.0 / mesh.n_nodes
    target = np.zeros(n)
    target[bnd[1::2]] = 0.02
    w = 1e6
    masked = np.zeros(n)
    masked[bnd] = 1.0
    eng.al_enable_host(w, masked, target)
    xs, info = eng.newton_solve(np.zeros(n), rhs)
    eng.al_disable()
    d = sp.diags(w * masked).tocsc()
    d.sort_indices()
    xr, iref = oracle.newton_solve(mesh, oracle.NEOHOOKEAN, lam, mu, rhs, np.zeros(0, dtype=np.int32), np.zeros(n),
                                   mass=(d.indptr.astype(np.int32), d.indices.astype(np.int32), d.data), x_tilde=target, weight=1.0)
    assert info["status"] == iref["status"] and info["iterations"] == iref["iterations"] >= 2
    assert np.abs(xs - xr).max() <= 1e-8 * np.abs(xr).max()
    xa, ia, used = eng.al_solve(np.zeros(n), rhs, bnd, target[bnd], force_al=True)
    assert used == [1e6] and np.array_equal(xa[bnd], target[bnd])
    eng.set_dirichlet(bnd)
    xd, _ = eng.newton_solve(target.copy(), rhs)
    eng.set_dirichlet(np.zeros(0, dtype=np.int32))
    assert np.abs(xa - xd).max() <= 1e-7 * np.abs(xd).max()


def test_neohookean_hip_against_sympy_derivatives(eng):
    """The HIP energy / gradient / Hessian of one triangle against the symbolic derivatives of the 2-D energy density
    (tests/golden/sympy/neohookean_sympy_2d.npz): independent of the oracle."""
    import os
    from tests.test_oracle_kat_2d import _single_triangle_with_gradient, neohookean_sympy_expected_2d
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "sympy", "neohookean_sympy_2d.npz"))
    lam, mu = float(g["lam"]), float(g["mu"])
    for F, psi, dpsi, d2psi in zip(g["F"], g["psi"], g["dpsi"], g["d2psi"]):
        mesh, u = _single_triangle_with_gradient(F)
        _setup(eng, mesh, E.NEOHOOKEAN, lam, mu)
        assert abs(eng.energy(u) - psi / 2.0) <= 1e-12 * max(abs(psi / 2.0), 1e-3 * mu)
        grad_ref, hess_ref = neohookean_sympy_expected_2d(F, dpsi, d2psi)
        assert np.abs(eng.gradient(u) - grad_ref).max() <= 1e-12 * max(np.abs(grad_ref).max(), mu)
        rp, ci = eng.get_pattern()
        H = csc_to_scipy((rp, ci, eng.hessian(u)), 6).toarray()
        assert np.abs(H - hess_ref).max() <= 1e-12 * np.abs(hess_ref).max()


def test_dimension_mismatches_fail_loudly(eng, oracle):
    mesh = MESHES["p2_aligned"]()
    eng.set_mesh(mesh)
    with pytest.raises(E.PfxError):
        eng.set_material(E.NEOHOOKEAN, 1.0, 1.0, dim=3)  # dim must match the mesh
    with pytest.raises(E.PfxError):
        eng.set_saint_venant_tensor(np.eye(6), dim=3)
    # a tetrahedral mesh with dim = 2 is refused as well
    eng.set_mesh(M.cube_tets(2, 1))
    with pytest.raises(E.PfxError):
        eng.set_material(E.NEOHOOKEAN, 1.0, 1.0, dim=2)


# ---- SaintVenant, elasticity tensors and element models with size 2 ------------------------------------------------


def _aniso_tensor_2d(seed=0, scale=2e4):
    """A symmetric positive definite 3 x 3 elasticity tensor (Voigt xx yy xy) with every entry set."""
    a = np.random.default_rng(seed).uniform(-1, 1, (3, 3))
    return scale * (a @ a.T + 1.5 * np.eye(3))


@pytest.mark.parametrize("name", list(MESHES))
@pytest.mark.parametrize("tensor", ["isotropic", "anisotropic"])
def test_hooke_linear_assembly_2d(eng, oracle, name, tensor):
    """HookeLinearElasticity::assemble with size_ == 2 (HookeLinearElasticity.cpp:92-122): pattern bit-exact (it depends
    on the values), values 1e-12."""
    mesh = MESHES[name]()
    lam, mu = oracle.lame(1e5, 0.3, 2)
    Ct = E.elasticity_tensor(lam=lam, mu=mu, dim=2) if tensor == "isotropic" else _aniso_tensor_2d(3)
    assert np.array_equal(E.elasticity_tensor(lam=lam, mu=mu, dim=2), oracle.elasticity_tensor(lam, mu, dim=2))
    eng.set_mesh(mesh)
    eng.set_elasticity_tensor(Ct)
    eng.build_pattern()
    eng.assemble_linear()
    rp, ci, v = eng.get_linear_csr()
    n = mesh.n_nodes * 2
    ref = csc_to_scipy(oracle.assemble_linear(mesh, oracle.HOOKE_LINEAR, Ct, None), n).tocsr()
    ref.sort_indices()
    assert np.array_equal(rp, ref.indptr), "row_ptr differs from the oracle"
    assert np.array_equal(ci, ref.indices), "col_idx differs from the oracle"
    assert rel_err(v, ref.data) <= TOL
    with pytest.raises(E.PfxError):
        eng.energy(np.zeros(n))  # a linear assembler only
    with pytest.raises(E.PfxError):
        eng.set_elasticity_tensor(Ct + np.triu(np.ones((3, 3)), 1))  # not symmetric
    with pytest.raises(E.PfxError):
        eng.set_elasticity_tensor(np.eye(6), dim=3)  # a 3-D tensor on a triangle mesh
    with pytest.raises(ValueError):
        eng.set_elasticity_tensor(np.eye(6))


@pytest.mark.parametrize("name", ["p1_jitter", "p2_aligned", "p2_jitter_polyfem", "p2_stretched"])
@pytest.mark.parametrize("tensor", [False, True])
def test_saint_venant_2d(eng, oracle, name, tensor):
    """SaintVenantElasticity with size() == 2 (SaintVenantElasticity.cpp:215-312, branch :284-293), from (lambda, mu) and
    from a general 3 x 3 elasticity tensor: energy, gradient, Hessian (plain and project_to_psd) at rest, at a moderate and
    at a large deformation, per-element parameters, and a Newton solve, against the oracle."""
    mesh = MESHES[name]()
    lam, mu = oracle.lame(2e4, 0.3, 2)
    SV = oracle.SAINT_VENANT
    nnz = _setup(eng, mesh, E.SAINT_VENANT, lam, mu)
    n = mesh.n_nodes * 2
    h = 1.0 / round((mesh.n_el / 2) ** 0.5)
    if tensor:
        x = M.seeded_displacement(mesh, 2, 0.1 * h, seed=5)
        e_iso, g_iso, v_iso = eng.energy(x), eng.gradient(x), eng.hessian(x)
        eng.set_saint_venant_tensor(E.elasticity_tensor(lam=lam, mu=mu, dim=2))
        assert abs(eng.energy(x) - e_iso) <= TOL * e_iso
        assert rel_err(eng.gradient(x), g_iso) <= TOL and rel_err(eng.hessian(x), v_iso) <= TOL
        a = _aniso_tensor_2d(11)
        eng.set_saint_venant_tensor(E.elasticity_tensor(entries=a[np.triu_indices(3)], dim=2))
        b = None
    else:
        a, b = lam, mu
    for amp in (0.0, 1e-2 * h, 0.3 * h):
        x = M.seeded_displacement(mesh, 2, amp)
        e_ref = oracle.assemble_energy(mesh, SV, a, b, x)
        g_ref = oracle.assemble_gradient(mesh, SV, a, b, x)
        outer, inner, vals = oracle.assemble_hessian(mesh, SV, a, b, x)
        e, g, v = eng.energy(x), eng.gradient(x), eng.hessian(x)
        rp, ci = eng.get_pattern()
        assert np.array_equal(rp, outer.astype(np.int64)) and np.array_equal(ci, inner) and nnz == inner.size
        if amp == 0.0:
            assert e == 0.0 and e_ref == 0.0 and not g.any()
        else:
            assert abs(e - e_ref) <= TOL * abs(e_ref)
            assert rel_err(g, g_ref) <= TOL
        assert rel_err(v, vals) <= TOL
    vp = eng.hessian(x, project_to_psd=True)
    ref = oracle.assemble_hessian_psd(mesh, SV, a, b, x)
    assert np.abs(vp - ref[2]).max() <= 1e-10 * np.abs(ref[2]).max()
    if tensor:
        with pytest.raises(E.PfxError):
            eng.set_element_models([E.NEOHOOKEAN] * mesh.n_el)  # element models take (lambda, mu)
    else:
        half = np.arange(mesh.n_el) < mesh.n_el // 2
        l2, m2 = oracle.lame(5e4, 0.2, 2)
        lam_e, mu_e = np.where(half, lam, l2), np.where(half, mu, m2)
        eng.set_material(E.SAINT_VENANT, lam_e, mu_e)
        x = M.seeded_displacement(mesh, 2, 0.1 * h, seed=4)
        assert abs(eng.energy(x) - oracle.assemble_energy(mesh, SV, lam_e, mu_e, x)) <= TOL * abs(eng.energy(x))
        assert rel_err(eng.gradient(x), oracle.assemble_gradient(mesh, SV, lam_e, mu_e, x)) <= TOL
        assert rel_err(eng.hessian(x), oracle.assemble_hessian(mesh, SV, lam_e, mu_e, x)[2]) <= TOL
        eng.set_material(E.SAINT_VENANT, lam, mu)
    bnd = M.boundary_dofs(mesh, 2, "x0")
    eng.set_dirichlet(bnd)
    rhs = np.zeros(n)
    rhs[1::2] = -100.0 / mesh.n_nodes
    xs, info = eng.newton_solve(np.zeros(n), rhs)
    xr, iref = oracle.newton_solve(mesh, SV, a, b, rhs, bnd, np.zeros(n))
    eng.set_dirichlet(np.zeros(0, dtype=np.int32))
    assert info["status"] == iref["status"] and info["iterations"] == iref["iterations"] >= 2
    assert np.abs(xs - xr).max() <= 1e-8 * np.abs(xr).max()


@pytest.mark.parametrize("name", ["p1_jitter", "p2_jitter_polyfem", "p2_stretched"])
def test_multi_model_2d(eng, oracle, name):
    """MultiModel (assembler/MultiModel.cpp:42-96) on triangles: LinearElasticity / NeoHookean / SaintVenant chosen per
    element, with uniform and per-element Lame parameters; energy, gradient, Hessian (plain and projected), Newton."""
    mesh = MESHES[name]()
    n = mesh.n_nodes * 2
    h = 1.0 / round((mesh.n_el / 2) ** 0.5)
    ids = np.array([oracle.NEOHOOKEAN, oracle.SAINT_VENANT, oracle.LINEAR_ELASTICITY])[(np.arange(mesh.n_el) * 7 // 3) % 3]
    lam, mu = oracle.lame(2e4, 0.3, 2)
    l2, m2 = oracle.lame(5e4, 0.2, 2)
    half = np.arange(mesh.n_el) % 2 == 0
    for lam_e, mu_e in ((lam, mu), (np.where(half, lam, l2), np.where(half, mu, m2))):
        eng.set_mesh(mesh)
        eng.set_material(E.NEOHOOKEAN, lam_e, mu_e)
        eng.set_element_models(ids)
        nnz = eng.build_pattern()
        for amp in (0.0, 2e-2 * h):
            x = M.seeded_displacement(mesh, 2, amp, seed=8)
            e_ref = oracle.assemble_energy(mesh, ids, lam_e, mu_e, x)
            g_ref = oracle.assemble_gradient(mesh, ids, lam_e, mu_e, x)
            outer, inner, vals = oracle.assemble_hessian(mesh, ids, lam_e, mu_e, x)
            e, g, v = eng.energy(x), eng.gradient(x), eng.hessian(x)
            rp, ci = eng.get_pattern()
            assert np.array_equal(rp, outer.astype(np.int64)) and np.array_equal(ci, inner) and nnz == inner.size
            if amp == 0.0:
                assert abs(e) <= 1e-9 and np.abs(g).max() <= 1e-9 * np.abs(vals).max()
            else:
                assert abs(e - e_ref) <= TOL * abs(e_ref)
                assert rel_err(g, g_ref) <= TOL
            assert rel_err(v, vals) <= TOL
        vp = eng.hessian(x, project_to_psd=True)
        ref = oracle.assemble_hessian_psd(mesh, ids, lam_e, mu_e, x)
        assert np.abs(vp - ref[2]).max() <= 1e-10 * np.abs(ref[2]).max()
    eng.set_material(E.NEOHOOKEAN, lam, mu)
    eng.set_element_models(["SaintVenant"] * mesh.n_el)
    e_multi, g_multi = eng.energy(x), eng.gradient(x)
    eng.set_material(E.SAINT_VENANT, lam, mu)
    assert abs(e_multi - eng.energy(x)) <= 1e-14 * abs(e_multi) and rel_err(g_multi, eng.gradient(x)) <= 1e-14
    eng.set_material(E.NEOHOOKEAN, lam, mu)
    eng.set_element_models(ids)
    bnd = M.boundary_dofs(mesh, 2, "x0")
    eng.set_dirichlet(bnd)
    rhs = np.zeros(n)
    rhs[1::2] = -100.0 / mesh.n_nodes
    xs, info = eng.newton_solve(np.zeros(n), rhs)
    xr, iref = oracle.newton_solve(mesh, ids, lam, mu, rhs, bnd, np.zeros(n))
    eng.set_dirichlet(np.zeros(0, dtype=np.int32))
    assert info["status"] == iref["status"] and info["iterations"] == iref["iterations"] >= 2
    assert np.abs(xs - xr).max() <= 1e-8 * np.abs(xr).max()


# ---- aggregation AMG for 2 x 2 blocks ------------------------------------------------------------------------------


@pytest.mark.parametrize("order", [1, 2])
@pytest.mark.parametrize("bc", ["clamped", "roller"])
def test_amg_2d_matches_mirror(oracle, order, bc):
    """Block size 2: three coarse dofs per aggregate on every level (the rigid-body modes of the plane: two translations and
    the rotation about the aggregate's centre), against the scipy mirror of the same algorithm built from the library's aggregates (one application 1e-10,
    PCG iterations +-2), against the oracle's matrix (relative residual 1e-10) and against Jacobi-CG.  "roller" fixes only
    the x component on the x = 0 edge (plus one y dof): partly constrained nodes stay in their aggregates and the
    constrained dofs stay out of the Galerkin sums and the transfers."""
    import scipy.sparse as sp
    import torch
    from tests import amg_mirror as AM
    eng = E.Engine(0)
    try:
        mesh = M.square_tris(24 if order == 2 else 40, order, jitter=0.1, seed=3)
        lam, mu = oracle.lame(1e5, 0.3, 2)
        n = 2 * mesh.n_nodes
        _setup(eng, mesh, E.LINEAR_ELASTICITY, lam, mu)
        eng.assemble_linear()
        A = csc_to_scipy(oracle.assemble_linear(mesh, oracle.LINEAR_ELASTICITY, lam, mu), n).tocsr()
        bd = M.boundary_dofs(mesh, 2, "x0")
        if bc == "roller":
            bd = np.concatenate([bd[bd % 2 == 0], bd[bd % 2 == 1][:1]]).astype(np.int32)
            bd.sort()
        eng.set_dirichlet(bd)
        fixed = np.zeros(n, dtype=bool)
        fixed[bd] = True
        D = sp.diags((~fixed).astype(float))
        Am = (D @ A @ D + sp.diags(fixed.astype(float))).tocsr()
        eng.amg_set_params(coarsest_dofs=40, mixed_precision=0)
        eng.amg_setup()
        info = eng.amg_info()
        aggs = [eng.amg_aggregates(l) for l in range(info["n_levels"] - 1)]
        assert info["n_levels"] >= 3
        all_fixed = fixed.reshape(-1, 2).all(axis=1)
        assert ((aggs[0] >= 0) == ~all_fixed).all()
        pos = np.hstack([M.node_positions(mesh), np.zeros((mesh.n_nodes, 1))])
        levels = AM.build(Am, pos, fixed, aggs, info["lambda_max"], ncycle=2, plane=True)
        rng = np.random.default_rng(8)
        r = rng.standard_normal(n)
        r[fixed] = 0
        d_r = torch.from_numpy(r).cuda()
        d_z = torch.zeros_like(d_r)
        eng.amg_apply_dev(d_r.data_ptr(), d_z.data_ptr())
        eng.synchronize()
        z = d_z.cpu().numpy()
        z_ref = AM.cycle(levels, 0, r)
        assert np.abs(z - z_ref).max() <= 1e-10 * np.abs(z_ref).max()
        assert np.all(z[fixed] == 0)
        b = rng.standard_normal(n)
        x, it, err = eng.cg(b, None, E.PRECOND_AMG, 1000, 1e-10)
        bm = np.where(fixed, 0.0, b)
        _, it_ref, _ = AM.pcg(Am, bm, lambda v: AM.cycle(levels, 0, v), 1e-10, 1000)
        assert abs(it - it_ref) <= 2 and err <= 1e-10, (it, it_ref, err)
        assert np.linalg.norm(bm - Am @ x) / np.linalg.norm(bm) <= 1.01e-10
        _, it_jac, _ = eng.cg(b, None, E.PRECOND_JACOBI, 20000, 1e-10)
        print("2-D AMG:", order, bc, "levels", info["n_vertices"], "iterations", it, "Jacobi", it_jac)
        assert it < it_jac / 2, (it, it_jac)
    finally:
        eng.close()


def test_amg_2d_large_and_newton(oracle):
    """2 M P1 triangles / 0.5 M P2 triangles (1 M nodes each, 2 M dofs): the AMG-preconditioned CG reaches 1e-10 in a small
    fraction of the Jacobi iterations, with the single-precision level matrices; and a Newton solve with the AMG-CG as its
    linear solver reaches the minimiser of the Jacobi-CG one."""
    import torch
    for nq, order in ((1000, 1), (500, 2)):
        mesh = M.square_tris(nq, order)
        eng = E.Engine(0)
        try:
            lam, mu = oracle.lame(1e5, 0.3, 2)
            _setup(eng, mesh, E.LINEAR_ELASTICITY, lam, mu)
            eng.set_dirichlet(M.boundary_dofs(mesh, 2, "x0"))
            eng.assemble_linear()
            nd = 2 * mesh.n_nodes
            b = torch.from_numpy(np.random.default_rng(2).standard_normal(nd)).cuda()
            x = torch.zeros(nd, dtype=torch.float64, device="cuda")
            it, err = eng.cg_dev(b.data_ptr(), x.data_ptr(), E.PRECOND_AMG, 2000, 1e-10)
            xj = torch.zeros_like(x)
            it_j, err_j = eng.cg_dev(b.data_ptr(), xj.data_ptr(), E.PRECOND_JACOBI, 40000, 1e-10)
            print("2-D AMG large:", nq, order, "iterations", it, err, "Jacobi", it_j, err_j, eng.amg_info()["n_vertices"][:8])
            assert err <= 1e-10 and err_j <= 1e-10 and it < 300 and it < it_j / 5
            assert float(torch.linalg.norm(x - xj) / torch.linalg.norm(xj)) < 1e-6
        finally:
            eng.close()
    mesh = M.square_tris(40, 2, jitter=0.1)
    eng = E.Engine(0)
    try:
        lam, mu = oracle.lame(2e4, 0.3, 2)
        _setup(eng, mesh, E.NEOHOOKEAN, lam, mu)
        n = 2 * mesh.n_nodes
        eng.set_dirichlet(M.boundary_dofs(mesh, 2, "x0"))
        rhs = np.zeros(n)
        rhs[1::2] = -40.0 / mesh.n_nodes
        xa, ia = eng.newton_solve(np.zeros(n), rhs, cg_precond=E.PRECOND_AMG)
        xj, ij = eng.newton_solve(np.zeros(n), rhs, cg_precond=E.PRECOND_JACOBI)
        assert ia["status"] == ij["status"] and abs(ia["iterations"] - ij["iterations"]) <= 1
        assert np.abs(xa - xj).max() <= 1e-7 * np.abs(xj).max()
    finally:
        eng.close()


def test_host_mirror_of_reference_interface_2d(oracle):
    """tests/test_assembler.cpp:12-64 ("hessian_lin") through the reference-shaped host classes on a triangle mesh:
    is_volume = False, set_size(2), (E, nu) converted with the plane formula."""
    mesh = M.square_tris(6, 2, jitter=0.1)
    asm = E.Assembler("LinearElasticity")
    asm.set_size(2)
    asm.add_multimaterial(0, {"E": 1e5, "nu": 0.3})
    K = asm.assemble(False, mesh.n_nodes, mesh, mesh, None)
    lam, mu = oracle.lame(1e5, 0.3, 2)
    assert asm.lam == lam and asm.mu == mu
    ref = oracle.assemble_linear(mesh, oracle.LINEAR_ELASTICITY, lam, mu)
    assert K.rows() == mesh.n_nodes * 2 and np.array_equal(K.inner, ref[1]) and rel_err(K.values, ref[2]) <= TOL
    nl = E.NLAssembler("LinearElasticity", engine=asm.engine)
    nl.set_size(2)
    nl.add_multimaterial(0, {"E": 1e5, "nu": 0.3})
    cache = E.SpareMatrixCache()
    Kd = K.to_scipy()
    rng = np.random.default_rng(0)
    for rand in range(4):
        x = (np.zeros(K.rows()) if rand == 0 else rng.uniform(-1, 1, K.rows())).reshape(-1, 1)
        H = nl.assemble_hessian(False, mesh.n_nodes, False, mesh, mesh, None, 0.0, x, x, cache)
        assert abs(H.to_scipy() - Kd).max() <= 1e-8 * abs(Kd).max()
    with pytest.raises(ValueError):
        nl.assemble(True, mesh, mesh, None, 0.0, x, x)  # is_volume = True on a triangle mesh
    bad = E.NLAssembler("NeoHookean", engine=asm.engine)  # size() left at 3
    with pytest.raises(ValueError):
        bad.assemble(False, mesh, mesh, None, 0.0, x, x)
    asm.engine.close()
