"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same seeded inputs.

Bars (north_star): sparsity pattern bit-exact; FP64 matrix entries, energies, gradients within 1e-12 relative
(measured against the largest entry of the compared array); CG relative residual <= 1e-10 and iteration count
within +-2 of the oracle's Eigen-order CG.
"""
import numpy as np
import pytest

from polyfem_amd import engine as E
from polyfem_amd import mesh as M
from tests.helpers import csc_to_scipy, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-12


@pytest.fixture(scope="module")
def eng():
    e = E.Engine(0)
    yield e
    e.close()


def _setup(eng, mesh, model, lam, mu):
    eng.set_mesh(mesh)
    eng.set_material(model, lam, mu)
    return eng.build_pattern()


MESHES = {
    "p1_aligned": lambda: M.cube_tets(4, 1),
    "p1_jitter": lambda: M.cube_tets(4, 1, jitter=0.2, seed=1),
    "p2_aligned": lambda: M.cube_tets(3, 2),
    "p2_jitter_polyfem": lambda: M.cube_tets(3, 2, numbering="polyfem", jitter=0.2, seed=2),
    "p2_stretched": lambda: M.cube_tets(2, 2, scale=(1.0, 2.0, 0.5)),
    "q1": lambda: M.cube_hexes(3, 1),
    "q2": lambda: M.cube_hexes(2, 2),
    "single_tet_cell": lambda: M.cube_tets(1, 2),
}


@pytest.mark.parametrize("name", list(MESHES))
@pytest.mark.parametrize("model", ["Laplacian", "LinearElasticity"])
def test_linear_assembly_pattern_bit_exact_and_values(eng, oracle, name, model):
    mesh = MESHES[name]()
    mid = E.MODEL_BY_NAME[model]
    lam, mu = oracle.lame(1e5, 0.3)
    _setup(eng, mesh, mid, lam, mu)
    eng.assemble_linear()
    rp, ci, v = eng.get_linear_csr()
    outer, inner, vals = oracle.assemble_linear(mesh, mid, lam, mu)
    assert np.array_equal(rp, outer.astype(np.int64)), "row_ptr differs from the oracle"
    assert np.array_equal(ci, inner), "col_idx differs from the oracle"
    assert rel_err(v, vals) <= TOL
    # the value-dependent pattern is a subset of the structural one
    assert v.size <= eng.info().nnz


def _orthotropic_tensor(seed):
    """A random symmetric positive-definite 6 x 6 tensor (fully anisotropic)."""
    a = np.random.default_rng(seed).standard_normal((6, 6))
    return 1e4 * (a @ a.T + 6 * np.eye(6))


@pytest.mark.parametrize("name", ["p1_jitter", "p2_aligned", "p2_jitter_polyfem", "p2_stretched", "q1", "q2"])
@pytest.mark.parametrize("tensor", ["isotropic", "anisotropic"])
def test_hooke_linear_assembly(eng, oracle, name, tensor):
    mesh = MESHES[name]()
    lam, mu = oracle.lame(1e5, 0.3)
    C = oracle.elasticity_tensor(lam, mu) if tensor == "isotropic" else _orthotropic_tensor(3)
    eng.set_mesh(mesh)
    eng.set_elasticity_tensor(C)
    eng.build_pattern()
    eng.assemble_linear()
    rp, ci, v = eng.get_linear_csr()
    outer, inner, vals = oracle.assemble_linear(mesh, oracle.HOOKE_LINEAR, C, 0.0)
    n = mesh.n_nodes * 3
    # the local blocks of this assembler are symmetric only up to rounding, so an entry can be exactly zero (dropped)
    # while its mirror image is not: compare row-major with row-major
    ref = csc_to_scipy((outer, inner, vals), n).tocsr()
    ref.sort_indices()
    assert np.array_equal(rp, ref.indptr), "row_ptr differs from the oracle"
    assert np.array_equal(ci, ref.indices), "col_idx differs from the oracle"
    assert rel_err(v, ref.data) <= TOL
    if tensor == "isotropic":
        # the isotropic tensor is LinearElasticity: same action on a vector (the value-dependent patterns may differ)
        xv = np.random.default_rng(5).standard_normal(n)
        a = ref
        b = csc_to_scipy(oracle.assemble_linear(mesh, oracle.LINEAR_ELASTICITY, lam, mu), n)
        assert rel_err(a @ xv, b @ xv) <= 1e-12
    with pytest.raises(E.PfxError):
        eng.energy(np.zeros(mesh.n_nodes * 3))  # the tensor model is a linear assembler only here
    with pytest.raises(E.PfxError):
        eng.set_elasticity_tensor(C + np.triu(np.ones((6, 6)), 1))  # not symmetric


@pytest.mark.parametrize("name", ["p1_jitter", "p2_aligned", "p2_jitter_polyfem", "p2_stretched", "q1", "q2"])
def test_neohookean_energy_gradient_hessian(eng, oracle, name):
    mesh = MESHES[name]()
    lam, mu = oracle.lame(2e4, 0.3)
    nnz = _setup(eng, mesh, E.NEOHOOKEAN, lam, mu)
    n = mesh.n_nodes * 3
    h = 1.0 / round((mesh.n_el / (6 if mesh.kind < 2 else 1)) ** (1 / 3))
    for amp in (0.0, 1e-2 * h):
        x = M.seeded_displacement(mesh, 3, amp)
        e_ref = oracle.assemble_energy(mesh, oracle.NEOHOOKEAN, lam, mu, x)
        g_ref = oracle.assemble_gradient(mesh, oracle.NEOHOOKEAN, lam, mu, x)
        outer, inner, vals = oracle.assemble_hessian(mesh, oracle.NEOHOOKEAN, lam, mu, x)
        e = eng.energy(x)
        g = eng.gradient(x)
        v = eng.hessian(x)
        rp, ci = eng.get_pattern()
        assert np.array_equal(rp, outer.astype(np.int64)) and np.array_equal(ci, inner), "Hessian pattern not bit-exact"
        assert nnz == inner.size
        if amp == 0.0:
            assert abs(e) <= 1e-9 and abs(e_ref) <= 1e-9
            assert np.abs(g).max() <= 1e-9 * np.abs(vals).max()
        else:
            assert abs(e - e_ref) <= TOL * abs(e_ref)
            assert rel_err(g, g_ref) <= TOL
        assert rel_err(v, vals) <= TOL


def _star_mesh(n=4):
    """P1 tetrahedra that all share the centre of a cube: one per triangle of its n x n x 6 surface grid."""
    pts, tris = {}, []

    def pid(p):
        return pts.setdefault(tuple(p), len(pts) + 1)  # 0 is the centre

    for axis in range(3):
        for side in (0, n):
            a, b = [d for d in range(3) if d != axis]
            for i in range(n):
                for j in range(n):
                    def P(u, v):
                        q = [0, 0, 0]
                        q[axis], q[a], q[b] = side, u, v
                        return pid(q)
                    quad = [P(i, j), P(i + 1, j), P(i + 1, j + 1), P(i, j + 1)]
                    tris += [(quad[0], quad[1], quad[2]), (quad[0], quad[2], quad[3])]
    xyz = np.zeros((len(pts) + 1, 3))
    xyz[0] = 0.5
    for q, k in pts.items():
        xyz[k] = np.array(q, dtype=np.float64) / n
    conn = np.array([(0,) + t for t in tris], dtype=np.int32)
    d = xyz[conn[:, 1:]] - xyz[conn[:, :1]]
    flip = np.linalg.det(d) < 0
    conn[flip, 2], conn[flip, 3] = conn[flip, 3].copy(), conn[flip, 2].copy()
    return M.Mesh(kind=0, gkind=0, conn=conn, gconn=conn.copy(), xyz=xyz, n_nodes=xyz.shape[0], node_xyz=xyz.copy())


def test_block_spanning_several_assembly_windows(eng, oracle):
    """The row assembly walks the contributions in windows of 64 and carries the partial sums of a block that a window
    boundary cuts.  A vertex shared by 192 tetrahedra gives a diagonal block with 192 contributions -- it starts in
    one window, fills two and ends in a fourth: stiffness (with the value-dependent pattern), mass and the NeoHookean
    Hessian against the oracle, and the result is the same bits on a second run."""
    mesh = _star_mesh(4)
    assert mesh.n_el == 192 and (mesh.conn[:, 0] == 0).all()
    lam, mu = oracle.lame(1e5, 0.3)
    _setup(eng, mesh, E.LINEAR_ELASTICITY, lam, mu)
    eng.assemble_linear()
    rp, ci, v = eng.get_linear_csr()
    outer, inner, vals = oracle.assemble_linear(mesh, oracle.LINEAR_ELASTICITY, lam, mu)
    assert np.array_equal(rp, outer.astype(np.int64)) and np.array_equal(ci, inner)
    assert rel_err(v, vals) <= TOL
    eng.assemble_linear()
    assert np.array_equal(eng.get_linear_csr()[2], v)
    eng.set_material(E.NEOHOOKEAN, lam, mu)
    x = M.seeded_displacement(mesh, 3, 2e-2)
    h = eng.hessian(x)
    assert rel_err(h, oracle.assemble_hessian(mesh, oracle.NEOHOOKEAN, lam, mu, x)[2]) <= TOL
    assert np.array_equal(eng.hessian(x), h)
    assert rel_err(eng.gradient(x), oracle.assemble_gradient(mesh, oracle.NEOHOOKEAN, lam, mu, x)) <= TOL


@pytest.mark.parametrize("name", ["p1_jitter", "p2_jitter_polyfem", "q1", "q2"])
def test_saint_venant_elasticity_tensor(eng, oracle, name):
    """SaintVenantElasticity with the material's `elasticity_tensor` (SaintVenantElasticity.cpp:64-68, 77-86, 281-309): an
    anisotropic symmetric tensor -- energy, gradient, Hessian (plain and project_to_psd) and a Newton solve against the
    oracle; the isotropic tensor reproduces the (lambda, mu) model; pfx_set_material returns to (lambda, mu)."""
    mesh = MESHES[name]()
    lam, mu = oracle.lame(2e4, 0.3)
    nnz = _setup(eng, mesh, E.SAINT_VENANT, lam, mu)
    n = mesh.n_nodes * 3
    h = 1.0 / round((mesh.n_el / (6 if mesh.kind < 2 else 1)) ** (1 / 3))
    x = M.seeded_displacement(mesh, 3, 0.1 * h, seed=5)
    e_iso, g_iso, v_iso = eng.energy(x), eng.gradient(x), eng.hessian(x)
    eng.set_saint_venant_tensor(E.elasticity_tensor(lam=lam, mu=mu))
    assert abs(eng.energy(x) - e_iso) <= TOL * e_iso
    assert rel_err(eng.gradient(x), g_iso) <= TOL and rel_err(eng.hessian(x), v_iso) <= TOL
    a = np.random.default_rng(11).standard_normal((6, 6))
    Ct = 2e3 * (a @ a.T + 6 * np.eye(6))
    eng.set_saint_venant_tensor(E.elasticity_tensor(entries=Ct[np.triu_indices(6)]))
    SV = oracle.SAINT_VENANT
    for amp in (0.0, 1e-2 * h, 0.3 * h):
        x = M.seeded_displacement(mesh, 3, amp)
        e_ref = oracle.assemble_energy(mesh, SV, Ct, None, x)
        g_ref = oracle.assemble_gradient(mesh, SV, Ct, None, x)
        outer, inner, vals = oracle.assemble_hessian(mesh, SV, Ct, None, x)
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
    ref = oracle.assemble_hessian_psd(mesh, SV, Ct, None, x)
    assert np.abs(vp - ref[2]).max() <= 1e-10 * np.abs(ref[2]).max()
    with pytest.raises(E.PfxError):
        eng.set_element_models([E.NEOHOOKEAN] * mesh.n_el)  # element models take (lambda, mu)
    if name != "q1":
        bnd = M.boundary_dofs(mesh, 3, "x0")
        eng.set_dirichlet(bnd)
        rhs = np.zeros(n)
        rhs[2::3] = -100.0 / mesh.n_nodes
        xs, info = eng.newton_solve(np.zeros(n), rhs)
        xr, iref = oracle.newton_solve(mesh, SV, Ct, None, rhs, bnd, np.zeros(n))
        eng.set_dirichlet(np.zeros(0, dtype=np.int32))
        assert info["status"] == iref["status"] and info["iterations"] == iref["iterations"] >= 2
        assert np.abs(xs - xr).max() <= 1e-8 * np.abs(xr).max()
    eng.set_material(E.SAINT_VENANT, lam, mu)
    x = M.seeded_displacement(mesh, 3, 0.1 * h, seed=5)
    assert abs(eng.energy(x) - e_iso) <= 1e-15 * e_iso


@pytest.mark.parametrize("name", ["p1_jitter", "p2_jitter_polyfem", "p2_stretched", "q1", "q2"])
def test_saint_venant_energy_gradient_hessian(eng, oracle, name):
    """SaintVenantElasticity (SaintVenantElasticity.cpp:104-143, 215-312; isotropic tensor): energy, gradient, Hessian
    (plain and project_to_psd) at rest, at a moderate and at a large deformation, per-element materials, and a Newton
    solve, against the oracle."""
    mesh = MESHES[name]()
    lam, mu = oracle.lame(2e4, 0.3)
    nnz = _setup(eng, mesh, E.SAINT_VENANT, lam, mu)
    n = mesh.n_nodes * 3
    h = 1.0 / round((mesh.n_el / (6 if mesh.kind < 2 else 1)) ** (1 / 3))
    for amp in (0.0, 1e-2 * h, 0.3 * h):
        x = M.seeded_displacement(mesh, 3, amp)
        e_ref = oracle.assemble_energy(mesh, oracle.SAINT_VENANT, lam, mu, x)
        g_ref = oracle.assemble_gradient(mesh, oracle.SAINT_VENANT, lam, mu, x)
        outer, inner, vals = oracle.assemble_hessian(mesh, oracle.SAINT_VENANT, lam, mu, x)
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
    ref = oracle.assemble_hessian_psd(mesh, oracle.SAINT_VENANT, lam, mu, x)
    assert np.abs(vp - ref[2]).max() <= 1e-10 * np.abs(ref[2]).max()
    # one (lambda, mu) per element
    half = np.arange(mesh.n_el) < mesh.n_el // 2
    l2, m2 = oracle.lame(5e4, 0.2)
    lam_e, mu_e = np.where(half, lam, l2), np.where(half, mu, m2)
    eng.set_material(E.SAINT_VENANT, lam_e, mu_e)
    x = M.seeded_displacement(mesh, 3, 0.1 * h, seed=4)
    assert abs(eng.energy(x) - oracle.assemble_energy(mesh, oracle.SAINT_VENANT, lam_e, mu_e, x)) <= TOL * abs(eng.energy(x))
    assert rel_err(eng.gradient(x), oracle.assemble_gradient(mesh, oracle.SAINT_VENANT, lam_e, mu_e, x)) <= TOL
    assert rel_err(eng.hessian(x), oracle.assemble_hessian(mesh, oracle.SAINT_VENANT, lam_e, mu_e, x)[2]) <= TOL
    # Newton solve (the Q1 one-point rule leaves the clamped static problem singular: skipped there)
    if name != "q1":
        eng.set_material(E.SAINT_VENANT, lam, mu)
        bnd = M.boundary_dofs(mesh, 3, "x0")
        eng.set_dirichlet(bnd)
        rhs = np.zeros(n)
        rhs[2::3] = -100.0 / mesh.n_nodes
        xs, info = eng.newton_solve(np.zeros(n), rhs)
        xr, iref = oracle.newton_solve(mesh, oracle.SAINT_VENANT, lam, mu, rhs, bnd, np.zeros(n))
        eng.set_dirichlet(np.zeros(0, dtype=np.int32))
        assert info["status"] == iref["status"] and info["iterations"] == iref["iterations"] >= 2
        assert info["line_search_iterations"] == iref["line_search_iterations"]
        assert np.abs(xs - xr).max() <= 1e-8 * np.abs(xr).max()


@pytest.mark.parametrize("name", ["p1_jitter", "p2_jitter_polyfem", "q1", "q2"])
def test_multi_model_energy_gradient_hessian(eng, oracle, name):
    """MultiModel (assembler/MultiModel.cpp:42-96): LinearElasticity / NeoHookean / SaintVenant chosen per element,
    with per-element Lame parameters; energy, gradient, Hessian (plain and projected) and a Newton solve."""
    mesh = MESHES[name]()
    n = mesh.n_nodes * 3
    h = 1.0 / round((mesh.n_el / (6 if mesh.kind < 2 else 1)) ** (1 / 3))
    ids = np.array([oracle.NEOHOOKEAN, oracle.SAINT_VENANT, oracle.LINEAR_ELASTICITY])[(np.arange(mesh.n_el) * 7 // 3) % 3]
    lam, mu = oracle.lame(2e4, 0.3)
    l2, m2 = oracle.lame(5e4, 0.2)
    half = np.arange(mesh.n_el) % 2 == 0
    for lam_e, mu_e in ((lam, mu), (np.where(half, lam, l2), np.where(half, mu, m2))):
        eng.set_mesh(mesh)
        eng.set_material(E.NEOHOOKEAN, lam_e, mu_e)
        eng.set_element_models(ids)
        nnz = eng.build_pattern()
        for amp in (0.0, 2e-2 * h):
            x = M.seeded_displacement(mesh, 3, amp, seed=8)
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
    # a single model through the per-element interface equals that model
    eng.set_material(E.NEOHOOKEAN, lam, mu)
    eng.set_element_models(["SaintVenant"] * mesh.n_el)
    e_multi, g_multi = eng.energy(x), eng.gradient(x)
    eng.set_material(E.SAINT_VENANT, lam, mu)
    # (same formulas; the run-time model test lets the compiler contract the arithmetic differently)
    assert abs(e_multi - eng.energy(x)) <= 1e-14 * abs(e_multi) and rel_err(g_multi, eng.gradient(x)) <= 1e-14
    # error behaviour
    with pytest.raises(E.PfxError):
        eng.set_element_models(ids[:-1])
    with pytest.raises(E.PfxError):
        eng.set_element_models(np.full(mesh.n_el, E.LAPLACIAN))
    if name != "q1":  # the Q1 one-point rule leaves the clamped static problem singular
        eng.set_material(E.NEOHOOKEAN, lam, mu)
        eng.set_element_models(ids)
        bnd = M.boundary_dofs(mesh, 3, "x0")
        eng.set_dirichlet(bnd)
        rhs = np.zeros(n)
        rhs[2::3] = -100.0 / mesh.n_nodes
        xs, info = eng.newton_solve(np.zeros(n), rhs)
        xr, iref = oracle.newton_solve(mesh, ids, lam, mu, rhs, bnd, np.zeros(n))
        eng.set_dirichlet(np.zeros(0, dtype=np.int32))
        assert info["status"] == iref["status"] and info["iterations"] == iref["iterations"] >= 2
        assert info["line_search_iterations"] == iref["line_search_iterations"]
        assert np.abs(xs - xr).max() <= 1e-8 * np.abs(xr).max()


@pytest.mark.parametrize("name", ["p1_jitter", "p2_jitter_polyfem", "q1", "q2"])
def test_linear_elasticity_through_nonlinear_interface(eng, oracle, name):
    """LinearElasticity::compute_energy / assemble_grad / assemble_hessian (LinearElasticity.cpp:56-169) against the
    oracle, and the reference's own property (tests/test_assembler.cpp:12-64): Hessian == stiffness for any x."""
    mesh = MESHES[name]()
    lam, mu = oracle.lame(1e5, 0.3)
    nnz = _setup(eng, mesh, E.LINEAR_ELASTICITY, lam, mu)
    n = mesh.n_nodes * 3
    eng.assemble_linear()
    k = eng.get_values()
    rng = np.random.default_rng(5)
    for rand in (0, 1):
        x = rng.standard_normal(n) if rand else np.zeros(n)
        e, g, v = eng.energy(x), eng.gradient(x), eng.hessian(x)
        outer, inner, vals = oracle.assemble_hessian(mesh, oracle.LINEAR_ELASTICITY, lam, mu, x)
        rp, ci = eng.get_pattern()
        assert np.array_equal(rp, outer.astype(np.int64)) and np.array_equal(ci, inner) and nnz == inner.size
        assert rel_err(v, vals) <= TOL
        assert rel_err(v, k) <= TOL
        if rand:
            assert abs(e - oracle.assemble_energy(mesh, oracle.LINEAR_ELASTICITY, lam, mu, x)) <= TOL * abs(e)
            assert rel_err(g, oracle.assemble_gradient(mesh, oracle.LINEAR_ELASTICITY, lam, mu, x)) <= TOL
        else:
            assert e == 0.0 and not g.any()


@pytest.mark.parametrize("name", ["p2_jitter_polyfem", "q1"])
def test_per_element_materials(eng, oracle, name):
    """LameParameters with one entry per element (multi-material bodies, utils/ElasticityUtils.cpp:490-517)."""
    mesh = MESHES[name]()
    rng = np.random.default_rng(11)
    E_el = 1e4 * (1.0 + 4.0 * rng.random(mesh.n_el))
    nu_el = 0.2 + 0.2 * rng.random(mesh.n_el)
    lam = E_el * nu_el / ((1 + nu_el) * (1 - 2 * nu_el))
    mu = E_el / (2 * (1 + nu_el))
    n = mesh.n_nodes * 3
    eng.set_mesh(mesh)
    eng.set_material(E.LINEAR_ELASTICITY, lam, mu)
    eng.build_pattern()
    eng.assemble_linear()
    rp, ci, v = eng.get_linear_csr()
    outer, inner, vals = oracle.assemble_linear(mesh, oracle.LINEAR_ELASTICITY, lam, mu)
    assert np.array_equal(rp, outer.astype(np.int64)) and np.array_equal(ci, inner)
    assert rel_err(v, vals) <= TOL
    eng.set_material(E.NEOHOOKEAN, lam, mu)
    h = 1.0 / round((mesh.n_el / (6 if mesh.kind < 2 else 1)) ** (1 / 3))
    x = M.seeded_displacement(mesh, 3, 1e-2 * h)
    e, g, hv = eng.energy(x), eng.gradient(x), eng.hessian(x)
    assert abs(e - oracle.assemble_energy(mesh, oracle.NEOHOOKEAN, lam, mu, x)) <= TOL * abs(e)
    assert rel_err(g, oracle.assemble_gradient(mesh, oracle.NEOHOOKEAN, lam, mu, x)) <= TOL
    assert rel_err(hv, oracle.assemble_hessian(mesh, oracle.NEOHOOKEAN, lam, mu, x)[2]) <= TOL
    # a uniform material set afterwards replaces the arrays
    eng.set_material(E.NEOHOOKEAN, float(lam[0]), float(mu[0]))
    assert abs(eng.energy(x) - oracle.assemble_energy(mesh, oracle.NEOHOOKEAN, float(lam[0]), float(mu[0]), x)) <= TOL * abs(e)


@pytest.mark.parametrize("name", ["p1_jitter", "p2_jitter_polyfem", "p2_aligned", "q1", "q2"])
@pytest.mark.parametrize("size", [1, 3])
def test_mass_matrix_and_body_force(eng, oracle, name, size):
    """Assembler<Mass>::assemble(is_mass=true) (mass quadrature of order 2p+1, value-dependent pattern) and
    RhsAssembler::assemble for a constant force, uniform and per-element density."""
    mesh = MESHES[name]()
    lam, mu = oracle.lame(1e5, 0.3)
    _setup(eng, mesh, E.LAPLACIAN if size == 1 else E.LINEAR_ELASTICITY, lam, mu)
    rho_e = 1.0 + np.random.default_rng(9).random(mesh.n_el)
    force = np.array([0.3, -0.2, -9.8])[:size]
    for rho in (2.5, rho_e):
        eng.assemble_mass(rho)
        rp, ci, v = eng.get_linear_csr()
        outer, inner, vals = oracle.assemble_mass(mesh, size, rho)
        assert np.array_equal(rp, outer.astype(np.int64)) and np.array_equal(ci, inner), "mass pattern differs from the oracle"
        assert rel_err(v, vals) <= TOL
        r = eng.assemble_rhs(force, rho)
        assert rel_err(r, oracle.assemble_rhs(mesh, size, force, rho)) <= TOL
    # the resident matrix is the mass matrix now: total mass through the operator
    ones = np.ones(mesh.n_nodes * size)
    assert abs(ones @ eng.spmv(ones) - size * oracle.assemble_rhs(mesh, 1, [1.0], rho_e).sum()) <= 1e-11 * size * rho_e.sum()
    # a stiffness assembly afterwards is unaffected
    eng.assemble_linear()
    k_ref = oracle.assemble_linear(mesh, oracle.LAPLACIAN if size == 1 else oracle.LINEAR_ELASTICITY, lam, mu)
    assert rel_err(eng.get_linear_csr()[2], k_ref[2]) <= TOL


@pytest.mark.parametrize("name", ["p1_jitter", "p2_jitter_polyfem", "q2"])
def test_density_per_quadrature_point(eng, oracle, name):
    """Density::operator() evaluated at every quadrature point (utils/ElasticityUtils.cpp:561-570; Mass.cpp:10,
    RhsAssembler.cpp:81): mass matrix with a density given at the mass rule's points, right-hand side with one at the
    stiffness rule's points (constant and pointwise force), positions of the mass rule's points, inertia mass."""
    mesh = MESHES[name]()
    lam, mu = oracle.lame(1e5, 0.3)
    _setup(eng, mesh, E.LINEAR_ELASTICITY, lam, mu)
    rho_fun = lambda p: 1000.0 * (1.0 + 0.5 * p[..., 0] + 0.25 * p[..., 1] * p[..., 2])  # noqa: E731
    mq = eng.mass_quadrature_points()
    mq_ref = oracle.mass_quadrature_points(mesh)
    assert mq.shape == mq_ref.shape and np.abs(mq - mq_ref).max() <= 1e-15 * max(np.abs(mq_ref).max(), 1.0) + 1e-16
    rho_m = rho_fun(mq)
    eng.assemble_mass(rho_m)
    rp, ci, v = eng.get_linear_csr()
    outer, inner, vals = oracle.assemble_mass(mesh, 3, rho_m)
    assert np.array_equal(rp, outer.astype(np.int64)) and np.array_equal(ci, inner)
    assert rel_err(v, vals) <= TOL
    # a constant given per point is the uniform density
    eng.assemble_mass(np.full(rho_m.shape, 2.5))
    assert rel_err(eng.get_linear_csr()[2], oracle.assemble_mass(mesh, 3, 2.5)[2]) <= TOL
    sq = eng.quadrature_points()
    rho_s = rho_fun(sq)
    force = np.array([0.3, -0.2, -9.8])
    assert rel_err(eng.assemble_rhs(force, rho_s), oracle.assemble_rhs(mesh, 3, force, rho_s)) <= TOL
    f_pts = np.stack([np.sin(sq[..., 0]), sq[..., 1] ** 2, -9.8 + sq[..., 2]], axis=-1)
    assert rel_err(eng.assemble_rhs_pointwise(f_pts, rho_s), oracle.assemble_rhs_pointwise(mesh, 3, f_pts, rho_s)) <= TOL
    # inertia term from the same mass: lumped mass = row sums
    eng.inertia_set_mass(rho_m)
    Mref = csc_to_scipy(oracle.assemble_mass(mesh, 3, rho_m), mesh.n_nodes * 3)
    assert rel_err(eng.lumped_mass(), np.asarray(Mref.sum(axis=1)).reshape(-1)) <= 1e-12
    # wrong shapes are rejected
    with pytest.raises(E.PfxError):
        eng.assemble_mass(rho_s if rho_s.shape[1] != rho_m.shape[1] else rho_m[:, :-1])


@pytest.mark.parametrize("name", ["p1_jitter", "p2_jitter_polyfem", "q1", "q2"])
def test_hessian_project_to_psd(eng, oracle, name):
    """assemble_hessian(project_to_psd = true): element Hessians projected onto the PSD cone (Assembler.cpp:554-555).
    The eigen-solvers differ (oracle: Jacobi sweeps; here and in the reference's Eigen: tridiagonalisation + QL / QR), the
    projection is the same: tolerance 1e-10 of the largest entry."""
    mesh = MESHES[name]()
    lam, mu = oracle.lame(2e4, 0.3)
    nnz = _setup(eng, mesh, E.NEOHOOKEAN, lam, mu)
    n = mesh.n_nodes * 3
    # a compression / stretch that makes many element Hessians indefinite
    x = np.zeros(n)
    x[0::3] = -0.35 * mesh.node_xyz[:, 0]
    x[1::3] = 0.3 * mesh.node_xyz[:, 1]
    x += M.seeded_displacement(mesh, 3, 2e-3)
    v = eng.hessian(x)
    vp = eng.hessian(x, project_to_psd=True)
    ref = oracle.assemble_hessian_psd(mesh, oracle.NEOHOOKEAN, lam, mu, x)[2]
    plain = oracle.assemble_hessian(mesh, oracle.NEOHOOKEAN, lam, mu, x)[2]
    assert vp.size == nnz == ref.size
    assert np.abs(plain - ref).max() > 1e-3 * np.abs(plain).max(), "test state is not indefinite anywhere"
    assert rel_err(v, plain) <= TOL
    assert np.abs(vp - ref).max() <= 1e-10 * np.abs(ref).max()
    # the projected matrix is positive semi-definite
    rp, ci = eng.get_pattern()
    import scipy.sparse as sp
    Hp = sp.csr_matrix((vp, ci, rp), shape=(n, n))
    for seed in range(3):
        z = np.random.default_rng(seed).standard_normal(n)
        assert z @ (Hp @ z) >= -1e-9 * np.abs(vp).max() * (z @ z)
    # at rest every element Hessian is PSD (its rigid modes have eigenvalue zero up to round-off): nothing to project
    zero = np.zeros(n)
    assert np.abs(eng.hessian(zero, project_to_psd=True) - eng.hessian(zero)).max() <= 1e-10 * np.abs(v).max()
    # LinearElasticity through the same interface: PSD by construction
    eng.set_material(E.LINEAR_ELASTICITY, lam, mu)
    assert np.abs(eng.hessian(x, project_to_psd=True) - eng.hessian(x)).max() <= 1e-10 * np.abs(v).max()


def test_hessian_pattern_reuse_is_deterministic(eng, oracle):
    """Newton steps reuse the pattern (SpareMatrixCache second pass == first pass) and repeated assemblies of
    the same state are bitwise identical (no atomics in the accumulation)."""
    mesh = M.cube_tets(3, 2, jitter=0.1)
    lam, mu = oracle.lame(2e4, 0.3)
    _setup(eng, mesh, E.NEOHOOKEAN, lam, mu)
    n = mesh.n_nodes * 3
    cache = oracle.Cache(n)
    for step in range(3):
        x = M.seeded_displacement(mesh, 3, 2e-3, seed=100 + step)
        ref = oracle.assemble_hessian_cached(mesh, oracle.NEOHOOKEAN, lam, mu, x, cache)
        v1 = eng.hessian(x)
        v2 = eng.hessian(x)
        assert np.array_equal(v1, v2)
        assert rel_err(v1, ref[2]) <= TOL
        g1, g2 = eng.gradient(x), eng.gradient(x)
        assert np.array_equal(g1, g2)


def test_invalid_state_reports_numeric_error(eng, oracle):
    mesh = M.cube_tets(2, 1)
    lam, mu = oracle.lame(2e4, 0.3)
    _setup(eng, mesh, E.NEOHOOKEAN, lam, mu)
    x = np.zeros(mesh.n_nodes * 3)
    x[0::3] = -5.0 * mesh.node_xyz[:, 0]  # collapses and inverts every element
    with pytest.raises(E.PfxError) as ei:
        eng.energy(x)
    assert ei.value.code == -4


def test_diag_shift_and_numeric_flag(eng, oracle):
    """pfx_set_diag_shift: SpMV and CG work with A + shift I on the free dofs (the regularised Hessian of the Newton
    solver) without touching the assembled values; pfx_take_numeric_flag reports and clears the inverted-element flag
    of the device-pointer evaluations."""
    import scipy.sparse as sp
    import torch
    mesh = M.cube_tets(3, 2, jitter=0.1)
    lam, mu = oracle.lame(2e4, 0.3)
    _setup(eng, mesh, E.NEOHOOKEAN, lam, mu)
    n = mesh.n_nodes * 3
    u = M.seeded_displacement(mesh, 3, 2e-3, seed=5)
    vals = eng.hessian(u).copy()
    bnd = M.boundary_dofs(mesh, 3, "x0")
    eng.set_dirichlet(bnd)
    free = np.ones(n, dtype=bool)
    free[bnd] = False
    x = np.random.default_rng(0).standard_normal(n)
    y0 = eng.spmv(x)
    shift = 37.5
    eng.set_diag_shift(shift)
    try:
        y1 = eng.spmv(x)
        assert np.abs(y1 - (y0 + shift * np.where(free, x, 0.0))).max() <= 1e-12 * np.abs(y0).max()
        assert np.array_equal(eng.get_values(), vals)
        b = np.random.default_rng(1).standard_normal(n)
        xs, its, err = eng.cg(b, None, E.PRECOND_JACOBI, 2000, 1e-12)
        rp, ci = eng.get_pattern()
        A = sp.csr_matrix((vals, ci, rp), shape=(n, n))
        r = (A @ xs + shift * xs - b)[free]
        assert np.linalg.norm(r) <= 1e-10 * np.linalg.norm(b[free]) and err < 1e-11
    finally:
        eng.set_diag_shift(0.0)
    assert np.array_equal(eng.spmv(x), y0)
    with pytest.raises(E.PfxError):
        eng.set_diag_shift(-1.0)
    eng.set_dirichlet(np.zeros(0, dtype=np.int32))
    # numeric flag
    assert not eng.take_numeric_flag()
    bad = np.zeros(n)
    bad[0::3] = -5.0 * mesh.node_xyz[:, 0]
    d_u = torch.from_numpy(bad).cuda()
    d_e = torch.zeros(1, dtype=torch.float64, device="cuda")
    eng.energy_dev(d_u.data_ptr(), d_e.data_ptr())
    eng.synchronize()
    assert not np.isfinite(d_e.item())
    assert eng.take_numeric_flag() and not eng.take_numeric_flag()
    assert abs(eng.energy(u) - oracle.assemble_energy(mesh, oracle.NEOHOOKEAN, lam, mu, u)) <= 1e-12 * abs(eng.energy(u))


@pytest.mark.parametrize("case", ["laplace_p1", "elastic_p2", "neohookean_p2"])
def test_reduced_matrix_and_cg(eng, oracle, case):
    if case == "laplace_p1":
        mesh, model, bs = M.cube_tets(6, 1), E.LAPLACIAN, 1
    elif case == "elastic_p2":
        mesh, model, bs = M.cube_tets(3, 2), E.LINEAR_ELASTICITY, 3
    else:
        mesh, model, bs = M.cube_tets(3, 2, jitter=0.1), E.NEOHOOKEAN, 3
    lam, mu = oracle.lame(1e5, 0.3)
    _setup(eng, mesh, model, lam, mu)
    n = mesh.n_nodes * bs
    if model == E.NEOHOOKEAN:
        x = M.seeded_displacement(mesh, 3, 1e-3)
        eng.hessian(x, fetch=False)
        full = oracle.assemble_hessian(mesh, oracle.NEOHOOKEAN, lam, mu, x)
    else:
        eng.assemble_linear()
        full = oracle.assemble_linear(mesh, model, lam, mu)
    bd = M.boundary_dofs(mesh, bs, "x0" if bs == 3 else "all")
    eng.set_dirichlet(bd)
    # full_to_reduced_matrix parity
    red = oracle.full_to_reduced_matrix(full, bd)
    rp, ci, v = eng.get_reduced_csr()
    assert np.array_equal(rp, red[0].astype(np.int64)) and np.array_equal(ci, red[1])
    assert rel_err(v, red[2]) <= TOL
    # CG on the reduced system == masked CG on the device
    keep = np.setdiff1d(np.arange(n), bd)
    rng = np.random.default_rng(5)
    b_full = rng.standard_normal(n)
    xr, it_ref, err_ref = oracle.cg(red, b_full[keep], jacobi=True, max_iter=1000, tol=1e-12)
    xg, it, err = eng.cg(b_full, None, E.PRECOND_JACOBI, 1000, 1e-12)
    assert abs(it - it_ref) <= 2, (it, it_ref)
    assert not xg[bd].any()
    A = csc_to_scipy(red, keep.size)
    res = np.linalg.norm(A @ xg[keep] - b_full[keep]) / np.linalg.norm(b_full[keep])
    assert res <= 1e-10
    assert np.linalg.norm(xg[keep] - xr) <= 1e-8 * np.linalg.norm(xr)
    # identity preconditioner too
    xr2, it_ref2, _ = oracle.cg(red, b_full[keep], jacobi=False, max_iter=2000, tol=1e-10)
    xg2, it2, _ = eng.cg(b_full, None, E.PRECOND_IDENTITY, 2000, 1e-10)
    assert abs(it2 - it_ref2) <= 2
    eng.set_dirichlet(np.zeros(0, dtype=np.int32))


def test_cg_edge_cases(eng, oracle):
    mesh = M.cube_tets(3, 1)
    _setup(eng, mesh, E.LAPLACIAN, 1.0, 1.0)
    eng.assemble_linear()
    bd = M.boundary_dofs(mesh, 1, "all")
    eng.set_dirichlet(bd)
    n = mesh.n_nodes
    # zero right-hand side -> x = 0, 0 iterations, error 0 (Eigen's early exit)
    x, it, err = eng.cg(np.zeros(n), np.ones(n))
    assert it == 0 and err == 0 and not x.any()
    # max_iter reached (random right-hand side: a symmetric one is an eigenvector and converges in one step)
    b = np.random.default_rng(11).standard_normal(n)
    x, it, err = eng.cg(b, None, E.PRECOND_JACOBI, 2, 1e-14)
    K = oracle.full_to_reduced_matrix(oracle.assemble_linear(mesh, oracle.LAPLACIAN, 1.0, 1.0), bd)
    keep = np.setdiff1d(np.arange(n), bd)
    xr, it_ref, err_ref = oracle.cg(K, b[keep], max_iter=2, tol=1e-14)
    assert it == it_ref == 2
    assert abs(err - err_ref) <= 1e-10 * err_ref
    assert np.abs(x[keep] - xr).max() <= 1e-12 * np.abs(xr).max()
    # initial guess already the solution -> 0 iterations
    xs, _, _ = eng.cg(b, None, E.PRECOND_JACOBI, 1000, 1e-13)
    x2, it2, _ = eng.cg(b, xs, E.PRECOND_JACOBI, 1000, 1e-10)
    assert it2 == 0 and np.array_equal(x2, xs)
    eng.set_dirichlet(np.zeros(0, dtype=np.int32))


def test_dirichlet_solve_matches_oracle(eng, oracle):
    mesh = M.cube_tets(5, 1)
    _setup(eng, mesh, E.LAPLACIAN, 1.0, 1.0)
    eng.assemble_linear()
    n = mesh.n_nodes
    bd = M.boundary_dofs(mesh, 1, "all")
    eng.set_dirichlet(bd)
    b = np.full(n, -1.0 / n)
    b[bd] = mesh.node_xyz[bd, 0] ** 2  # prescribed boundary values
    K = oracle.assemble_linear(mesh, oracle.LAPLACIAN, 1.0, 1.0)
    (o2, i2, v2), b2 = oracle.dirichlet_modify(K, b, bd)
    xr, it_ref, _ = oracle.cg((o2, i2, v2), b2, jacobi=True, tol=1e-12)
    x, it, err = eng.dirichlet_solve(b)
    assert abs(it - it_ref) <= 2
    assert np.abs(x - xr).max() <= 1e-9 * np.abs(xr).max()
    assert np.array_equal(x[bd], b[bd])
    eng.set_dirichlet(np.zeros(0, dtype=np.int32))


def test_spmv_matches_oracle(eng, oracle):
    mesh = M.cube_tets(3, 2, jitter=0.1)
    lam, mu = oracle.lame(1e5, 0.3)
    _setup(eng, mesh, E.LINEAR_ELASTICITY, lam, mu)
    eng.assemble_linear()
    K = oracle.assemble_linear(mesh, oracle.LINEAR_ELASTICITY, lam, mu)
    x = np.random.default_rng(0).standard_normal(mesh.n_nodes * 3)
    assert rel_err(eng.spmv(x), oracle.spmv(K, x)) <= TOL


@pytest.mark.parametrize("case", ["neohookean_p2", "neohookean_p1_large_load", "linear_elasticity_p2"])
def test_newton_solve_matches_oracle(eng, oracle, case):
    """NonlinearSolver::minimize + SparseNewtonDescentSolver + BacktrackingLineSearch on the device against the
    oracle's restatement: same number of Newton iterations and line-search evaluations, same minimiser."""
    if case == "neohookean_p2":
        mesh, model, load = M.cube_tets(3, 2, jitter=0.1), E.NEOHOOKEAN, -50.0
    elif case == "neohookean_p1_large_load":
        mesh, model, load = M.cube_tets(4, 1, jitter=0.1), E.NEOHOOKEAN, -2500.0
    else:
        mesh, model, load = M.cube_tets(2, 2), E.LINEAR_ELASTICITY, -50.0
    lam, mu = oracle.lame(2e4, 0.3)
    _setup(eng, mesh, model, lam, mu)
    n = mesh.n_nodes * 3
    bnd = M.boundary_dofs(mesh, 3, "x0")
    eng.set_dirichlet(bnd)
    rhs = np.zeros(n)
    rhs[2::3] = load / mesh.n_nodes
    x0 = np.zeros(n)
    x_ref, info_ref = oracle.newton_solve(mesh, model, lam, mu, rhs, bnd, x0, n_threads=4)
    x, info = eng.newton_solve(x0, rhs)
    eng.set_dirichlet(np.zeros(0, dtype=np.int32))
    assert info["status"] == info_ref["status"] and info["status"] in (2, 3)
    assert info["iterations"] == info_ref["iterations"] >= 2
    # the last steps of a converged run compare energies that differ in their last bits: one trial step more or less
    assert abs(info["line_search_iterations"] - info_ref["line_search_iterations"]) <= 1
    assert info["descent_strategy"] == info_ref["descent_strategy"] == 0
    # +-2 per linear solve; the last solve of a converged run has a right-hand side at round-off level, where the
    # count is decided by the last bits of the gradient: allow 3 % of the total as well
    assert abs(info["cg_iterations"] - info_ref["cg_iterations"]) <= max(2 * info["iterations"], 0.03 * info_ref["cg_iterations"])
    assert np.abs(x - x_ref).max() <= 1e-9 * np.abs(x_ref).max()
    assert abs(info["energy"] - info_ref["energy"]) <= 1e-9 * abs(info_ref["energy"])
    assert not x[bnd].any()
    # stationarity of the minimiser, checked with the oracle's gradient
    free = np.ones(n, dtype=bool)
    free[bnd] = False
    g = oracle.assemble_gradient(mesh, model, lam, mu, x) - rhs
    assert np.linalg.norm(g[free]) <= 1e-7 * np.linalg.norm(rhs)


def test_newton_with_backtracking_matches_oracle(eng, oracle):
    """A load large enough that full Newton steps invert elements / increase the energy: the line search has to halve
    steps (more than 3 objective evaluations per iteration)."""
    mesh = M.cube_tets(3, 1, jitter=0.1)
    lam, mu = oracle.lame(2e4, 0.3)
    _setup(eng, mesh, E.NEOHOOKEAN, lam, mu)
    n = mesh.n_nodes * 3
    bnd = M.boundary_dofs(mesh, 3, "x0")
    eng.set_dirichlet(bnd)
    rhs = np.zeros(n)
    rhs[0::3] = -2e4 / mesh.n_nodes
    x_ref, info_ref = oracle.newton_solve(mesh, oracle.NEOHOOKEAN, lam, mu, rhs, bnd, np.zeros(n), n_threads=4)
    x, info = eng.newton_solve(np.zeros(n), rhs)
    eng.set_dirichlet(np.zeros(0, dtype=np.int32))
    assert info_ref["line_search_iterations"] > 3 * info_ref["iterations"]
    assert info["iterations"] == info_ref["iterations"] and info["status"] == info_ref["status"]
    # the run passes through the projected-Newton fallback; near convergence "energy went up" is decided by the last
    # bits of two energies, so the number of halvings may differ by a few evaluations
    assert abs(info["line_search_iterations"] - info_ref["line_search_iterations"]) <= 0.1 * info_ref["line_search_iterations"]
    assert info["line_search_iterations"] > 3 * info["iterations"]
    assert np.abs(x - x_ref).max() <= 1e-7 * np.abs(x_ref).max()


def test_newton_not_started_and_iteration_limit(eng, oracle):
    mesh = M.cube_tets(2, 2, jitter=0.1)
    lam, mu = oracle.lame(2e4, 0.3)
    _setup(eng, mesh, E.NEOHOOKEAN, lam, mu)
    n = mesh.n_nodes * 3
    bnd = M.boundary_dofs(mesh, 3, "x0")
    eng.set_dirichlet(bnd)
    x, info = eng.newton_solve(np.zeros(n), None)  # rest state, no load: gradient is zero
    assert info["status"] == 4 and info["iterations"] == 0 and not x.any()
    rhs = np.zeros(n)
    rhs[2::3] = -200.0 / mesh.n_nodes
    x_ref, info_ref = oracle.newton_solve(mesh, oracle.NEOHOOKEAN, lam, mu, rhs, bnd, np.zeros(n), max_iterations=1, f_delta=0.0,
                                          grad_norm=1e-14)
    assert info_ref["status"] == 1
    with pytest.raises(E.PfxError) as ei:  # the reference throws "Reached iteration limit"
        eng.newton_solve(np.zeros(n), rhs, max_iterations=1, f_delta=0.0, grad_norm=1e-14)
    assert ei.value.code == -4 and "iteration limit" in str(ei.value)
    eng.set_dirichlet(np.zeros(0, dtype=np.int32))


@pytest.mark.parametrize("name,model", [("p2_jitter_polyfem", "NeoHookean"), ("p1_jitter", "NeoHookean"),
                                        ("q2", "NeoHookean"), ("p2_jitter_polyfem", "LinearElasticity")])
def test_implicit_euler_steps_match_oracle(eng, oracle, name, model):
    """Three implicit Euler steps of the transient solve (inertia term from the device mass matrix, dt^2-weighted
    elastic and body forms, Newton + backtracking per step) against the oracle's restatement: same Newton decisions,
    positions / velocities / accelerations to solver accuracy.  (No Q1 case: with the reference's one-point rule for Q1
    hexes the clamped static problem of the last check is singular -- hourglass modes.)"""
    mesh = MESHES[name]()
    lam, mu = oracle.lame(2e4, 0.3)
    mdl = E.NEOHOOKEAN if model == "NeoHookean" else E.LINEAR_ELASTICITY
    omdl = oracle.NEOHOOKEAN if model == "NeoHookean" else oracle.LINEAR_ELASTICITY
    _setup(eng, mesh, mdl, lam, mu)
    n = mesh.n_nodes * 3
    bnd = M.boundary_dofs(mesh, 3, "x0")
    eng.set_dirichlet(bnd)
    rho, dt = 1000.0, 0.02
    half = np.arange(mesh.n_el) < mesh.n_el // 2
    rho_e = np.where(half, rho, 2.5 * rho)
    mass = oracle.assemble_mass(mesh, 3, rho_e)
    force = np.array([0.0, 3.0, -9.81])
    rhs = oracle.assemble_rhs(mesh, 3, force, rho_e)
    assert rel_err(eng.assemble_rhs(force, rho_e), rhs) <= TOL
    eng.inertia_set_mass(rho_e)
    free = np.ones(n, dtype=bool)
    free[bnd] = False
    x = np.zeros(n)
    v = np.where(free, 0.05 * np.random.default_rng(9).standard_normal(n), 0.0)
    xo, vo = x.copy(), v.copy()
    try:
        for step in range(3):
            x, v, a, info = eng.implicit_euler_step(dt, x, v, rhs)
            xo, vo, ao, iref = oracle.implicit_euler_step(mesh, omdl, lam, mu, rhs, bnd, mass, dt, xo, vo)
            assert info["status"] == iref["status"] and info["status"] in (2, 3)
            assert info["iterations"] == iref["iterations"] >= 1
            assert info["line_search_iterations"] == iref["line_search_iterations"]
            assert abs(info["cg_iterations"] - iref["cg_iterations"]) <= max(3 * info["iterations"], 0.03 * iref["cg_iterations"])
            assert abs(info["energy"] - iref["energy"]) <= 1e-9 * abs(iref["energy"])
            assert np.abs(x - xo).max() <= 1e-8 * np.abs(xo).max()
            assert np.abs(v - vo).max() <= 1e-7 * np.abs(vo).max()
            assert np.abs(a - ao).max() <= 1e-6 * np.abs(ao).max()
            assert not x[bnd].any()
        # the static objective is back after the step, and the enable / disable switch by hand gives the same objective
        import torch
        # (a fraction of the load: the full weight of this soft body is a large-deformation problem of its own)
        xs, infs = eng.newton_solve(np.zeros(n), 0.02 * rhs)
        xr, ir = oracle.newton_solve(mesh, omdl, lam, mu, 0.02 * rhs, bnd, np.zeros(n))
        assert infs["iterations"] == ir["iterations"] and np.abs(xs - xr).max() <= 1e-8 * np.abs(xr).max()
        xt = torch.from_numpy(xo + dt * vo).cuda()
        eng.inertia_enable(dt * dt, xt.data_ptr())
        x4, i4 = eng.newton_solve(xo, rhs)
        eng.inertia_disable()
        x4o, i4o = oracle.newton_solve(mesh, omdl, lam, mu, rhs, bnd, xo, mass=mass, x_tilde=xo + dt * vo, weight=dt * dt)
        assert i4["iterations"] == i4o["iterations"] and np.abs(x4 - x4o).max() <= 1e-8 * np.abs(x4o).max()
    finally:
        eng.inertia_disable()
        eng.set_dirichlet(np.zeros(0, dtype=np.int32))


@pytest.mark.parametrize("name", ["p2_jitter_polyfem", "q2"])
def test_inertia_form_pieces(eng, oracle, name):
    """The pieces of InertiaForm a caller-driven Newton loop uses (the reference checks the form by finite differences,
    tests/test_form_derivatives.cpp:290-305; its value is quadratic, so the pieces are compared directly): M v against
    the oracle's mass matrix, the value 1/2 d^T M d, and weight * H + M."""
    import scipy.sparse as sp
    import torch
    mesh = MESHES[name]()
    lam, mu = oracle.lame(2e4, 0.3)
    _setup(eng, mesh, E.NEOHOOKEAN, lam, mu)
    n = mesh.n_nodes * 3
    rho_e = 500.0 + 100.0 * (np.arange(mesh.n_el) % 4)
    mo = oracle.assemble_mass(mesh, 3, rho_e)
    Mo = sp.csc_matrix((mo[2], mo[1], mo[0]), shape=(n, n)).tocsr()
    eng.inertia_set_mass(rho_e)
    v = np.random.default_rng(3).standard_normal(n)
    d_v = torch.from_numpy(v).cuda()
    d_y = torch.zeros_like(d_v)
    eng.mass_apply_dev(d_v.data_ptr(), d_y.data_ptr())
    eng.synchronize()
    y = d_y.cpu().numpy()
    assert rel_err(y, Mo @ v) <= TOL
    assert abs(0.5 * v @ y - 0.5 * v @ (Mo @ v)) <= TOL * abs(v @ y)
    u = M.seeded_displacement(mesh, 3, 2e-3, seed=8)
    h = eng.hessian(u).copy()
    eng.hessian_add_mass_dev(2.5e-4)
    hm = eng.get_values()
    rp, ci = eng.get_pattern()
    H = sp.csr_matrix((h, ci, rp), shape=(n, n))
    Hm = sp.csr_matrix((hm, ci, rp), shape=(n, n))
    ref = 2.5e-4 * H + Mo
    assert abs(Hm - ref).max() <= TOL * abs(ref).max()


def test_augmented_lagrangian_dirichlet(eng, oracle):
    """ALForm / ALSolver (solver/forms/ALForm.cpp:7-86, solver/ALSolver.cpp:22-108): the penalised full-size solve against
    the oracle (for the static problem the penalty w/2 (x-t)^T D (x-t) is the oracle's inertia term with mass w D,
    predictor t and weight 1), the lumped mass, and the solver loop."""
    import scipy.sparse as sp
    mesh = MESHES["p2_jitter_polyfem"]()
    n = mesh.n_nodes * 3
    lam, mu = oracle.lame(2e4, 0.3)
    _setup(eng, mesh, E.NEOHOOKEAN, lam, mu)
    bnd = M.boundary_dofs(mesh, 3, "x0")
    target = np.zeros(n)
    target[bnd[0::3]] = 0.02  # the clamped face is pushed along x
    rhs = np.zeros(n)
    rhs[2::3] = -50.0 / mesh.n_nodes
    # lumped mass = row sums of the mass matrix
    eng.inertia_set_mass(1000.0)
    mo = csc_to_scipy(oracle.assemble_mass(mesh, 3, 1000.0), n)
    lumped = eng.lumped_mass()
    assert rel_err(lumped, np.asarray(mo.sum(axis=1)).ravel()) <= 1e-13
    eng.set_material(E.NEOHOOKEAN, lam, mu)
    # one penalised solve, identity lumped mass (static problem), no dof held
    w = 1e6
    masked = np.zeros(n)
    masked[bnd] = 1.0
    eng.set_dirichlet(np.zeros(0, dtype=np.int32))
    eng.al_enable_host(w, masked, target)
    xs, info = eng.newton_solve(np.zeros(n), rhs)
    eng.al_disable()
    d = sp.diags(w * masked).tocsc()
    d.sort_indices()
    xr, iref = oracle.newton_solve(mesh, oracle.NEOHOOKEAN, lam, mu, rhs, np.zeros(0, dtype=np.int32), np.zeros(n),
                                   mass=(d.indptr.astype(np.int32), d.indices.astype(np.int32), d.data), x_tilde=target, weight=1.0)
    assert info["status"] == iref["status"] and info["iterations"] == iref["iterations"] >= 2
    assert np.abs(xs - xr).max() <= 1e-8 * np.abs(xr).max()
    assert np.abs(xs[bnd] - target[bnd]).max() <= 1e-3 * 0.02  # the penalty holds the face near its target
    # the loop: forced penalty pass, then the solve with the dofs held; equals the direct constrained solve
    xa, ia, used = eng.al_solve(np.zeros(n), rhs, bnd, target[bnd], force_al=True)
    assert used == [1e6] and np.array_equal(xa[bnd], target[bnd])
    eng.set_dirichlet(bnd)
    xd, _ = eng.newton_solve(target.copy(), rhs)
    assert np.abs(xa - xd).max() <= 1e-7 * np.abs(xd).max()
    # Dirichlet values that invert the elements next to the face when imposed at once: the penalty passes run unforced
    big = np.zeros(n)
    big[bnd[0::3]] = 0.45
    xb, ib, used_b = eng.al_solve(np.zeros(n), None, bnd, big[bnd], initial_weight=1e3)
    assert len(used_b) >= 1 and np.array_equal(xb[bnd], big[bnd]) and np.isfinite(eng.energy(xb))
    assert ib["status"] in (2, 3, 4)
    # the weight limit: the reference's ALSolver throws "Unable to solve AL problem" (solver/ALSolver.cpp:63-69)
    with pytest.raises(E.PfxError) as ex:
        eng.al_solve(np.zeros(n), rhs, bnd, target[bnd], force_al=True, initial_weight=1e6, max_weight=1.5e6)
    assert ex.value.code == E.PFX_ERR_NUMERIC and "Unable to solve AL problem" in str(ex.value)
    eng.set_dirichlet(np.zeros(0, dtype=np.int32))
    # with the term enabled but weight 0 nothing changes; null pointers disable
    eng.al_enable_host(0.0, masked, target)
    with pytest.raises(E.PfxError):
        eng.al_enable_host(1.0, -masked, target)


def test_inertia_error_behaviour(eng, oracle):
    mesh = M.cube_tets(2, 1)
    lam, mu = oracle.lame(2e4, 0.3)
    _setup(eng, mesh, E.NEOHOOKEAN, lam, mu)
    n = mesh.n_nodes * 3
    with pytest.raises(E.PfxError):  # no mass matrix captured since the pattern was built
        eng.implicit_euler_step(0.1, np.zeros(n), np.zeros(n))
    eng.inertia_set_mass(1.0)
    with pytest.raises(E.PfxError):
        eng.implicit_euler_step(0.0, np.zeros(n), np.zeros(n))
    _setup(eng, mesh, E.LAPLACIAN, 1.0, 1.0)
    with pytest.raises(E.PfxError):
        eng.inertia_set_mass(1.0)


def test_host_mirror_of_reference_interface(oracle):
    """Reads like the reference's tests/test_assembler.cpp:12-64 ("hessian_lin"): stiffness from the linear
    Assembler == what the NL interface returns, through the reference-shaped host classes."""
    mesh = M.cube_tets(2, 2, jitter=0.1)
    asm = E.Assembler("LinearElasticity")
    asm.set_size(3)
    asm.add_multimaterial(0, {"E": 1e5, "nu": 0.3})
    K = asm.assemble(True, mesh.n_nodes, mesh, mesh, None)
    lam, mu = oracle.lame(1e5, 0.3)
    ref = oracle.assemble_linear(mesh, oracle.LINEAR_ELASTICITY, lam, mu)
    assert K.rows() == mesh.n_nodes * 3 and K.non_zeros() == ref[1].size
    assert np.array_equal(K.inner, ref[1]) and rel_err(K.values, ref[2]) <= TOL
    nl = E.NLAssembler("NeoHookean", engine=asm.engine)
    nl.add_multimaterial(0, {"E": 2e4, "nu": 0.3})
    cache = E.SpareMatrixCache()
    x = M.seeded_displacement(mesh, 3, 1e-3).reshape(-1, 1)
    lam, mu = oracle.lame(2e4, 0.3)
    H = nl.assemble_hessian(True, mesh.n_nodes, False, mesh, mesh, None, 0.0, x, x, cache)
    g = nl.assemble_grad(True, mesh.n_nodes, mesh, mesh, None, 0.0, x, x)
    e = nl.assemble(True, mesh, mesh, None, 0.0, x, x)
    href = oracle.assemble_hessian(mesh, oracle.NEOHOOKEAN, lam, mu, x.reshape(-1))
    assert np.array_equal(H.inner, href[1]) and rel_err(H.values, href[2]) <= TOL
    assert rel_err(g.reshape(-1), oracle.assemble_gradient(mesh, oracle.NEOHOOKEAN, lam, mu, x.reshape(-1))) <= TOL
    assert abs(e - oracle.assemble_energy(mesh, oracle.NEOHOOKEAN, lam, mu, x.reshape(-1))) <= TOL * abs(e)
    solver = E.LinearSolver.create("Eigen::ConjugateGradient", "Eigen::DiagonalPreconditioner", asm.engine)
    solver.set_parameters({"Eigen::ConjugateGradient": {"max_iter": 500, "tolerance": 1e-10}})
    bd = M.boundary_dofs(mesh, 3, "x0")
    asm.engine.set_dirichlet(bd)
    dx = np.zeros(mesh.n_nodes * 3)
    solver.analyze_pattern(H, H.rows())
    solver.factorize(H)
    solver.solve(-g.reshape(-1), dx)
    info = solver.get_info()
    assert info["solver_error"] < 1e-10 and 0 < info["solver_iter"] <= 500
    asm.engine.close()


def test_cpp_shim_with_reference_signatures():
    """include/polyfem_amd/assembler.hpp: Assembler<L>::assemble, NLAssembler<L>::assemble/_grad/_hessian and the
    polysolve-shaped LinearSolver, driven from C++ (tests/cpp/test_shim.cpp, built by __graft_entry__.build())."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "test_shim")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "shim ok" in out.stdout, out.stdout + out.stderr


def test_error_behaviour_of_the_abi(oracle):
    """Status codes instead of a CPU fallback: wrong call order, unsupported inputs and invalid arguments are reported
    through pfx_last_error, and the context stays usable afterwards."""
    e2 = E.Engine(0)
    try:
        mesh = M.cube_tets(2, 2)
        n = mesh.n_nodes * 3
        with pytest.raises(E.PfxError) as ei:  # nothing bound yet
            e2.build_pattern()
        assert ei.value.code == -1
        e2.set_mesh(mesh)
        with pytest.raises(E.PfxError) as ei:  # dim = 2 belongs to triangle meshes
            e2.set_material(E.LINEAR_ELASTICITY, 1.0, 1.0, dim=2)
        assert ei.value.code == -1 and "dim" in str(ei.value)
        with pytest.raises(E.PfxError) as ei:  # there is no 4-D
            e2.set_material(E.LINEAR_ELASTICITY, 1.0, 1.0, dim=4)
        assert ei.value.code == -3
        with pytest.raises(E.PfxError) as ei:  # one value per element required
            e2.set_material(E.NEOHOOKEAN, np.ones(mesh.n_el - 1), np.ones(mesh.n_el - 1))
        assert ei.value.code == -1
        e2.set_material(E.NEOHOOKEAN, 1e4, 1e4)
        with pytest.raises(E.PfxError) as ei:  # pattern missing
            e2.energy(np.zeros(n))
        assert ei.value.code == -1 and "pfx_build_pattern" in str(ei.value)
        e2.build_pattern()
        with pytest.raises(E.PfxError) as ei:  # the linear assembler needs a linear model
            e2.assemble_linear()
        assert ei.value.code == -1
        with pytest.raises(E.PfxError) as ei:  # Dirichlet dofs must be sorted, unique and in range
            e2.set_dirichlet(np.array([5, 3], dtype=np.int32))
        assert ei.value.code == -1
        with pytest.raises(E.PfxError) as ei:
            e2.set_dirichlet(np.array([n], dtype=np.int32))
        assert ei.value.code == -1
        e2.set_material(E.LAPLACIAN)
        e2.build_pattern()
        with pytest.raises(E.PfxError) as ei:  # no energy for the Laplacian
            e2.energy(np.zeros(mesh.n_nodes))
        assert ei.value.code == -3
        # still usable
        e2.set_material(E.NEOHOOKEAN, 1e4, 1e4)
        e2.build_pattern()
        assert e2.energy(np.zeros(n)) == 0.0
        bad = M.cube_tets(2, 2)
        bad.conn = bad.conn.copy()
        bad.conn[0, 0] = bad.n_nodes  # out-of-range basis index
        with pytest.raises(E.PfxError) as ei:
            e2.set_mesh(bad)
        assert ei.value.code == -1
    finally:
        e2.close()


def test_isolated_nodes_and_single_element(eng, oracle):
    """Ragged input: basis indices that no element uses (empty rows / columns) and a mesh of a single element."""
    import copy
    base = M.cube_tets(1, 2)
    mesh = copy.copy(base)
    mesh.n_nodes = base.n_nodes + 3  # three unused trailing basis indices
    lam, mu = oracle.lame(1e5, 0.3)
    _setup(eng, mesh, E.LINEAR_ELASTICITY, lam, mu)
    eng.assemble_linear()
    rp, ci, v = eng.get_linear_csr()
    outer, inner, vals = oracle.assemble_linear(mesh, oracle.LINEAR_ELASTICITY, lam, mu)
    assert np.array_equal(rp, outer.astype(np.int64)) and np.array_equal(ci, inner) and rel_err(v, vals) <= TOL
    n = mesh.n_nodes * 3
    assert rp.size == n + 1 and np.all(np.diff(rp)[-9:] == 0)
    x = np.random.default_rng(0).standard_normal(n)
    y = eng.spmv(x)
    assert not y[-9:].any() and rel_err(y, oracle.spmv((outer, inner, vals), x)) <= TOL
    # CG: the empty rows have a zero diagonal (Jacobi uses 1 there) and a zero right-hand side
    bnd = np.concatenate([M.boundary_dofs(base, 3, "x0"), np.arange(n - 9, n)]).astype(np.int32)
    eng.set_dirichlet(bnd)
    b = np.random.default_rng(1).standard_normal(n)
    xs, it, err = eng.cg(b, None, E.PRECOND_JACOBI, 500, 1e-12)
    eng.set_dirichlet(np.zeros(0, dtype=np.int32))
    assert err <= 1e-10 and not xs[bnd].any()
    one = copy.copy(base)
    one.conn, one.gconn = base.conn[:1].copy(), base.gconn[:1].copy()
    _setup(eng, one, E.NEOHOOKEAN, lam, mu)
    u = np.random.default_rng(3).standard_normal(one.n_nodes * 3) * 1e-3
    assert rel_err(eng.hessian(u), oracle.assemble_hessian(one, oracle.NEOHOOKEAN, lam, mu, u)[2]) <= TOL
    assert rel_err(eng.gradient(u), oracle.assemble_gradient(one, oracle.NEOHOOKEAN, lam, mu, u)) <= TOL


@pytest.mark.parametrize("name", ["p1_jitter", "p2_jitter_polyfem", "q1", "q2"])
def test_space_dependent_body_force(eng, oracle, name):
    """RhsAssembler::assemble with problem.rhs() evaluated per quadrature point (assembler/RhsAssembler.cpp:55-98)."""
    mesh = MESHES[name]()
    _setup(eng, mesh, E.LINEAR_ELASTICITY, 1.0, 1.0)
    pts = eng.quadrature_points()
    ref_pts = oracle.quadrature_points(mesh)
    assert pts.shape == ref_pts.shape and np.abs(pts - ref_pts).max() <= 1e-15
    f = np.stack([np.sin(3 * pts[..., 0]) * pts[..., 1], pts[..., 2] ** 2 - 0.3, np.exp(-pts[..., 0]) + pts[..., 1] * pts[..., 2]], axis=-1)
    rho_e = 900.0 + 200.0 * (np.arange(mesh.n_el) % 3)
    for rho in (1000.0, rho_e):
        got = eng.assemble_rhs_pointwise(f, rho)
        assert rel_err(got, oracle.assemble_rhs_pointwise(mesh, 3, f, rho)) <= 1e-14
    const = np.array([0.0, -9.81, 0.5])
    assert np.array_equal(eng.assemble_rhs_pointwise(np.broadcast_to(const, pts.shape).copy(), 1000.0), eng.assemble_rhs(const, 1000.0))
    with pytest.raises(ValueError):
        eng.assemble_rhs_pointwise(f[:, :-1] if f.shape[1] > 1 else f[:-1], 1.0)


@pytest.mark.parametrize("name", ["p1_jitter", "p2_jitter_polyfem", "q1", "q2"])
def test_space_dependent_materials(eng, oracle, name):
    """LameParameters::lambda_mu evaluated at every quadrature point (utils/ElasticityUtils.cpp:490-517): linear
    stiffness (bit-exact pattern), energy, gradient, Hessian for the three elastic models."""
    mesh = MESHES[name]()
    eng.set_mesh(mesh)
    eng.set_material(E.LINEAR_ELASTICITY, 1.0, 1.0)
    pts = eng.quadrature_points()
    young = 1e5 * (1.0 + 0.5 * pts[..., 0] + 0.25 * np.sin(4 * pts[..., 1]))
    nu = 0.25 + 0.1 * pts[..., 2]
    lam = young * nu / ((1 + nu) * (1 - 2 * nu))
    mu = young / (2 * (1 + nu))
    eng.set_material(E.LINEAR_ELASTICITY, lam, mu)
    eng.build_pattern()
    eng.assemble_linear()
    rp, ci, v = eng.get_linear_csr()
    outer, inner, vals = oracle.assemble_linear(mesh, oracle.LINEAR_ELASTICITY, lam, mu)
    assert np.array_equal(rp, outer.astype(np.int64)) and np.array_equal(ci, inner)
    assert rel_err(v, vals) <= TOL
    h = 1.0 / round((mesh.n_el / (6 if mesh.kind < 2 else 1)) ** (1 / 3))
    x = M.seeded_displacement(mesh, 3, 2e-2 * h, seed=11)
    for mid in (E.NEOHOOKEAN, E.SAINT_VENANT, E.LINEAR_ELASTICITY):
        eng.set_material(mid, lam, mu)
        e_ref = oracle.assemble_energy(mesh, mid, lam, mu, x)
        assert abs(eng.energy(x) - e_ref) <= TOL * abs(e_ref)
        assert rel_err(eng.gradient(x), oracle.assemble_gradient(mesh, mid, lam, mu, x)) <= TOL
        assert rel_err(eng.hessian(x), oracle.assemble_hessian(mesh, mid, lam, mu, x)[2]) <= TOL
    # per-point arrays that are constant inside every element reproduce the per-element interface
    le = lam[:, 0].copy()
    me = mu[:, 0].copy()
    eng.set_material(E.NEOHOOKEAN, np.repeat(le[:, None], lam.shape[1], axis=1), np.repeat(me[:, None], lam.shape[1], axis=1))
    g_pp = eng.gradient(x)
    eng.set_material(E.NEOHOOKEAN, le, me)
    assert np.array_equal(g_pp, eng.gradient(x))
    with pytest.raises(E.PfxError):
        eng.set_material(E.NEOHOOKEAN, lam[:, :-1] if lam.shape[1] > 1 else lam[:-1], mu[:, :-1] if lam.shape[1] > 1 else mu[:-1])
    eng.set_material(E.LINEAR_ELASTICITY, 1.0, 1.0)


def test_no_device_memory_drift(oracle):
    """Engines created, driven through every kind of call and closed again leave the free device memory unchanged."""
    import torch
    mesh = M.cube_tets(6, 2)
    n = mesh.n_nodes * 3
    bnd = M.boundary_dofs(mesh, 3, "x0")
    rhs = np.zeros(n)
    rhs[2::3] = -1.0 / mesh.n_nodes
    free = []
    for rep in range(4):
        e = E.Engine(0)
        e.set_mesh(mesh)
        e.set_material(E.LINEAR_ELASTICITY, 5e4, 4e4)
        e.build_pattern()
        e.assemble_linear()
        e.get_linear_csr()
        e.set_elasticity_tensor(E.elasticity_tensor(lam=5e4, mu=4e4))
        e.assemble_linear()
        e.set_material(E.NEOHOOKEAN, 1e4, 8e3)
        e.set_element_models([E.NEOHOOKEAN, E.SAINT_VENANT] * (mesh.n_el // 2))
        e.set_dirichlet(bnd)
        e.newton_solve(np.zeros(n), rhs)
        e.hessian(np.zeros(n), project_to_psd=True)
        e.inertia_set_mass(1000.0)
        e.implicit_euler_step(1e-3, np.zeros(n), np.zeros(n), rhs)
        e.al_solve(np.zeros(n), rhs, bnd, np.zeros(bnd.size), force_al=True)
        e.set_mesh(M.cube_hexes(3, 2))
        e.set_material(E.LAPLACIAN, 1, 1)
        e.build_pattern()
        e.assemble_linear()
        e.close()
        free.append(torch.cuda.mem_get_info()[0])
    assert free[-1] == free[1], free


def test_neohookean_hip_against_sympy_derivatives(eng):
    """The HIP energy / gradient / Hessian of one element against the symbolic derivatives of the energy density
    (tests/golden/sympy/neohookean_sympy.npz): independent of the oracle."""
    import os
    from tests.test_oracle_kat import _single_tet_with_gradient, neohookean_sympy_expected
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "sympy", "neohookean_sympy.npz"))
    lam, mu = float(g["lam"]), float(g["mu"])
    for F, psi, dpsi, d2psi in zip(g["F"], g["psi"], g["dpsi"], g["d2psi"]):
        mesh, u = _single_tet_with_gradient(F)
        _setup(eng, mesh, E.NEOHOOKEAN, lam, mu)
        assert abs(eng.energy(u) - psi / 6.0) <= 1e-12 * max(abs(psi / 6.0), 1e-3 * mu)
        grad_ref, hess_ref = neohookean_sympy_expected(F, dpsi, d2psi)
        assert np.abs(eng.gradient(u) - grad_ref).max() <= 1e-12 * max(np.abs(grad_ref).max(), mu)
        rp, ci = eng.get_pattern()
        H = csc_to_scipy((rp, ci, eng.hessian(u)), 12).toarray()
        assert np.abs(H - hess_ref).max() <= 1e-12 * np.abs(hess_ref).max()


@pytest.mark.parametrize("name", ["p1_jitter", "p2_jitter_polyfem", "q1", "q2"])
def test_neohookean_hessian_at_large_deformations(eng, oracle, name):
    """Stretch, shear and bending of up to 30 % plus noise: the row functor builds the NeoHookean block from products the
    element kernel prepares (J^-1 J^-T, cof(F) J^-T, c2 / det F), whose algebra only shows at deformations far from the
    identity.  Plain and projected Hessian, gradient and energy against the oracle."""
    mesh = MESHES[name]()
    lam, mu = oracle.lame(2e4, 0.4)
    _setup(eng, mesh, E.NEOHOOKEAN, lam, mu)
    pos = M.node_positions(mesh)
    for amp in (0.05, 0.15, 0.3):
        u = np.zeros((mesh.n_nodes, 3))
        u[:, 0] = amp * pos[:, 0] + 0.5 * amp * pos[:, 1] ** 2
        u[:, 1] = -0.3 * amp * pos[:, 1] + amp * np.sin(2 * pos[:, 0])
        u[:, 2] = 0.7 * amp * pos[:, 2] * pos[:, 0]
        u += np.random.default_rng(3).standard_normal(u.shape) * amp * 0.02
        x = u.reshape(-1)
        ho = oracle.assemble_hessian(mesh, oracle.NEOHOOKEAN, lam, mu, x)
        assert rel_err(eng.hessian(x), ho[2]) <= TOL
        assert rel_err(eng.gradient(x), oracle.assemble_gradient(mesh, oracle.NEOHOOKEAN, lam, mu, x)) <= TOL
        e_ref = oracle.assemble_energy(mesh, oracle.NEOHOOKEAN, lam, mu, x)
        assert abs(eng.energy(x) - e_ref) <= TOL * abs(e_ref)
        hp = oracle.assemble_hessian_psd(mesh, oracle.NEOHOOKEAN, lam, mu, x)
        assert np.abs(eng.hessian(x, project_to_psd=True) - hp[2]).max() <= 1e-10 * np.abs(hp[2]).max()
