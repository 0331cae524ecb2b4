"""GPU tests at the full benchmark sizes (1 M P2 tets, 1 M Q2 hexes, 105 k P1 tets), through size-independent properties (the oracle cannot run these sizes in
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


def test_config1_laplacian_p1_full_size_parity(oracle):
    """Config 1 (Laplacian, P1, 26^3 Kuhn grid = 105 456 tets) is small enough for the oracle: pattern bit-exact,
    values, and the Dirichlet CG solve (x = 0 face, rhs = 1) against the oracle's CG on the reduced matrix."""
    mesh = M.cube_tets(26, 1)
    assert mesh.n_el == 105456 and mesh.n_nodes == 19683
    eng = E.Engine(0)
    try:
        eng.set_mesh(mesh)
        eng.set_material(E.LAPLACIAN)
        eng.build_pattern()
        eng.assemble_linear()
        rp, ci, v = eng.get_linear_csr()
        outer, inner, vals = oracle.assemble_linear(mesh, oracle.LAPLACIAN, 1.0, 1.0, n_threads=8)
        assert np.array_equal(rp, outer.astype(np.int64)) and np.array_equal(ci, inner)
        assert np.abs(v - vals).max() <= 1e-12 * np.abs(vals).max()
        bnd = M.boundary_dofs(mesh, 1, "x0")
        eng.set_dirichlet(bnd)
        n = mesh.n_nodes
        b = np.ones(n)
        x, it, err = eng.cg(b, None, E.PRECOND_JACOBI, 1000, 1e-12)
        red = oracle.full_to_reduced_matrix((outer, inner, vals), bnd)
        free = np.ones(n, dtype=bool)
        free[bnd] = False
        xr, it_ref, err_ref = oracle.cg(red, b[free], jacobi=True, max_iter=1000, tol=1e-12)
        assert abs(it - it_ref) <= 2 and err <= 1e-10 and err_ref <= 1e-10
        assert np.abs(x[free] - xr).max() <= 1e-9 * np.abs(xr).max() and not x[bnd].any()
    finally:
        eng.close()


def test_config4_q2_hex_full_size_properties():
    """Config 4 (LinearElasticity, Q2 hexes, structured 100^3 grid: 1 M elements, 24.4 M dofs, 4.6e9 non-zeros) through
    size-independent properties of the operator; the matrix itself (37 GB) stays on the device."""
    n_side = 100
    mesh = M.cube_hexes(n_side, 2)
    assert mesh.n_el == n_side ** 3 and mesh.n_nodes == (2 * n_side + 1) ** 3
    eng = E.Engine(0)
    try:
        eng.set_mesh(mesh)
        lam, mu = 1e5 * 0.3 / (1.3 * 0.4), 1e5 / 2.6
        eng.set_material(E.LINEAR_ELASTICITY, lam, mu)
        nnz = eng.build_pattern()
        # node couplings of a tensor grid: vertex / edge / face / cell nodes see 5 / 3 nodes per direction, clipped at the boundary
        per_dir = np.where(np.arange(2 * n_side + 1) % 2 == 0, 5, 3).astype(np.int64)
        per_dir[0] = per_dir[-1] = 3
        assert nnz == 9 * int(per_dir.sum()) ** 3
        eng.assemble_linear()
        n = mesh.n_nodes * 3
        rng = np.random.default_rng(4)
        x, y = rng.standard_normal(n), rng.standard_normal(n)
        Ax, Ay = eng.spmv(x), eng.spmv(y)
        scale = np.abs(Ax).max()
        assert abs(y @ Ax - x @ Ay) <= 1e-11 * abs(y @ Ax)
        assert np.abs(eng.spmv(2.0 * x - 3.0 * y) - (2.0 * Ax - 3.0 * Ay)).max() <= 1e-11 * scale
        for c in range(3):
            t = np.zeros(n)
            t[c::3] = 1.0
            assert np.abs(eng.spmv(t)).max() <= 1e-10 * scale
        rot = np.zeros(n)
        rot[0::3], rot[1::3] = -mesh.node_xyz[:, 1], mesh.node_xyz[:, 0]
        assert np.abs(eng.spmv(rot)).max() <= 1e-9 * scale
        # x^T K x = 2 * energy, with the energy from the element kernel of the nonlinear interface
        assert abs(x @ Ax - 2.0 * eng.energy(x)) <= 1e-10 * abs(x @ Ax)
        # a uniform stretch u = (a X, 0, 0): energy = (lambda/2 + mu) a^2 * volume
        a = 1e-3
        u = np.zeros(n)
        u[0::3] = a * mesh.node_xyz[:, 0]
        assert abs(eng.energy(u) - (lam / 2 + mu) * a * a) <= 1e-10 * (lam / 2 + mu) * a * a
    finally:
        eng.close()


def test_ten_million_p2_tets_on_one_gpu():
    """10.1 M P2 tets (119^3 Kuhn grid, 41 M dofs, 3.5e9 non-zeros: past 2^31 entries, 30 GB of values) on ONE device,
    through size-independent properties: NeoHookean at rest, null space and symmetry of the Hessian, agreement of the
    row assembly with the element energy kernel, a closed-form energy and a finite-difference check."""
    n_side = 119
    mesh = M.cube_tets(n_side, 2)
    assert mesh.n_el == 6 * n_side ** 3 == 10110954 and mesh.n_nodes == (2 * n_side + 1) ** 3
    eng = E.Engine(0)
    try:
        eng.set_mesh(mesh)
        lam, mu = 2e4 * 0.3 / (1.3 * 0.4), 2e4 / 2.6
        eng.set_material(E.NEOHOOKEAN, lam, mu)
        nnz = eng.build_pattern()
        assert nnz == 3505892805 and nnz > 2 ** 31
        n = mesh.n_nodes * 3
        zero = np.zeros(n)
        assert abs(eng.energy(zero)) < 1e-6 and np.abs(eng.gradient(zero)).max() < 1e-9
        eng.hessian(zero, fetch=False)
        rng = np.random.default_rng(11)
        x, y = rng.standard_normal(n), rng.standard_normal(n)
        Ax, Ay = eng.spmv(x), eng.spmv(y)
        scale = np.abs(Ax).max()
        assert abs(y @ Ax - x @ Ay) <= 1e-11 * abs(y @ Ax)
        for c in range(3):
            t = np.zeros(n)
            t[c::3] = 1.0
            assert np.abs(eng.spmv(t)).max() <= 1e-10 * scale
        rot = np.zeros(n)
        rot[0::3], rot[1::3] = -mesh.node_xyz[:, 1], mesh.node_xyz[:, 0]
        assert np.abs(eng.spmv(rot)).max() <= 1e-9 * scale
        del Ay, y, rot
        # Hessian at rest == linear elasticity: x^T H x = 2 * (linear-elastic energy of x), closed-form stretch energy
        eng.set_material(E.LINEAR_ELASTICITY, lam, mu)
        assert abs(x @ Ax - 2.0 * eng.energy(x)) <= 1e-10 * abs(x @ Ax)
        a = 1e-3
        u = np.zeros(n)
        u[0::3] = a * mesh.node_xyz[:, 0]
        assert abs(eng.energy(u) - (lam / 2 + mu) * a * a) <= 1e-10 * (lam / 2 + mu) * a * a
        eng.assemble_linear()
        assert np.abs(eng.spmv(x) - Ax).max() <= 1e-11 * scale
        # deformed state: energy / gradient / Hessian agree by central differences along the gradient
        eng.set_material(E.NEOHOOKEAN, lam, mu)
        u = M.seeded_displacement(mesh, 3, 1e-2 / n_side)
        g = eng.gradient(u)
        d = g / np.abs(g).max()
        hh = 1e-6
        fd = (eng.energy(u + hh * d) - eng.energy(u - hh * d)) / (2 * hh)
        assert abs(fd - g @ d) <= 1e-5 * abs(g @ d)
        eng.hessian(u, fetch=False)
        Hd = eng.spmv(d)
        gfd = (eng.gradient(u + hh * d) - eng.gradient(u - hh * d)) / (2 * hh)
        assert np.linalg.norm(gfd - Hd) <= 1e-5 * np.linalg.norm(Hd)
        del gfd, Hd, g, d
        # CG at this size sums 213 k per-slice partials of p.Ap with the two-stage kernel (fixed ranges, fixed order):
        # two runs give the same bits, and the residual the solver reports is the one of the iterate it returns
        bd = M.boundary_dofs(mesh, 3, "x0")
        eng.set_dirichlet(bd)
        b = rng.standard_normal(n)
        x1, it1, err1 = eng.cg(b, None, E.PRECOND_JACOBI, 24, 1e-12)
        x2, it2, err2 = eng.cg(b, None, E.PRECOND_JACOBI, 24, 1e-12)
        assert it1 == it2 == 24 and err1 == err2 and np.array_equal(x1, x2)
        free = np.ones(n, dtype=bool)
        free[bd] = False
        r = (b - eng.spmv(x1))[free]
        assert abs(np.linalg.norm(r) / np.linalg.norm(b[free]) - err1) <= 1e-6 * err1
    finally:
        eng.close()


def test_placement_search_changes_nothing_but_where_the_values_lie(monkeypatch):
    """A CG solve that has proven long (192 iterations done, 256 or more to go) tries several allocations of the solver's
    value array and keeps the fastest (polyfem_amd/csrc/pfx_sell.hip: sell_place).  Results, the context's memory record
    and the products the caller is charged for must be what they are without the search; short solves never search."""
    import torch
    mesh = M.cube_tets(24, 2)  # 82 944 P2 tets, 117 649 nodes (the search needs the slice-ordered product: >= 65 536 rows)
    monkeypatch.setenv("PFX_PLACEMENT_MIN_MB", "16")  # default threshold: 1 GiB of values
    n = mesh.n_nodes * 3
    b = np.random.default_rng(5).standard_normal(n)
    out = {}
    for tries in ("0", "4"):
        monkeypatch.setenv("PFX_PLACEMENT_TRIES", tries)
        torch.cuda.synchronize()
        eng = E.Engine(0)
        try:
            eng.set_mesh(mesh)
            lam, mu = 2e4 * 0.3 / (1.3 * 0.4), 2e4 / 2.6
            eng.set_material(E.LINEAR_ELASTICITY, lam, mu)
            eng.build_pattern()
            eng.assemble_linear()
            eng.set_dirichlet(M.boundary_dofs(mesh, 3, "x0"))
            eng.timing_enable(True)
            xs, its, errs = eng.cg(b, None, E.PRECOND_JACOBI, 100, 1e-30)  # too short to be worth a search
            _, searches_short = eng.timing_get("placement")
            x, it, err = eng.cg(b, None, E.PRECOND_JACOBI, 460, 1e-30)  # 192 done, 268 to go: searches (once)
            x2, it2, err2 = eng.cg(b, None, E.PRECOND_JACOBI, 460, 1e-30)
            _, searches = eng.timing_get("placement")
            _, products = eng.timing_get("spmv")
            eng.timing_enable(False)
            held = eng.info().bytes_device
            assert searches_short == 0 and searches == (1 if tries != "0" else 0), (searches_short, searches)
            assert it == it2 == 460 and err == err2 and np.array_equal(x, x2)
            # on request: the single-precision copy of an AMG hierarchy (its solves are too short to trigger a search)
            eng.amg_setup()
            eng.timing_enable(True)
            eng.tune_placement()
            eng.tune_placement()  # once per array
            _, searches_all = eng.timing_get("placement")  # the family keeps counting: the CG's search + the copy's
            eng.timing_enable(False)
            assert searches_all == (2 if tries != "0" else 0), searches_all
            xa, ita, erra = eng.cg(b, None, E.PRECOND_AMG, 200, 1e-10)
            assert erra <= 1e-10
            out[tries] = (x, it, err, held, products, xa, ita, erra)
        finally:
            eng.close()
    x0, it0, err0, held0, products0, xa0, ita0, erra0 = out["0"]
    x4, it4, err4, held4, products4, xa4, ita4, erra4 = out["4"]
    assert it0 == it4 and err0 == err4 and np.array_equal(x0, x4)
    assert held0 == held4 and products0 == products4
    assert ita0 == ita4 and erra0 == erra4 and np.array_equal(xa0, xa4)


def test_two_million_p2_triangles_properties():
    """2-D at size (1000^2 cells = 2 M P2 triangles, 4 M nodes, 8 M dofs: the slice-ordered product and the multi-window row
    groups of the block-size-2 path), through properties the oracle is not needed for: a homogeneous deformation has the
    energy psi(F) |Omega| exactly (constant integrand) and is in equilibrium in the interior; the Hessian at rest is the
    linear stiffness matrix, symmetric, with the three rigid modes in its null space; the CG reaches its tolerance and is
    bitwise repeatable."""
    mesh = M.square_tris(1000, 2, jitter=0.2, seed=11)
    assert mesh.n_el == 2_000_000
    n = 2 * mesh.n_nodes
    E_, nu = 2e4, 0.3
    lam, mu = nu * E_ / (1 - nu * nu), E_ / (2 * (1 + nu))
    eng = E.Engine(0)
    try:
        eng.set_mesh(mesh)
        eng.set_material(E.NEOHOOKEAN, lam, mu)
        eng.build_pattern()
        pos = M.node_positions(mesh)
        F = np.array([[1.07, 0.05], [-0.03, 0.94]])
        u = (pos @ (F - np.eye(2)).T).reshape(-1)
        J = np.linalg.det(F)
        psi = mu / 2 * ((F * F).sum() - 2 - 2 * np.log(J)) + lam / 2 * np.log(J) ** 2
        e = eng.energy(u)
        assert abs(e - psi * 1.0) <= 1e-11 * psi, (e, psi)
        g = eng.gradient(u).reshape(-1, 2)
        lo, hi = pos.min(axis=0), pos.max(axis=0)
        interior = np.all((pos > lo + 1e-9) & (pos < hi - 1e-9), axis=1)
        assert np.abs(g[interior]).max() <= 1e-9 * np.abs(g).max()
        # the boundary tractions of the homogeneous stress sum to zero
        assert np.abs(g.sum(axis=0)).max() <= 1e-9 * np.abs(g).sum()
        # at rest: Hessian == linear stiffness
        eng.hessian(np.zeros(n), fetch=False)
        rng = np.random.default_rng(3)
        x, y = rng.standard_normal(n), rng.standard_normal(n)
        Hx, Hy = eng.spmv(x), eng.spmv(y)
        assert abs(y @ Hx - x @ Hy) <= 1e-11 * abs(y @ Hx)
        scale = np.abs(Hx).max()
        modes = [np.tile([1.0, 0.0], mesh.n_nodes), np.tile([0.0, 1.0], mesh.n_nodes), np.stack([-pos[:, 1], pos[:, 0]], axis=1).reshape(-1)]
        for t in modes:
            assert np.abs(eng.spmv(t)).max() <= 1e-9 * scale
        eng.set_material(E.LINEAR_ELASTICITY, lam, mu)
        eng.assemble_linear()
        assert np.abs(eng.spmv(x) - Hx).max() <= 1e-10 * scale
        # clamped on x = 0: CG to 1e-8 in the limit of iterations it is given, twice, bit for bit
        bd = M.boundary_dofs(mesh, 2, "x0")
        eng.set_dirichlet(bd)
        b = np.zeros(n)
        b[1::2] = -1.0 / mesh.n_nodes
        x1, it1, err1 = eng.cg(b, None, E.PRECOND_JACOBI, 300, 1e-30)
        x2, it2, err2 = eng.cg(b, None, E.PRECOND_JACOBI, 300, 1e-30)
        assert it1 == it2 == 300 and np.array_equal(x1, x2) and err1 == err2
        assert not x1[bd].any() and np.isfinite(x1).all()
        # the reported residual is the residual of the returned iterate (eng.spmv applies the masked operator)
        free = np.ones(n, dtype=bool)
        free[bd] = False
        r = (b - eng.spmv(x1))[free]
        assert abs(np.linalg.norm(r) / np.linalg.norm(b[free]) - err1) <= 1e-6 * err1
        eng.set_dirichlet(np.zeros(0, dtype=np.int32))
    finally:
        eng.close()
