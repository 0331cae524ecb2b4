"""The polysolve-shaped LinearSolver on matrices it is HANDED (pfx_solver_analyze / pfx_solver_factorize), as the
reference's Newton solver uses it: analyzePattern / factorize / solve on the reduced sum of all forms' Hessians
(src/polyfem/solver/SparseNewtonDescentSolver.tpp:128-155, src/polyfem/solver/FullNLProblem.cpp:115-135,
src/polyfem/solver/NLProblem.cpp:92-99) and on the stiffness matrix of a linear problem
(src/polyfem/state/StateSolveLinear.cpp:27-29,51-54).

Bars: iteration count within +-2 of the oracle's Eigen-order CG on the same matrix, relative residual <= 1e-10.
Eigen / polysolve are not part of the reference tree: the oracle's CG restates Eigen 3.4 (parity unpinned)."""
import numpy as np
import pytest
import scipy.sparse as sp

from polyfem_amd import engine as E
from polyfem_amd import mesh as M
from tests.helpers import csc_to_scipy

pytestmark = pytest.mark.gpu


def _neohookean_reduced(oracle, n=3, dirichlet="x0", jitter=0.1):
    mesh = M.cube_tets(n, 2, jitter=jitter)
    lam, mu = oracle.lame(2e4, 0.3)
    x = M.seeded_displacement(mesh, 3, 1e-3)
    full = oracle.assemble_hessian(mesh, oracle.NEOHOOKEAN, lam, mu, x)
    bd = M.boundary_dofs(mesh, 3, dirichlet)
    red = oracle.full_to_reduced_matrix(full, bd)
    keep = np.setdiff1d(np.arange(mesh.n_nodes * 3), bd)
    g = oracle.assemble_gradient(mesh, oracle.NEOHOOKEAN, lam, mu, x)
    return mesh, red, keep, -g[keep]


def _solve_and_compare(oracle, solver, A, red, b, max_iter=1000, tol=1e-12, jacobi=True):
    solver.set_parameters({"max_iter": max_iter, "tolerance": tol})
    solver.analyze_pattern(A, A.rows())
    solver.factorize(A)
    x = np.zeros(A.rows())
    solver.solve(b, x)
    info = solver.get_info()
    xr, it_ref, err_ref = oracle.cg(red, b, jacobi=jacobi, max_iter=max_iter, tol=tol)
    assert abs(info["solver_iter"] - it_ref) <= 2, (info, it_ref)
    S = csc_to_scipy(red, A.rows())
    assert np.linalg.norm(S @ x - b) <= 1e-10 * np.linalg.norm(b)
    assert np.linalg.norm(x - xr) <= 1e-8 * np.linalg.norm(xr)
    return x, info


def test_reduced_neohookean_hessian_handed_in(oracle):
    """(a) the oracle's full_to_reduced_matrix of a NeoHookean Hessian, handed in by value."""
    mesh, red, keep, b = _neohookean_reduced(oracle)
    A = E.StiffnessMatrix(keep.size, red[0].astype(np.int32), red[1], red[2])
    solver = E.LinearSolver.create("Eigen::ConjugateGradient", "Eigen::DiagonalPreconditioner")
    try:
        _solve_and_compare(oracle, solver, A, red, b)
        # whole nodes are constrained on the face x = 0: the reduced numbering keeps the 3-alignment -> 3 x 3 blocks
        bs, nb, stored = solver.own.solver_layout()
        assert bs == 3 and stored == 9 * nb and stored >= red[1].size
        # a second matrix on the same pattern: factorize only (what every later Newton iteration does)
        red2 = (red[0], red[1], red[2] * 1.5)
        A2 = E.StiffnessMatrix(keep.size, A.outer, A.inner, red2[2])
        solver.factorize(A2)
        x = np.zeros(keep.size)
        solver.solve(b, x)
        xr, it_ref, _ = oracle.cg(red2, b, jacobi=True, max_iter=1000, tol=1e-12)
        assert abs(solver.get_info()["solver_iter"] - it_ref) <= 2
        assert np.linalg.norm(x - xr) <= 1e-8 * np.linalg.norm(xr)
    finally:
        solver.close()


def test_sum_of_forms_is_what_gets_solved(oracle):
    """(b) the same matrix plus a second form (a positive diagonal: lumped inertia / penalty term).  A solver that
    ignored the matrix it is handed would return the solution of (a)."""
    mesh, red, keep, b = _neohookean_reduced(oracle)
    n = keep.size
    d = 50.0 + 100.0 * np.random.default_rng(3).random(n)
    S = (csc_to_scipy(red, n) + sp.diags(d)).tocsc()
    S.sort_indices()
    summed = (S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data.copy())
    A = E.StiffnessMatrix(n, summed[0], summed[1], summed[2])
    solver = E.LinearSolver.create("Eigen::ConjugateGradient", "Eigen::DiagonalPreconditioner")
    try:
        x, _ = _solve_and_compare(oracle, solver, A, summed, b)
        x_a, _, _ = oracle.cg(red, b, jacobi=True, max_iter=1000, tol=1e-12)
        assert np.linalg.norm(x - x_a) > 1e-3 * np.linalg.norm(x_a)
    finally:
        solver.close()


def test_resident_matrix_is_used_only_while_unchanged(oracle):
    """The zero-copy path needs the tag of the device-resident matrix; once the caller adds another form to the matrix
    object the tag is gone and the values that are handed in are the ones solved with."""
    mesh = M.cube_tets(3, 2, jitter=0.1)
    lam, mu = oracle.lame(2e4, 0.3)
    nl = E.NLAssembler("NeoHookean")
    nl.add_multimaterial(0, {"E": 2e4, "nu": 0.3})
    cache = E.SpareMatrixCache()
    x0 = M.seeded_displacement(mesh, 3, 1e-3)
    H = nl.assemble_hessian(True, mesh.n_nodes, False, mesh, mesh, None, 0.0, x0, x0, cache)
    n = H.rows()
    solver = E.LinearSolver.create("Eigen::ConjugateGradient", "Eigen::DiagonalPreconditioner", nl.engine)
    try:
        # unconstrained NeoHookean Hessian is singular (rigid modes): regularise both ways by a mass-like diagonal
        rng = np.random.default_rng(9)
        b = rng.standard_normal(n)
        solver.analyze_pattern(H, n)
        assert solver.active is nl.engine  # resident: no upload
        d = 10.0 + rng.random(n)
        H.add_diagonal(d)                   # "hessian += other form": the tag is dropped
        assert H.resident is None
        full = (H.outer, H.inner, H.values)
        solver.set_parameters({"max_iter": 2000, "tolerance": 1e-12})
        solver.analyze_pattern(H, n)
        solver.factorize(H)
        assert solver.active is solver.own
        x = np.zeros(n)
        solver.solve(b, x)
        xr, it_ref, _ = oracle.cg(full, b, jacobi=True, max_iter=2000, tol=1e-12)
        assert abs(solver.get_info()["solver_iter"] - it_ref) <= 2
        assert np.linalg.norm(H.to_scipy_csc() @ x - b) <= 1e-10 * np.linalg.norm(b)
    finally:
        solver.close()
        nl.engine.close()


def test_scalar_layout_and_unsymmetric_pattern(oracle):
    """Single dofs removed (not whole nodes) break the 3-alignment -> scalar format; and a pattern that is not symmetric
    (explicit entries on one side only) is transposed correctly: y = A x, not A^T x."""
    mesh, red, keep, b = _neohookean_reduced(oracle, dirichlet="x0")
    n = keep.size
    S = csc_to_scipy(red, n).tocsc()
    # drop dof 1 and dof 5: the remaining numbering is no longer aligned with the nodes
    sel = np.setdiff1d(np.arange(n), [1, 5])
    R = S[sel][:, sel].tocsc()
    R.sort_indices()
    eng = E.Engine(0)
    try:
        eng.solver_analyze(sel.size, R.indptr.astype(np.int32), R.indices.astype(np.int32), 0)
        assert eng.solver_layout()[0] == 1
        eng.solver_factorize(R.data)
        xv = np.random.default_rng(1).standard_normal(sel.size)
        assert np.abs(eng.spmv(xv) - R @ xv).max() <= 1e-12 * np.abs(R @ xv).max()
        x, it, err = eng.cg(b[sel], None, E.PRECOND_JACOBI, 1000, 1e-12)
        xr, it_ref, _ = oracle.cg((R.indptr.astype(np.int32), R.indices.astype(np.int32), R.data), b[sel], jacobi=True, max_iter=1000, tol=1e-12)
        assert abs(it - it_ref) <= 2 and err <= 1e-10
        # unsymmetric pattern and values: upper-triangular perturbation with explicit entries
        U = sp.triu(S, k=1).tocsc()
        U.data = 1e-3 * np.abs(U.data)
        T = (S + U).tocsc()
        extra = sp.csc_matrix((np.array([7.0]), (np.array([0]), np.array([n - 1]))), shape=(n, n))  # entry (0, n-1) only
        T = (T + extra).tocsc()
        T.sort_indices()
        for bsz in (0, 1, 3):
            eng.solver_analyze(n, T.indptr.astype(np.int64), T.indices.astype(np.int32), bsz)
            eng.solver_factorize(T.data)
            xv = np.random.default_rng(2).standard_normal(n)
            y = eng.spmv(xv)
            assert np.abs(y - T @ xv).max() <= 1e-12 * np.abs(T @ xv).max(), bsz
            assert np.abs(y - T.T @ xv).max() > 1e-6 * np.abs(y).max()
    finally:
        eng.close()


def test_stiffness_with_dropped_zeros_and_dirichlet_solve(oracle):
    """State::solve_linear's path (state/StateSolveLinear.cpp:27-29,51-54): the LinearElasticity stiffness matrix in the
    reference's value-dependent pattern (exact zeros dropped, so not every 3 x 3 block is full) handed to
    dirichlet_solve.  The blocks are padded with explicit zeros and the solve equals the oracle's."""
    mesh = M.cube_tets(3, 2)
    lam, mu = oracle.lame(1e5, 0.3)
    K = oracle.assemble_linear(mesh, oracle.LINEAR_ELASTICITY, lam, mu)
    n = mesh.n_nodes * 3
    eng = E.Engine(0)
    try:
        eng.solver_analyze(n, K[0].astype(np.int32), K[1], 0)
        bs, nb, stored = eng.solver_layout()
        assert bs == 3 and stored > K[1].size  # axis-aligned mesh: exact zeros were dropped by the assembler
        eng.solver_factorize(K[2])
        bd = M.boundary_dofs(mesh, 3, "x0")
        eng.set_dirichlet(bd)
        rng = np.random.default_rng(4)
        b = rng.standard_normal(n)
        b[bd] = 0.01 * rng.standard_normal(bd.size)  # prescribed values
        x, it, err = eng.dirichlet_solve(b, E.PRECOND_JACOBI, 2000, 1e-12)
        Am, bm = oracle.dirichlet_modify(K, b, bd)
        xr, it_ref, _ = oracle.cg(Am, bm, jacobi=True, max_iter=2000, tol=1e-12)
        assert abs(it - it_ref) <= 2, (it, it_ref)
        assert np.allclose(x[bd], b[bd], rtol=0, atol=0)
        assert np.linalg.norm(x - xr) <= 1e-8 * np.linalg.norm(xr)
    finally:
        eng.close()


def test_factorize_reports_what_the_newton_solver_regularises_on(oracle):
    mesh, red, keep, b = _neohookean_reduced(oracle, n=2)
    n = keep.size
    A = E.StiffnessMatrix(n, red[0].astype(np.int32), red[1], red[2].copy())
    solver = E.LinearSolver.create("Eigen::ConjugateGradient", "Eigen::DiagonalPreconditioner")
    try:
        solver.analyze_pattern(A, n)
        S = csc_to_scipy(red, n).tocsc()
        diag_pos = np.array([S.indptr[j] + np.searchsorted(S.indices[S.indptr[j]:S.indptr[j + 1]], j) for j in range(n)])
        bad = red[2].copy()
        bad[diag_pos[7]] = -1.0  # indefinite
        with pytest.raises(RuntimeError, match="Factorization failed"):
            solver.factorize(E.StiffnessMatrix(n, A.outer, A.inner, bad))
        with pytest.raises(E.PfxError):  # nothing to solve with after a failed factorization
            solver.solve(b, np.zeros(n))
        bad = red[2].copy()
        bad[3] = np.nan
        with pytest.raises(RuntimeError, match="Factorization failed"):
            solver.factorize(E.StiffnessMatrix(n, A.outer, A.inner, bad))
        # the regularised matrix (hessian += w I, SparseNewtonDescentSolver.tpp:116-121) factorizes
        ok = E.StiffnessMatrix(n, A.outer, A.inner, red[2].copy())
        ok.add_diagonal(1e-3)
        solver.analyze_pattern(ok, n)
        solver.factorize(ok)
        x = np.zeros(n)
        solver.solve(b, x)
        assert solver.get_info()["solver_error"] <= 1e-10
    finally:
        solver.close()


def test_argument_checks_of_the_solver_entry_points(oracle):
    mesh, red, keep, b = _neohookean_reduced(oracle, n=2)
    n = keep.size
    outer, inner = red[0].astype(np.int32), red[1]
    eng = E.Engine(0)
    try:
        with pytest.raises(E.PfxError):  # factorize before analyze
            eng.solver_factorize(red[2])
        with pytest.raises(E.PfxError):
            eng.solver_analyze(n, outer, inner, 4)
        with pytest.raises(E.PfxError):
            eng.solver_analyze(n - n % 2 + 1, outer, inner, 2)  # an odd number of rows in 2 x 2 blocks
        bad = inner.copy()
        bad[5] = n
        with pytest.raises(E.PfxError):
            eng.solver_analyze(n, outer, bad, 0)
        dup = inner.copy()
        dup[1] = dup[0]
        with pytest.raises(E.PfxError, match="twice"):
            eng.solver_analyze(n, outer, dup, 0)
        eng.solver_analyze(n, outer, inner, 0)
        with pytest.raises(E.PfxError):  # no values yet
            eng.cg(b)
        with pytest.raises(E.PfxError):  # no mesh behind this context
            eng.energy(np.zeros(n))
        eng.solver_factorize(red[2])
        x, it, err = eng.cg(b)
        assert err <= 1e-10
        # a context with a mesh refuses a caller's matrix
        e2 = E.Engine(0)
        try:
            e2.set_mesh(mesh)
            with pytest.raises(E.PfxError):
                e2.solver_analyze(n, outer, inner, 0)
        finally:
            e2.close()
    finally:
        eng.close()


def test_amg_on_a_handed_in_matrix(oracle):
    """AMGCL's slot with a matrix handed in by value: block-3 with node positions (rigid-body coarse spaces), and the
    scalar format without positions.  Own design, so the bar is convergence, not an iteration count of the reference."""
    mesh = M.cube_tets(6, 2)
    lam, mu = oracle.lame(1e5, 0.3)
    K = oracle.assemble_linear(mesh, oracle.LINEAR_ELASTICITY, lam, mu)
    n = mesh.n_nodes * 3
    bd = M.boundary_dofs(mesh, 3, "x0")
    red = oracle.full_to_reduced_matrix(K, bd)
    keep = np.setdiff1d(np.arange(n), bd)
    b = np.random.default_rng(8).standard_normal(keep.size)
    S = csc_to_scipy(red, keep.size)
    A = E.StiffnessMatrix(keep.size, red[0].astype(np.int32), red[1], red[2])
    pos = M.node_positions(mesh)[keep[::3] // 3]
    it_jacobi = oracle.cg(red, b, jacobi=True, max_iter=5000, tol=1e-10)[1]
    for positions in (pos, None):
        solver = E.LinearSolver.create("AMGCL", "")
        try:
            solver.node_positions = positions
            solver.block_size = 3 if positions is not None else 1
            solver.analyze_pattern(A, A.rows())
            solver.factorize(A)
            x = np.zeros(keep.size)
            solver.solve(b, x)
            info = solver.get_info()
            assert np.linalg.norm(S @ x - b) <= 1e-9 * np.linalg.norm(b)
            assert info["solver_iter"] < it_jacobi / 2, (info, it_jacobi)
        finally:
            solver.close()
    # block-3 without positions is refused with a message that says what to do
    eng = E.Engine(0)
    try:
        eng.solver_analyze(keep.size, red[0].astype(np.int32), red[1], 3)
        eng.solver_factorize(red[2])
        with pytest.raises(E.PfxError, match="node positions"):
            eng.cg(b, None, E.PRECOND_AMG, 100, 1e-10)
    finally:
        eng.close()


def test_two_by_two_blocks_handed_in(oracle):
    """A 2-D elasticity matrix handed in by value with block_size 2: 2 x 2 node blocks in the solver copy.  The reduced
    NeoHookean Hessian of a triangle mesh (whole nodes constrained) through Jacobi-CG against the oracle's Eigen-order CG,
    and through the AMG with node positions (rigid-body modes of the plane) against the scalar layout without them."""
    mesh = M.square_tris(12, 2, jitter=0.1, seed=2)
    lam, mu = oracle.lame(2e4, 0.3, 2)
    n = 2 * mesh.n_nodes
    x = M.seeded_displacement(mesh, 2, 2e-3)
    H = oracle.assemble_hessian(mesh, oracle.NEOHOOKEAN, lam, mu, x)
    bd = M.boundary_dofs(mesh, 2, "x0")
    red = oracle.full_to_reduced_matrix(H, bd)
    keep = np.setdiff1d(np.arange(n), bd)
    b = np.random.default_rng(8).standard_normal(keep.size)
    S = csc_to_scipy(red, keep.size)
    x_o, it_o, _ = oracle.cg(red, b, jacobi=True, max_iter=5000, tol=1e-10)
    eng = E.Engine(0)
    try:
        eng.solver_analyze(keep.size, red[0].astype(np.int32), red[1], 2)
        bs, n_blocks, stored = eng.solver_layout()
        assert bs == 2 and stored == 4 * n_blocks and stored <= 1.05 * red[1].size
        eng.solver_factorize(red[2])
        y = eng.spmv(b)
        assert np.abs(y - S @ b).max() <= 1e-12 * np.abs(S @ b).max()
        xs, it, err = eng.cg(b, None, E.PRECOND_JACOBI, 5000, 1e-10)
        assert abs(it - it_o) <= 2 and np.abs(xs - x_o).max() <= 1e-8 * np.abs(x_o).max(), (it, it_o)
        with pytest.raises(E.PfxError, match="node positions"):
            eng.cg(b, None, E.PRECOND_AMG, 100, 1e-10)
        pos = M.node_positions(mesh)[keep[::2] // 2]
        eng.solver_set_node_positions(np.hstack([pos, np.zeros((pos.shape[0], 1))]))
        xa, it_a, err_a = eng.cg(b, None, E.PRECOND_AMG, 500, 1e-10)
        assert err_a <= 1e-10 and np.linalg.norm(S @ xa - b) <= 1e-9 * np.linalg.norm(b)
        assert it_a < it_o / 4, (it_a, it_o)
        # the scalar layout of the same matrix: constants only in the coarse space
        eng.solver_analyze(keep.size, red[0].astype(np.int32), red[1], 1)
        eng.solver_factorize(red[2])
        xb, it_b, err_b = eng.cg(b, None, E.PRECOND_AMG, 2000, 1e-10)
        print("2 x 2 blocks handed in: Jacobi", it, "AMG (plane modes)", it_a, "AMG (scalar layout)", it_b)
        assert err_b <= 1e-10 and it_a <= it_b
    finally:
        eng.close()
