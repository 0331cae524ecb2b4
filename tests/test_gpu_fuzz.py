"""Randomised parity sweep on the GPU: small meshes of every element kind with RANDOM node numberings (the numbering is the
caller's, `Local2Global.index`), elements removed at random (holes, disconnected pieces, nodes of degree one) and jittered
vertices -- the inputs the structured meshes of the other tests never produce.  Linear assembly (pattern bit-exact, values
1e-12), NeoHookean energy / gradient / Hessian and a Jacobi-CG solve against the oracle, through the C ABI.

The energy bar is scaled with the cancellation in the reference's formula: mu/2 (|F|^2 - d - 2 ln J) is a difference of O(1)
terms that leaves O(strain^2); on a single element with one quadrature point the oracle itself is 1e-12 .. 3e-11 away from
a 50-digit evaluation of the same expression at strains of 4e-3 (measured with mpmath on a one-hexahedron mesh), so with few
points the last digits are rounding, not a difference in what is computed.  Gradient and Hessian have no such cancellation
and keep 1e-12."""
import numpy as np
import pytest

from polyfem_amd import engine as E
from polyfem_amd import mesh as M
from tests.helpers import csc_to_scipy, rel_err

pytestmark = pytest.mark.gpu


def _scramble(m, rng):
    keep = rng.random(m.n_el) < rng.uniform(0.5, 1.0)
    if not keep.any():
        keep[rng.integers(m.n_el)] = True
    conn, gconn = m.conn[keep], m.gconn[keep]
    used, gused = np.unique(conn), np.unique(gconn)
    perm, gperm = rng.permutation(used.size), rng.permutation(gused.size)
    new_id = np.full(m.n_nodes, -1, dtype=np.int64)
    new_id[used] = perm
    gnew = np.full(m.xyz.shape[0], -1, dtype=np.int64)
    gnew[gused] = gperm
    xyz = np.zeros((gused.size, m.xyz.shape[1]))
    xyz[gperm] = m.xyz[gused]
    node_xyz = np.zeros((used.size, m.node_xyz.shape[1]))
    node_xyz[perm] = m.node_xyz[used]
    return M.Mesh(m.kind, m.gkind, np.ascontiguousarray(new_id[conn].astype(np.int32)), np.ascontiguousarray(gnew[gconn].astype(np.int32)),
                  np.ascontiguousarray(xyz), int(used.size), node_xyz)


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_random_numberings_holes_and_jitter(oracle, seed):
    rng = np.random.default_rng(seed)
    eng = E.Engine(0)
    try:
        for case in range(30):
            kind = int(rng.integers(6))
            n = int(rng.integers(1, 5))
            jit = float(rng.choice([0.0, 0.15, 0.3]))
            if kind < 2:
                m = M.cube_tets(n, kind + 1, jitter=jit, seed=int(rng.integers(1 << 30)))
            elif kind < 4:
                m = M.cube_hexes(min(n, 3), kind - 1)
            else:
                m = M.square_tris(n + 1, kind - 3, jitter=jit, seed=int(rng.integers(1 << 30)))
            m = _scramble(m, rng)
            dim = M.KIND_DIM[m.kind]
            lam, mu = oracle.lame(float(rng.uniform(1e3, 1e6)), float(rng.uniform(0.1, 0.45)), dim)
            tag = f"seed {seed} case {case}: {M.KIND_NAME[m.kind]} n={n} elements={m.n_el} nodes={m.n_nodes} jitter={jit}"
            for model in (E.LAPLACIAN, E.LINEAR_ELASTICITY):
                eng.set_mesh(m)
                eng.set_material(model, lam, mu)
                eng.build_pattern()
                eng.assemble_linear()
                rp, ci, v = eng.get_linear_csr()
                ref = oracle.assemble_linear(m, model, lam, mu)
                assert np.array_equal(rp, ref[0].astype(np.int64)) and np.array_equal(ci, ref[1]), tag
                assert rel_err(v, ref[2]) <= 1e-12, tag
            eng.set_material(E.NEOHOOKEAN, lam, mu)
            nnz = eng.build_pattern()
            x = M.seeded_displacement(m, dim, 5e-3 / (n + 1), seed=int(rng.integers(1 << 30)))
            e, g, h = eng.energy(x), eng.gradient(x), eng.hessian(x)
            eo = oracle.assemble_energy(m, oracle.NEOHOOKEAN, lam, mu, x)
            go = oracle.assemble_gradient(m, oracle.NEOHOOKEAN, lam, mu, x)
            ho = oracle.assemble_hessian(m, oracle.NEOHOOKEAN, lam, mu, x)
            rp, ci = eng.get_pattern()
            assert np.array_equal(rp, ho[0].astype(np.int64)) and np.array_equal(ci, ho[1]) and nnz == ho[1].size, tag
            volume = abs(csc_to_scipy(oracle.assemble_mass(m, 1, 1.0), m.n_nodes).sum())
            strain2 = max(eo / (mu * volume), 1e-300)  # the energy density in units of mu: what the cancellation leaves
            assert abs(e - eo) <= max(1e-12, 1e-15 / strain2) * abs(eo), (tag, e, eo, strain2)
            assert rel_err(g, go) <= 1e-12 and rel_err(h, ho[2]) <= 1e-12, tag
            # the bare Hessian of a free (with elements removed: disconnected) body is singular and one-point Q1 hexahedra have
            # hourglass modes: the solver is compared on H + shift I, well conditioned
            ndof = m.n_nodes * dim
            shift = float(np.abs(ho[2]).max()) * 0.05
            eng.set_diag_shift(shift)
            b = rng.standard_normal(ndof)
            xs, it, _ = eng.cg(b, None, E.PRECOND_JACOBI, 2000, 1e-10)
            eng.set_diag_shift(0.0)
            A = csc_to_scipy(ho, ndof).tolil()
            A.setdiag(A.diagonal() + shift)
            A = A.tocsc()
            A.sort_indices()
            xr, itr, _ = oracle.cg((A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data), b, jacobi=True, max_iter=2000, tol=1e-10)
            assert abs(it - itr) <= 2, (tag, it, itr)
            assert np.linalg.norm(xs - xr) <= 1e-8 * np.linalg.norm(xr), tag
    finally:
        eng.close()


def _spd_tensor(rng, n, scale):
    a = rng.uniform(-1, 1, (n, n))
    return scale * (a @ a.T + 0.5 * n * np.eye(n))


@pytest.mark.parametrize("seed", [21, 22])
def test_random_meshes_tensor_and_mixed_models(oracle, seed):
    """The same scrambled meshes through the other local assemblers: HookeLinearElasticity with a random symmetric positive
    definite tensor (6 x 6, or 3 x 3 on triangles; pattern bit-exact, it depends on the values), SaintVenant from (lambda, mu)
    or such a tensor, and MultiModel with a random model per element and per-element Lame parameters."""
    rng = np.random.default_rng(seed)
    eng = E.Engine(0)
    try:
        for case in range(24):
            kind = int(rng.integers(6))
            n = int(rng.integers(1, 4))
            jit = float(rng.choice([0.0, 0.15, 0.3]))
            if kind < 2:
                m = M.cube_tets(n, kind + 1, jitter=jit, seed=int(rng.integers(1 << 30)))
            elif kind < 4:
                m = M.cube_hexes(min(n, 2), kind - 1)
            else:
                m = M.square_tris(n + 1, kind - 3, jitter=jit, seed=int(rng.integers(1 << 30)))
            m = _scramble(m, rng)
            dim = M.KIND_DIM[m.kind]
            ndof = m.n_nodes * dim
            young = float(rng.uniform(1e3, 1e6))
            lam, mu = oracle.lame(young, float(rng.uniform(0.1, 0.45)), dim)
            tag = f"seed {seed} case {case}: {M.KIND_NAME[m.kind]} n={n} elements={m.n_el} nodes={m.n_nodes} jitter={jit}"
            Ct = _spd_tensor(rng, 6 if dim == 3 else 3, young)
            eng.set_mesh(m)
            eng.set_elasticity_tensor(Ct)
            eng.build_pattern()
            eng.assemble_linear()
            rp, ci, v = eng.get_linear_csr()
            ref = csc_to_scipy(oracle.assemble_linear(m, oracle.HOOKE_LINEAR, Ct, None), ndof).tocsr()
            ref.sort_indices()
            assert np.array_equal(rp, ref.indptr) and np.array_equal(ci, ref.indices), tag
            assert rel_err(v, ref.data) <= 1e-12, tag
            x = M.seeded_displacement(m, dim, 3e-2 / (n + 1), seed=int(rng.integers(1 << 30)))
            # SaintVenant: (lambda, mu) or the tensor
            if rng.random() < 0.5:
                a, b = lam, mu
                eng.set_material(E.SAINT_VENANT, lam, mu)
            else:
                a, b = Ct, None
                eng.set_saint_venant_tensor(Ct)
            lam_e, mu_e = lam * rng.uniform(0.5, 2.0, m.n_el), mu * rng.uniform(0.5, 2.0, m.n_el)
            ids = rng.choice([oracle.LINEAR_ELASTICITY, oracle.NEOHOOKEAN, oracle.SAINT_VENANT], m.n_el).astype(np.int32)
            for model, pa, pb in ((oracle.SAINT_VENANT, a, b), (ids, lam_e, mu_e)):
                if model is ids:
                    eng.set_material(E.NEOHOOKEAN, lam_e, mu_e)
                    eng.set_element_models(ids)
                nnz = eng.build_pattern()
                e, g, h = eng.energy(x), eng.gradient(x), eng.hessian(x)
                eo = oracle.assemble_energy(m, model, pa, pb, x)
                go = oracle.assemble_gradient(m, model, pa, pb, x)
                ho = oracle.assemble_hessian(m, model, pa, pb, x)
                rp, ci = eng.get_pattern()
                assert np.array_equal(rp, ho[0].astype(np.int64)) and np.array_equal(ci, ho[1]) and nnz == ho[1].size, tag
                assert abs(e - eo) <= 1e-12 * abs(eo), (tag, e, eo)
                assert rel_err(g, go) <= 1e-12 and rel_err(h, ho[2]) <= 1e-12, tag
                hp = eng.hessian(x, project_to_psd=True)
                hpo = oracle.assemble_hessian_psd(m, model, pa, pb, x)
                assert np.abs(hp - hpo[2]).max() <= 1e-9 * np.abs(hpo[2]).max(), tag
    finally:
        eng.close()
