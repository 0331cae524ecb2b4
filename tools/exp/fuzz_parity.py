"""Randomised parity sweep (GPU box): small meshes of every element kind with random node numberings, elements removed at
random and jittered vertices; linear assembly (pattern bit-exact, values), NeoHookean energy / gradient / Hessian and a
Jacobi-CG solve against the oracle.  usage: python tools/exp/fuzz_parity.py [n_cases] [seed]"""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch  # noqa: F401
from polyfem_amd import engine as E, mesh as M
from oracle import pf_oracle as O
from tests.helpers import csc_to_scipy, rel_err

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
eng = E.Engine(0)
fails = 0


def scramble(m, rng):
    # drop elements at random, renumber the nodes that remain by a random permutation
    keep = rng.random(m.n_el) < rng.uniform(0.5, 1.0)
    if not keep.any():
        keep[rng.integers(m.n_el)] = True
    conn, gconn = m.conn[keep], m.gconn[keep]
    used = np.unique(conn)
    perm = rng.permutation(used.size)
    new_id = np.full(m.n_nodes, -1, dtype=np.int64)
    new_id[used] = perm
    gused = np.unique(gconn)
    gperm = rng.permutation(gused.size)
    gnew = np.full(m.xyz.shape[0], -1, dtype=np.int64)
    gnew[gused] = gperm
    xyz = np.zeros((gused.size, m.xyz.shape[1]))
    xyz[gperm] = m.xyz[gused]
    node_xyz = np.zeros((used.size, m.node_xyz.shape[1]))
    node_xyz[perm] = m.node_xyz[used]
    return M.Mesh(m.kind, m.gkind, np.ascontiguousarray(new_id[conn].astype(np.int32)), np.ascontiguousarray(gnew[gconn].astype(np.int32)),
                  np.ascontiguousarray(xyz), int(used.size), node_xyz)


for case in range(n_cases):
    kind = rng.integers(6)
    n = int(rng.integers(1, 5))
    jit = float(rng.choice([0.0, 0.15, 0.3]))
    if kind < 2:
        m = M.cube_tets(n, kind + 1, jitter=jit, seed=int(rng.integers(1 << 30)))
    elif kind < 4:
        m = M.cube_hexes(min(n, 3), kind - 1)
    else:
        m = M.square_tris(n + 1, kind - 3, jitter=jit, seed=int(rng.integers(1 << 30)))
    m = scramble(m, rng)
    dim = M.KIND_DIM[m.kind]
    lam, mu = O.lame(float(rng.uniform(1e3, 1e6)), float(rng.uniform(0.1, 0.45)), dim)
    tag = f"case {case}: {M.KIND_NAME[m.kind]} n={n} el={m.n_el} nodes={m.n_nodes} jitter={jit}"
    try:
        for model in (E.LAPLACIAN, E.LINEAR_ELASTICITY):
            eng.set_mesh(m)
            eng.set_material(model, lam, mu)
            eng.build_pattern()
            eng.assemble_linear()
            rp, ci, v = eng.get_linear_csr()
            ref = O.assemble_linear(m, model, lam, mu)
            assert np.array_equal(rp, ref[0].astype(np.int64)) and np.array_equal(ci, ref[1]), "linear pattern"
            assert rel_err(v, ref[2]) <= 1e-12, ("linear values", rel_err(v, ref[2]))
        eng.set_material(E.NEOHOOKEAN, lam, mu)
        nnz = eng.build_pattern()
        x = M.seeded_displacement(m, dim, 5e-3 / (n + 1), seed=int(rng.integers(1 << 30)))
        e, g, h = eng.energy(x), eng.gradient(x), eng.hessian(x)
        eo = O.assemble_energy(m, O.NEOHOOKEAN, lam, mu, x)
        go = O.assemble_gradient(m, O.NEOHOOKEAN, lam, mu, x)
        ho = O.assemble_hessian(m, O.NEOHOOKEAN, lam, mu, x)
        rp, ci = eng.get_pattern()
        assert np.array_equal(rp, ho[0].astype(np.int64)) and np.array_equal(ci, ho[1]) and nnz == ho[1].size, "hessian pattern"
        # the density mu/2 (|F|^2 - d - 2 ln J) cancels to O(strain^2): with few quadrature points its rounding error, not a
        # difference in what is computed, decides the last digits -- the bar scales with the cancellation
        strain2 = max(eo / (mu * max(abs(csc_to_scipy(O.assemble_mass(m, 1, 1.0), m.n_nodes).sum()), 1e-300)), 1e-300)
        assert abs(e - eo) <= max(1e-12, 4e-16 / strain2) * abs(eo), ("energy", e, eo, strain2)
        assert rel_err(g, go) <= 1e-12 and rel_err(h, ho[2]) <= 1e-12, ("grad/hess", rel_err(g, go), rel_err(h, ho[2]))
        # projected Hessian, per-element materials, mass matrix and right-hand side
        hp = eng.hessian(x, project_to_psd=True)
        hpo = O.assemble_hessian_psd(m, O.NEOHOOKEAN, lam, mu, x)[2]
        assert np.abs(hp - hpo).max() <= 1e-9 * np.abs(hpo).max(), ("psd", np.abs(hp - hpo).max() / np.abs(hpo).max())
        lam_e, mu_e = lam * rng.uniform(0.5, 2, m.n_el), mu * rng.uniform(0.5, 2, m.n_el)
        eng.set_material(E.NEOHOOKEAN, lam_e, mu_e)
        assert rel_err(eng.hessian(x), O.assemble_hessian(m, O.NEOHOOKEAN, lam_e, mu_e, x)[2]) <= 1e-12, "per-element hessian"
        assert rel_err(eng.gradient(x), O.assemble_gradient(m, O.NEOHOOKEAN, lam_e, mu_e, x)) <= 1e-12, "per-element gradient"
        eng.set_material(E.NEOHOOKEAN, lam, mu)
        rho = rng.uniform(0.5, 2, m.n_el)
        eng.assemble_mass(rho)
        rpm, cim, vm = eng.get_linear_csr()
        mo = O.assemble_mass(m, dim, rho)
        assert np.array_equal(rpm, mo[0].astype(np.int64)) and np.array_equal(cim, mo[1]) and rel_err(vm, mo[2]) <= 1e-12, "mass"
        f = rng.standard_normal(dim)
        assert rel_err(eng.assemble_rhs(f, rho), O.assemble_rhs(m, dim, f, rho)) <= 1e-12, "rhs"
        eng.hessian(x, fetch=False)
        # the bare Hessian of a free (or, with elements removed, disconnected) body is singular, and one-point Q1 hexahedra
        # have hourglass modes: the solver is compared on H + shift I with a shift that makes the system well conditioned
        ndof = m.n_nodes * dim
        shift = float(np.abs(ho[2]).max()) * 0.05
        eng.set_diag_shift(shift)
        b = rng.standard_normal(ndof)
        xs, it, err = eng.cg(b, None, E.PRECOND_JACOBI, 2000, 1e-10)
        eng.set_diag_shift(0.0)
        A = csc_to_scipy(ho, ndof).tolil()
        A.setdiag(A.diagonal() + shift)
        A = A.tocsc()
        A.sort_indices()
        xr, itr, _ = O.cg((A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data), b, jacobi=True, max_iter=2000, tol=1e-10)
        assert abs(it - itr) <= 2, ("cg iterations", it, itr)
        assert np.linalg.norm(xs - xr) <= 1e-8 * np.linalg.norm(xr), ("cg solution", np.linalg.norm(xs - xr) / np.linalg.norm(xr))
    except AssertionError as ex:
        fails += 1
        print("FAIL", tag, ex.args, flush=True)
    except Exception as ex:
        fails += 1
        print("ERROR", tag, repr(ex), flush=True)
    if case % 25 == 24:
        print(f"... {case + 1} cases, {fails} failures", flush=True)
print(f"done: {n_cases} cases, {fails} failures")
sys.exit(1 if fails else 0)
