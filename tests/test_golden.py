"""Golden fixtures (tests/golden/*.npz, made by tests/tools/make_golden.py from the oracle): the oracle must keep
reproducing them (CPU), and the HIP path must match them through the C ABI (GPU)."""
import glob
import os

import numpy as np
import pytest

from polyfem_amd import mesh as M

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz")))
MODEL = {"Laplacian": 0, "LinearElasticity": 1, "NeoHookean": 2, "SaintVenant": 3}


def _mesh(d):
    kind = int(d["kind"])
    return M.Mesh(kind, M.GEOM_KIND[kind], d["conn"], d["gconn"], d["xyz"], int(d["n_nodes"]), None)


def test_fixtures_present():
    assert len(GOLDEN) >= 8


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_golden(oracle, path):
    d = np.load(path, allow_pickle=False)
    mesh, model = _mesh(d), MODEL[str(d["model"])]
    lam, mu = float(d["lam"]), float(d["mu"])
    if model >= 2:
        outer, inner, vals = oracle.assemble_hessian(mesh, model, lam, mu, d["x"])
        assert oracle.assemble_energy(mesh, model, lam, mu, d["x"]) == float(d["energy"])
        assert np.array_equal(oracle.assemble_gradient(mesh, model, lam, mu, d["x"]), d["gradient"])
    else:
        outer, inner, vals = oracle.assemble_linear(mesh, model, lam, mu)
    assert np.array_equal(outer, d["outer"]) and np.array_equal(inner, d["inner"]) and np.array_equal(vals, d["values"])
    red = oracle.full_to_reduced_matrix((outer, inner, vals), d["boundary"])
    xs, it, err = oracle.cg(red, d["cg_b"], jacobi=True, max_iter=1000, tol=1e-12)
    assert it == int(d["cg_iters"]) and np.array_equal(xs, d["cg_x"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_hip_matches_golden(path):
    from polyfem_amd import engine as E
    d = np.load(path, allow_pickle=False)
    mesh, model = _mesh(d), MODEL[str(d["model"])]
    bs = 1 if model == 0 else 3
    eng = E.Engine(0)
    eng.set_mesh(mesh)
    eng.set_material(model, float(d["lam"]), float(d["mu"]))
    eng.build_pattern()
    scale = np.abs(d["values"]).max()
    if model >= 2:
        v = eng.hessian(d["x"])
        rp, ci = eng.get_pattern()
        assert abs(eng.energy(d["x"]) - float(d["energy"])) <= 1e-12 * abs(float(d["energy"]))
        assert np.abs(eng.gradient(d["x"]) - d["gradient"]).max() <= 1e-12 * np.abs(d["gradient"]).max()
    else:
        eng.assemble_linear()
        rp, ci, v = eng.get_linear_csr()
    assert np.array_equal(rp, d["outer"].astype(np.int64)) and np.array_equal(ci, d["inner"])
    assert np.abs(v - d["values"]).max() <= 1e-12 * scale
    if float(d["cg_err"]) > 1e-10:  # Q1 with the default 1-point rule is rank deficient: no converged solve to compare
        eng.close()
        return
    eng.set_dirichlet(d["boundary"])
    n = int(d["n_nodes"]) * bs
    keep = np.setdiff1d(np.arange(n), d["boundary"])
    b = np.zeros(n)
    b[keep] = d["cg_b"]
    x, it, err = eng.cg(b, None, E.PRECOND_JACOBI, 1000, 1e-12)
    assert abs(it - int(d["cg_iters"])) <= 2 and err <= 1e-10
    assert np.abs(x[keep] - d["cg_x"]).max() <= 1e-8 * np.abs(d["cg_x"]).max()
    eng.close()
