#!/usr/bin/env python3
"""Generates the small golden fixtures under tests/golden/ from the CPU oracle (oracle/pf_oracle.c).

The reference itself cannot be built or imported in this environment (C++ with un-vendored Eigen/TBB/polysolve), and
its own test meshes live in the un-vendored polyfem-data repository, so these vectors pin the ORACLE (a regression
guard for it, and a second, file-based target for the HIP path); the oracle in turn is pinned by the reference's
self-contained known-answer tests (tests/test_oracle_kat.py).  Usage: python tests/tools/make_golden.py [--force]   (without --force only missing fixtures are written)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pf_oracle as O  # noqa: E402
from polyfem_amd import mesh as M  # noqa: E402

CASES = {
    "p1_n2_laplacian": (lambda: M.cube_tets(2, 1), "Laplacian"),
    "p2_n2_linear_elasticity": (lambda: M.cube_tets(2, 2), "LinearElasticity"),
    "p2_n2_jitter_neohookean": (lambda: M.cube_tets(2, 2, jitter=0.2, seed=4), "NeoHookean"),
    "p1_n3_neohookean": (lambda: M.cube_tets(3, 1, jitter=0.1, seed=5), "NeoHookean"),
    "q1_n2_linear_elasticity": (lambda: M.cube_hexes(2, 1), "LinearElasticity"),
    "q2_n1_neohookean": (lambda: M.cube_hexes(1, 2), "NeoHookean"),
    "p2_n2_jitter_saint_venant": (lambda: M.cube_tets(2, 2, jitter=0.2, seed=6), "SaintVenant"),
    "q2_n1_saint_venant": (lambda: M.cube_hexes(1, 2), "SaintVenant"),
}
NONLINEAR = {"NeoHookean": O.NEOHOOKEAN, "SaintVenant": O.SAINT_VENANT}


def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    force = "--force" in sys.argv
    for name, (mk, model) in CASES.items():
        if os.path.exists(os.path.join(out_dir, name + ".npz")) and not force:
            continue  # committed fixtures are only rewritten on request
        mesh = mk()
        d = dict(kind=mesh.kind, conn=mesh.conn, gconn=mesh.gconn, xyz=mesh.xyz, n_nodes=mesh.n_nodes, model=model)
        if model in NONLINEAR:
            mid = NONLINEAR[model]
            lam, mu = O.lame(2e4, 0.3)
            x = M.seeded_displacement(mesh, 3, 2e-3 if model == "NeoHookean" else 5e-2)
            outer, inner, vals = O.assemble_hessian(mesh, mid, lam, mu, x)
            d.update(x=x, energy=O.assemble_energy(mesh, mid, lam, mu, x), gradient=O.assemble_gradient(mesh, mid, lam, mu, x))
        else:
            lam, mu = O.lame(1e5, 0.3)
            mid = O.LAPLACIAN if model == "Laplacian" else O.LINEAR_ELASTICITY
            outer, inner, vals = O.assemble_linear(mesh, mid, lam, mu)
        bs = 1 if model == "Laplacian" else 3
        bd = M.boundary_dofs(mesh, bs, "x0")
        red = O.full_to_reduced_matrix((outer, inner, vals), bd)
        b = np.random.default_rng(9).standard_normal(red[0].size - 1)
        xs, it, err = O.cg(red, b, jacobi=True, max_iter=1000, tol=1e-12)
        d.update(lam=lam, mu=mu, outer=outer, inner=inner, values=vals, boundary=bd, cg_b=b, cg_x=xs, cg_iters=it, cg_err=err)
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **d)
        print(name, "nnz", inner.size, "cg iters", it)


if __name__ == "__main__":
    main()
