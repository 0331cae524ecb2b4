#!/usr/bin/env python3
"""Generates tests/golden/sympy/neohookean_sympy.npz: energy density, first and second derivatives of the NeoHookean energy

    psi(F) = mu/2 (tr(F^T F) - 3 - 2 ln J) + lambda/2 ln^2 J ,  J = det F

(the density NeoHookeanElasticity::compute_energy_aux integrates, src/polyfem/assembler/NeoHookeanElasticity.cpp:250-331)
at a few seeded deformation gradients.  The derivatives come from sympy's symbolic differentiation of psi -- not from the
closed forms the reference, the oracle or the HIP kernels use (:364-565) -- and are evaluated with 50 digits, so the
fixture is an independent statement of what gradient and Hessian have to be."""
import os

import mpmath
import numpy as np
import sympy as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    f = sp.symbols("f0:9", real=True)
    lam, mu = sp.symbols("lambda mu", positive=True)
    F = sp.Matrix(3, 3, f)
    J = F.det()
    psi = mu / 2 * ((F.T * F).trace() - 3 - 2 * sp.log(J)) + lam / 2 * sp.log(J) ** 2
    grad = [sp.diff(psi, v) for v in f]
    hess = [[sp.diff(g, v) for v in f] for g in grad]
    args = list(f) + [lam, mu]
    f_psi = sp.lambdify(args, psi, "mpmath")
    f_grad = sp.lambdify(args, grad, "mpmath")
    f_hess = sp.lambdify(args, hess, "mpmath")
    mpmath.mp.dps = 50
    rng = np.random.default_rng(20240611)
    E, nu = 2e4, 0.3  # tests/test_form_derivatives.cpp:37-39
    lam_v, mu_v = E * nu / ((1 + nu) * (1 - 2 * nu)), E / (2 * (1 + nu))
    Fs, psis, grads, hesss = [], [], [], []
    for k in range(6):
        amp = [0.0, 1e-3, 0.05, 0.2, 0.4, 0.6][k]
        Fk = np.eye(3) + amp * rng.uniform(-1, 1, (3, 3))
        assert np.linalg.det(Fk) > 0.1
        a = [mpmath.mpf(float(v)) for v in Fk.reshape(-1)] + [mpmath.mpf(lam_v), mpmath.mpf(mu_v)]
        Fs.append(Fk)
        psis.append(float(f_psi(*a)))
        grads.append(np.array([float(v) for v in f_grad(*a)]))
        hesss.append(np.array([[float(v) for v in row] for row in f_hess(*a)]))
    out = os.path.join(ROOT, "tests", "golden", "sympy", "neohookean_sympy.npz")
    np.savez(out, F=np.array(Fs), lam=lam_v, mu=mu_v, psi=np.array(psis), dpsi=np.array(grads), d2psi=np.array(hesss))
    print("wrote", out)
    main_2d()


def main_2d():
    """The same for the 2-D density (size() == 2: tr(F^T F) - 2, 2 x 2 determinant; the dim == 2 branches at
    NeoHookeanElasticity.cpp:410-417, 501-512), with lambda from the plane formula of utils/ElasticityUtils.cpp:208:
    tests/golden/sympy/neohookean_sympy_2d.npz."""
    f = sp.symbols("f0:4", real=True)
    lam, mu = sp.symbols("lambda mu", positive=True)
    F = sp.Matrix(2, 2, f)
    J = F.det()
    psi = mu / 2 * ((F.T * F).trace() - 2 - 2 * sp.log(J)) + lam / 2 * sp.log(J) ** 2
    grad = [sp.diff(psi, v) for v in f]
    hess = [[sp.diff(g, v) for v in f] for g in grad]
    args = list(f) + [lam, mu]
    f_psi, f_grad, f_hess = (sp.lambdify(args, q, "mpmath") for q in (psi, grad, hess))
    mpmath.mp.dps = 50
    rng = np.random.default_rng(20241112)
    E, nu = 2e4, 0.3
    lam_v, mu_v = nu * E / (1 - nu * nu), E / (2 * (1 + nu))
    Fs, psis, grads, hesss = [], [], [], []
    for amp in [0.0, 1e-3, 0.05, 0.2, 0.4, 0.6]:
        Fk = np.eye(2) + amp * rng.uniform(-1, 1, (2, 2))
        assert np.linalg.det(Fk) > 0.1
        a = [mpmath.mpf(float(v)) for v in Fk.reshape(-1)] + [mpmath.mpf(lam_v), mpmath.mpf(mu_v)]
        Fs.append(Fk)
        psis.append(float(f_psi(*a)))
        grads.append(np.array([float(v) for v in f_grad(*a)]))
        hesss.append(np.array([[float(v) for v in row] for row in f_hess(*a)]))
    out = os.path.join(ROOT, "tests", "golden", "sympy", "neohookean_sympy_2d.npz")
    np.savez(out, F=np.array(Fs), lam=lam_v, mu=mu_v, psi=np.array(psis), dpsi=np.array(grads), d2psi=np.array(hesss))
    print("wrote", out)


if __name__ == "__main__":
    main()
