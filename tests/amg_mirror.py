"""Test-only scipy mirror of the aggregation-AMG preconditioner of polyfem_amd/csrc/pfx_amg.hip.

Given the aggregates the library chose (pfx_amg_get_aggregates) and its spectral-radius estimates, it rebuilds the
hierarchy with sparse matrices -- prolongation with the six rigid-body modes per aggregate, Galerkin products,
Chebyshev relaxation, dense coarsest solve -- so that the device kernels (Galerkin lists, transfers, relaxation, cycle)
can be checked against an independent statement of the same algorithm.  Nothing in polyfem_amd/ imports it."""
import numpy as np
import scipy.sparse as sp


def node_positions(mesh):
    """Physical position of every basis node (polyfem_amd.mesh.node_positions)."""
    from polyfem_amd import mesh as M
    return M.node_positions(mesh)


def skew(d):
    return np.array([[0, -d[2], d[1]], [d[2], 0, -d[0]], [-d[1], d[0], 0]])


def prolongation(agg, na, pos, level0, fixed=None, scalar=False, plane=False):
    """Rows: 3 dofs per vertex (level 0) or 6 (translation, rotation); columns: 6 per aggregate.  Scalar problems: one dof
    per vertex and per aggregate, piecewise constant; plane: the three rigid-body modes of 2-D elasticity.  Returns P, centres."""
    nv = agg.size
    ok = agg >= 0
    cnt = np.bincount(agg[ok], minlength=na).astype(float)
    c = np.stack([np.bincount(agg[ok], weights=pos[ok, k], minlength=na) / cnt for k in range(3)], axis=1)
    if scalar:
        v = np.nonzero(ok)[0]
        return sp.csr_matrix((np.ones(v.size), (v, agg[v])), shape=(nv, na)), c
    if plane:
        # 2-D elasticity: rows 2 dofs per vertex (level 0) or 3 (translations, rotation); columns 3 per aggregate
        rows, cols, vals = [], [], []
        per = 2 if level0 else 3
        for v in np.nonzero(ok)[0]:
            a = agg[v]
            d = pos[v] - c[a]
            ent = [(0, 0, 1.0), (1, 1, 1.0), (0, 2, -d[1]), (1, 2, d[0])] + ([] if level0 else [(2, 2, 1.0)])
            for r, q, val in ent:
                if val != 0:
                    rows.append(per * v + r), cols.append(3 * a + q), vals.append(val)
        P = sp.csr_matrix((vals, (rows, cols)), shape=(per * nv, 3 * na))
        if fixed is not None:
            P = sp.diags((~fixed).astype(float)) @ P
        return P.tocsr(), c
    rows, cols, vals = [], [], []
    per = 3 if level0 else 6
    for v in np.nonzero(ok)[0]:
        a = agg[v]
        S = -skew(pos[v] - c[a])
        blocks = [(0, 0, np.eye(3)), (0, 3, S)] if level0 else [(0, 0, np.eye(3)), (0, 3, S), (3, 3, np.eye(3))]
        for r0, c0, B in blocks:
            for r in range(3):
                for q in range(3):
                    if B[r, q] != 0:
                        rows.append(per * v + r0 + r)
                        cols.append(6 * a + c0 + q)
                        vals.append(B[r, q])
    P = sp.csr_matrix((vals, (rows, cols)), shape=(per * nv, 6 * na))
    if fixed is not None:
        P = sp.diags((~fixed).astype(float)) @ P
    return P.tocsr(), c


class Level:
    pass


def build(A_masked, node_xyz, fixed, aggregates, lambda_max, relax_degree=1, relax_lower=0.1, ncycle=1, coarse_scale=1.6, scalar=False, plane=False):
    """A_masked: level-0 operator with identity rows/columns on the constrained dofs; aggregates[l]: vertex -> aggregate."""
    levels, A, pos = [], A_masked.tocsr(), np.asarray(node_xyz, dtype=float)
    for l, agg in enumerate(aggregates):
        L = Level()
        L.A = A
        d = A.diagonal().copy()
        L.dinv = np.where(d != 0, 1.0 / np.where(d != 0, d, 1.0), 1.0)
        L.lmax = lambda_max[l]
        na = int(agg.max()) + 1
        L.P, centres = prolongation(np.asarray(agg), na, pos, l == 0, fixed if l == 0 else None, scalar, plane)
        levels.append(L)
        A = (L.P.T @ A @ L.P).tocsr()
        pos = centres
    C = Level()
    M = A.toarray()
    dmax = np.abs(np.diag(M)).max()
    for k in range(M.shape[0]):
        if M[k, k] == 0.0:
            M[k, k] = dmax if dmax > 0 else 1.0
        else:
            M[k, k] += 1e-12 * dmax
    C.Ainv = np.linalg.inv(M)
    C.A = A
    levels.append(C)
    for L in levels:
        L.deg, L.lower, L.ncycle, L.omega = relax_degree, relax_lower, ncycle, coarse_scale
    return levels


def relax(L, b, x):
    lmax, lmin = L.lmax, L.lmax * L.lower
    theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
    sigma = theta / delta
    rho = 1.0 / sigma
    r = b.copy() if x is None else b - L.A @ x
    dv = L.dinv * r / theta
    x = dv.copy() if x is None else x + dv
    for _ in range(1, L.deg):
        r = r - L.A @ dv
        rho_new = 1.0 / (2.0 * sigma - rho)
        dv = rho_new * rho * dv + 2.0 * rho_new / delta * (L.dinv * r)
        x = x + dv
        rho = rho_new
    return x


def cycle(levels, l, b):
    L = levels[l]
    if l == len(levels) - 1:
        return L.Ainv @ b
    x = relax(L, b, None)
    rc = L.P.T @ (b - L.A @ x)
    xc = cycle(levels, l + 1, rc)
    if L.ncycle > 1 and l + 2 < len(levels):
        for _ in range(1, L.ncycle):
            xc = xc + cycle(levels, l + 1, rc - levels[l + 1].A @ xc)
    x = x + L.omega * (L.P @ xc)
    return relax(L, b, x)


def pcg(A, b, Minv, tol, max_iter):
    """Eigen's ConjugateGradient with a general preconditioner; returns (x, Eigen's iteration count, relative residual)."""
    x = np.zeros_like(b)
    r = b.copy()
    bb = b @ b
    thr = max(tol * tol * bb, np.finfo(float).tiny)
    rr = r @ r
    if rr < thr:
        return x, 0, np.sqrt(rr / bb)
    p = Minv(r)
    abs_new = r @ p
    i = 0
    while i < max_iter:
        t = A @ p
        alpha = abs_new / (p @ t)
        x += alpha * p
        r -= alpha * t
        rr = r @ r
        if rr < thr:
            break
        z = Minv(r)
        abs_old, abs_new = abs_new, r @ z
        p = z + (abs_new / abs_old) * p
        i += 1
    return x, i, np.sqrt(rr / bb)
