"""ctypes binding of the CPU oracle (oracle/pf_oracle.c).  TEST INFRASTRUCTURE ONLY: imported by tests/,
bench.py's cpu_baseline leg and __graft_entry__.smoke() as the checker, never by the product package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpf_oracle.so")

P1TET, P2TET, Q1HEX, Q2HEX, P1TRI, P2TRI = 0, 1, 2, 3, 4, 5
LAPLACIAN, LINEAR_ELASTICITY, NEOHOOKEAN, SAINT_VENANT, HOOKE_LINEAR, MULTI_MODEL = 0, 1, 2, 3, 4, 5
MAX_PTS, MAX_LOC = 125, 27


class Quad(C.Structure):
    _fields_ = [("n_pts", C.c_int), ("pts", C.c_double * 3 * MAX_PTS), ("w", C.c_double * MAX_PTS)]


class Vals(C.Structure):
    _fields_ = [
        ("n_pts", C.c_int), ("n_loc", C.c_int), ("dim", C.c_int),
        ("val", C.c_double * MAX_PTS * MAX_LOC),
        ("grad", C.c_double * 3 * MAX_PTS * MAX_LOC),
        ("grad_t_m", C.c_double * 3 * MAX_PTS * MAX_LOC),
        ("det", C.c_double * MAX_PTS),
        ("jac_it", C.c_double * 9 * MAX_PTS),
        ("quad", Quad),
    ]


class MeshS(C.Structure):
    _fields_ = [
        ("kind", C.c_int), ("gkind", C.c_int), ("n_el", C.c_int64), ("n_nodes", C.c_int64), ("n_gnodes", C.c_int64),
        ("conn", C.POINTER(C.c_int32)), ("gconn", C.POINTER(C.c_int32)), ("xyz", C.POINTER(C.c_double)),
        ("quad_order", C.c_int),
    ]


class Material(C.Structure):
    _fields_ = [("model", C.c_int), ("dim", C.c_int), ("lambda_", C.c_double), ("mu", C.c_double),
                ("lambda_e", C.POINTER(C.c_double)), ("mu_e", C.POINTER(C.c_double)), ("tensor", C.POINTER(C.c_double)),
                ("per_point", C.c_int), ("model_e", C.POINTER(C.c_int32))]


def make_material(model: int, lam, mu, tensor=None):
    """Uniform Lame parameters (scalars), one value per element (1-D arrays) or per quadrature point (2-D arrays, n_el x
    n_pts); for HOOKE_LINEAR `lam` is the symmetric
    6 x 6 elasticity tensor (and `mu` is ignored); SAINT_VENANT takes such a tensor as `tensor`
    (SaintVenantElasticity::add_multimaterial, assembler/SaintVenantElasticity.cpp:42-69).  Returns (Material, keep-alive)."""
    if tensor is None and mu is None and np.ndim(model) == 0 and model == SAINT_VENANT:
        tensor = lam  # (SAINT_VENANT, 6 x 6 tensor, None): the positional form the assemble_* wrappers pass through
    if tensor is not None:
        assert model == SAINT_VENANT
        t = _voigt_tensor(tensor)
        return Material(model, 3 if t.shape[0] == 6 else 2, 0.0, 0.0, None, None, _dp(t), 0, None), t
    if np.ndim(model) == 1:  # MultiModel: one model id per element
        me = np.ascontiguousarray(model, dtype=np.int32)
        assert np.isin(me, (LINEAR_ELASTICITY, NEOHOOKEAN, SAINT_VENANT)).all()
        mat, keep = make_material(MULTI_MODEL, lam, mu)
        mat.model_e = me.ctypes.data_as(C.POINTER(C.c_int32))
        return mat, (keep, me)
    if model == HOOKE_LINEAR:
        t = _voigt_tensor(lam)
        return Material(model, 3 if t.shape[0] == 6 else 2, 0.0, 0.0, None, None, _dp(t), 0, None), t
    if np.ndim(lam) == 0:
        return Material(model, 3, float(lam), float(mu), None, None, None, 0, None), None
    la = np.ascontiguousarray(lam, dtype=np.float64)
    ma = np.ascontiguousarray(mu, dtype=np.float64)
    # (n_el,) = one value per element; (n_el, n_pts) = one value per quadrature point
    per_point = int(la.shape[1]) if la.ndim == 2 else 0
    assert la.shape == ma.shape
    return Material(model, 3, float(la.flat[0]), float(ma.flat[0]), _dp(la), _dp(ma), None, per_point, None), (la, ma)


def _voigt_tensor(tensor):
    """The symmetric elasticity tensor in Voigt form: 6 x 6 (xx yy zz yz xz xy) in 3-D, 3 x 3 (xx yy xy) in 2-D
    (ElasticityTensor::resize, utils/ElasticityUtils.cpp:264-274)."""
    t = np.ascontiguousarray(tensor, dtype=np.float64)
    n = {36: 6, 9: 3}[t.size]
    t = t.reshape(n, n)
    assert np.array_equal(t, t.T), "the elasticity tensor is symmetric"
    return t


def kind_dim(kind: int) -> int:
    return 2 if kind in (P1TRI, P2TRI) else 3


def _material_for(om, model, lam, mu):
    """make_material with the problem dimension of the mesh (2 for triangles)."""
    mat, keep = make_material(model, lam, mu)
    if mat.tensor:
        assert mat.dim == kind_dim(om.s.kind), "a 2-D problem takes a 3 x 3 elasticity tensor, a 3-D problem a 6 x 6 one"
    mat.dim = kind_dim(om.s.kind)
    return mat, keep


def elasticity_tensor(lam=None, mu=None, entries=None, dim=3):
    """ElasticityTensor (utils/ElasticityUtils.cpp): set_from_lambda_mu (:372-413) or set_from_entries with the 21
    upper-triangular entries row by row (:298-370), or with the 9 orthotropic constants Ex Ey Ez nuYX nuZX nuZY muYZ muZX
    muXY, which the reference stores as the upper triangle written at :439-455; returns the full symmetric 6 x 6 matrix
    that ElasticityTensor::operator() (:276-296) reads.  dim = 2: the 3 x 3 tensor from (lambda, mu) (:374-384), from 6
    upper-triangular entries (:313-322) or from the 4 orthotropic constants Ex Ey nuYX muXY (:457-466)."""
    if dim == 2:
        t = np.zeros((3, 3))
        if entries is not None:
            entries = np.asarray(entries, dtype=np.float64)
            if entries.size == 4:
                ex, ey, nyx, mxy = entries
                t[0, :2] = 1.0 / ex, -nyx / ey
                t[1, 1] = 1.0 / ey
                t[2, 2] = 1.0 / mxy
            else:
                assert entries.size == 6
                t[np.triu_indices(3)] = entries
            return t + np.triu(t, 1).T
        t[:2, :2] = lam
        t[0, 0] = t[1, 1] = 2 * mu + lam
        t[2, 2] = mu
        return t
    t = np.zeros((6, 6))
    if entries is not None:
        entries = np.asarray(entries, dtype=np.float64)
        if entries.size == 9:
            ex, ey, ez, nyx, nzx, nzy, myz, mzx, mxy = entries
            t[0, :3] = 1.0 / ex, -nyx / ey, -nzx / ez
            t[1, 1:3] = 1.0 / ey, -nzy / ez
            t[2, 2] = 1.0 / ez
            t[3, 3], t[4, 4], t[5, 5] = 1.0 / myz, 1.0 / mzx, 1.0 / mxy
            return t + np.triu(t, 1).T
        assert entries.size == 21
        t[np.triu_indices(6)] = entries
        t = t + np.triu(t, 1).T
        return t
    t[:3, :3] = lam
    t[np.arange(3), np.arange(3)] = 2 * mu + lam
    t[np.arange(3, 6), np.arange(3, 6)] = mu
    return t


class Csc(C.Structure):
    _fields_ = [("n", C.c_int32), ("nnz", C.c_int64), ("outer", C.POINTER(C.c_int32)),
                ("inner", C.POINTER(C.c_int32)), ("values", C.POINTER(C.c_double))]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "pf_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.pfo_assemble_energy.restype = C.c_double
        L.pfo_determinant3.restype = C.c_double
        L.pfo_lambda_from_E_nu.restype = C.c_double
        L.pfo_mu_from_E_nu.restype = C.c_double
        L.pfo_lambda_from_E_nu.argtypes = [C.c_double, C.c_double]
        L.pfo_lambda_from_E_nu_2d.restype = C.c_double
        L.pfo_lambda_from_E_nu_2d.argtypes = [C.c_double, C.c_double]
        L.pfo_mu_from_E_nu.argtypes = [C.c_double, C.c_double]
        L.pfo_local_neohookean_energy.restype = C.c_double
        L.pfo_cache_create.restype = C.c_void_p
        L.pfo_cache_clone.restype = C.c_void_p
        L.pfo_cache_mapping_size.restype = C.c_int64
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def lame(E: float, nu: float, dim: int = 3):
    L = lib()
    return (L.pfo_lambda_from_E_nu(E, nu) if dim == 3 else L.pfo_lambda_from_E_nu_2d(E, nu)), L.pfo_mu_from_E_nu(E, nu)


def quadrature(kind: int, quad_order: int = -1):
    q = Quad()
    rc = lib().pfo_element_quadrature(kind, quad_order, C.byref(q))
    assert rc == 0
    n = q.n_pts
    pts = np.array([[q.pts[k][c] for c in range(3)] for k in range(n)])
    w = np.array([q.w[k] for k in range(n)])
    return pts, w


def tet_quadrature(order: int):
    q = Quad()
    assert lib().pfo_tet_quadrature(order, C.byref(q)) == 0
    n = q.n_pts
    return np.array([[q.pts[k][c] for c in range(3)] for k in range(n)]), np.array([q.w[k] for k in range(n)])


def tri_quadrature(order: int):
    q = Quad()
    assert lib().pfo_tri_quadrature(order, C.byref(q)) == 0
    n = q.n_pts
    return np.array([[q.pts[k][c] for c in range(2)] for k in range(n)]), np.array([q.w[k] for k in range(n)])


def line_quadrature(order: int):
    q = Quad()
    assert lib().pfo_line_quadrature(order, C.byref(q)) == 0
    n = q.n_pts
    return np.array([q.pts[k][0] for k in range(n)]), np.array([q.w[k] for k in range(n)])


def basis_value(kind: int, li: int, pt) -> float:
    p = (C.c_double * 3)(*pt)
    v = C.c_double()
    lib().pfo_basis_value(kind, li, p, C.byref(v))
    return v.value


def basis_grad(kind: int, li: int, pt):
    p = (C.c_double * 3)(*pt)
    g = (C.c_double * 3)()
    lib().pfo_basis_grad(kind, li, p, g)
    return np.array(list(g))


def q2_tensor_basis(li: int, pt):
    p = np.ascontiguousarray(pt, dtype=np.float64)
    v = C.c_double()
    g = np.zeros(3)
    lib().pfo_q2_tensor_basis(int(li), _dp(p), C.byref(v), _dp(g))
    return v.value, g


def basis_nodes(kind: int):
    n = lib().pfo_n_loc(kind)
    out = np.zeros((n, 3))
    lib().pfo_basis_nodes(kind, _dp(out))
    return out


def determinant3(m) -> float:
    a = np.ascontiguousarray(np.asarray(m, dtype=np.float64).reshape(9))
    return lib().pfo_determinant3(_dp(a))


class OMesh:
    """Keeps the numpy arrays alive next to the C struct."""

    def __init__(self, mesh):
        self.conn = np.ascontiguousarray(mesh.conn, dtype=np.int32)
        self.gconn = np.ascontiguousarray(mesh.gconn, dtype=np.int32)
        self.xyz = np.ascontiguousarray(mesh.xyz, dtype=np.float64)
        self.n_nodes = int(mesh.n_nodes)
        self.s = MeshS(mesh.kind, mesh.gkind, self.conn.shape[0], mesh.n_nodes, self.xyz.shape[0],
                       _ip(self.conn), _ip(self.gconn), _dp(self.xyz), mesh.quad_order)


def _csc_to_numpy(m: Csc, free: bool = True):
    n, nnz = m.n, m.nnz
    outer = np.ctypeslib.as_array(m.outer, shape=(n + 1,)).copy()
    inner = np.ctypeslib.as_array(m.inner, shape=(max(nnz, 1),))[:nnz].copy()
    values = np.ctypeslib.as_array(m.values, shape=(max(nnz, 1),))[:nnz].copy()
    if free:
        lib().pfo_csc_free(C.byref(m))
    return outer, inner, values


def _numpy_to_csc(outer, inner, values):
    outer = np.ascontiguousarray(outer, dtype=np.int32)
    inner = np.ascontiguousarray(inner, dtype=np.int32)
    values = np.ascontiguousarray(values, dtype=np.float64)
    m = Csc(outer.size - 1, inner.size, _ip(outer), _ip(inner), _dp(values))
    return m, (outer, inner, values)


def compute_vals(mesh, e: int, quad_order: int = -1):
    om = mesh if isinstance(mesh, OMesh) else OMesh(mesh)
    q = Quad()
    assert lib().pfo_element_quadrature(om.s.kind, quad_order, C.byref(q)) == 0
    v = Vals()
    lib().pfo_compute_vals(C.byref(om.s), C.c_int64(e), C.byref(q), C.byref(v))
    n_loc, m = v.n_loc, v.n_pts
    val = np.ctypeslib.as_array(v.val).reshape(MAX_LOC, MAX_PTS)[:n_loc, :m].copy()
    grad = np.ctypeslib.as_array(v.grad).reshape(MAX_LOC, MAX_PTS, 3)[:n_loc, :m].copy()
    gtm = np.ctypeslib.as_array(v.grad_t_m).reshape(MAX_LOC, MAX_PTS, 3)[:n_loc, :m].copy()
    det = np.ctypeslib.as_array(v.det)[:m].copy()
    jac_it = np.ctypeslib.as_array(v.jac_it).reshape(MAX_PTS, 3, 3)[:m].copy()
    w = np.array([q.w[k] for k in range(m)])
    return dict(val=val, grad=grad, grad_t_m=gtm, det=det, jac_it=jac_it, w=w)


def assemble_linear(mesh, model: int, lam: float, mu: float, n_threads: int = 1):
    om = mesh if isinstance(mesh, OMesh) else OMesh(mesh)
    mat, _keep_mat = _material_for(om, model, lam, mu)
    out = Csc()
    rc = lib().pfo_assemble_linear(C.byref(om.s), C.byref(mat), n_threads, C.byref(out))
    assert rc == 0
    return _csc_to_numpy(out)


def assemble_energy(mesh, model: int, lam: float, mu: float, x, n_threads: int = 1) -> float:
    om = mesh if isinstance(mesh, OMesh) else OMesh(mesh)
    mat, _keep_mat = _material_for(om, model, lam, mu)
    x = np.ascontiguousarray(x, dtype=np.float64)
    return lib().pfo_assemble_energy(C.byref(om.s), C.byref(mat), _dp(x), n_threads)


def assemble_gradient(mesh, model: int, lam: float, mu: float, x, n_threads: int = 1):
    om = mesh if isinstance(mesh, OMesh) else OMesh(mesh)
    mat, _keep_mat = _material_for(om, model, lam, mu)
    x = np.ascontiguousarray(x, dtype=np.float64)
    rhs = np.zeros(om.n_nodes * kind_dim(om.s.kind))
    rc = lib().pfo_assemble_gradient(C.byref(om.s), C.byref(mat), _dp(x), n_threads, _dp(rhs))
    assert rc == 0
    return rhs


def assemble_hessian(mesh, model: int, lam: float, mu: float, x, n_threads: int = 1):
    om = mesh if isinstance(mesh, OMesh) else OMesh(mesh)
    mat, _keep_mat = _material_for(om, model, lam, mu)
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = Csc()
    rc = lib().pfo_assemble_hessian(C.byref(om.s), C.byref(mat), _dp(x), n_threads, C.byref(out))
    assert rc == 0
    return _csc_to_numpy(out)


def assemble_hessian_psd(mesh, model: int, lam, mu, x, n_threads: int = 1):
    """NLAssembler::assemble_hessian with project_to_psd = true."""
    om = mesh if isinstance(mesh, OMesh) else OMesh(mesh)
    mat, _keep_mat = _material_for(om, model, lam, mu)
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = Csc()
    rc = lib().pfo_assemble_hessian_psd(C.byref(om.s), C.byref(mat), _dp(x), int(n_threads), C.byref(out))
    assert rc == 0
    return _csc_to_numpy(out)


def project_to_psd(a):
    m = np.array(a, dtype=np.float64, copy=True, order="C")
    assert m.ndim == 2 and m.shape[0] == m.shape[1]
    lib().pfo_project_to_psd(_dp(m), int(m.shape[0]))
    return m


class Cache:
    """utils::SpareMatrixCache restatement."""

    def __init__(self, size: int = 0, _ptr=None):
        self.p = C.c_void_p(_ptr if _ptr is not None else lib().pfo_cache_create(C.c_int64(size)))

    def clone(self):
        return Cache(_ptr=lib().pfo_cache_clone(self.p))

    def add_value(self, e, i, j, v):
        lib().pfo_cache_add_value(self.p, int(e), int(i), int(j), C.c_double(v))

    def prune(self):
        lib().pfo_cache_prune(self.p)

    def get_matrix(self, compute_mapping: bool = True):
        out = Csc()
        rc = lib().pfo_cache_get_matrix(self.p, int(compute_mapping), C.byref(out))
        assert rc == 0
        return _csc_to_numpy(out)

    def mapping_size(self) -> int:
        return lib().pfo_cache_mapping_size(self.p)

    def __del__(self):
        try:
            lib().pfo_cache_destroy(self.p)
        except Exception:
            pass


def assemble_hessian_cached(mesh, model, lam, mu, x, cache: Cache):
    om = mesh if isinstance(mesh, OMesh) else OMesh(mesh)
    mat, _keep_mat = _material_for(om, model, lam, mu)
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = Csc()
    rc = lib().pfo_assemble_hessian_cached(C.byref(om.s), C.byref(mat), _dp(x), cache.p, C.byref(out))
    assert rc == 0
    return _csc_to_numpy(out)


def full_to_reduced_matrix(csc, removed_vars):
    m, keep = _numpy_to_csc(*csc)
    rv = np.ascontiguousarray(removed_vars, dtype=np.int32)
    out = Csc()
    rc = lib().pfo_full_to_reduced_matrix(C.byref(m), _ip(rv), C.c_int64(rv.size), C.byref(out))
    assert rc == 0
    return _csc_to_numpy(out)


def spmv(csc, x):
    m, keep = _numpy_to_csc(*csc)
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.zeros_like(x)
    lib().pfo_spmv(C.byref(m), _dp(x), _dp(y))
    return y


def cg(csc, b, x0=None, jacobi: bool = True, max_iter: int = 1000, tol: float = 1e-12):
    m, keep = _numpy_to_csc(*csc)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros_like(b) if x0 is None else np.array(x0, dtype=np.float64, copy=True)
    iters, err = C.c_int(), C.c_double()
    lib().pfo_cg(C.byref(m), _dp(b), _dp(x), int(jacobi), int(max_iter), C.c_double(tol), C.byref(iters), C.byref(err))
    return x, iters.value, err.value


def dirichlet_modify(csc, b, boundary_nodes):
    outer, inner, values = (np.array(a, copy=True) for a in csc)
    m, keep = _numpy_to_csc(outer, inner, values)
    b = np.array(b, dtype=np.float64, copy=True)
    bn = np.ascontiguousarray(boundary_nodes, dtype=np.int32)
    lib().pfo_dirichlet_modify(C.byref(m), _dp(b), _ip(bn), C.c_int64(bn.size))
    return keep, b


class NewtonParams(C.Structure):
    _fields_ = [("f_delta", C.c_double), ("grad_norm", C.c_double), ("first_grad_norm_tol", C.c_double),
                ("use_grad_norm_tol", C.c_double), ("max_iterations", C.c_int), ("use_grad_norm", C.c_int),
                ("relative_gradient", C.c_int), ("line_search", C.c_int), ("cg_precond_jacobi", C.c_int),
                ("cg_max_iter", C.c_int), ("cg_tol", C.c_double)]


class NewtonInfo(C.Structure):
    _fields_ = [("status", C.c_int), ("iterations", C.c_int), ("line_search_iterations", C.c_int),
                ("descent_strategy", C.c_int), ("cg_iterations", C.c_int64), ("energy", C.c_double),
                ("grad_norm", C.c_double)]


NEWTON_STATUS = ["continue", "iteration_limit", "f_delta", "grad_norm", "not_started", "failed"]


def newton_solve(mesh, model, lam, mu, rhs, boundary_nodes, x0, n_threads: int = 1, mass=None, x_tilde=None,
                 weight: float = 1.0, **params):
    """cppoptlib NonlinearSolver::minimize with SparseNewtonDescentSolver + BacktrackingLineSearch, restated.
    Returns (x, info dict).  `params` override the reference's defaults (f_delta, grad_norm, max_iterations, ...).
    With `mass` (CSC tuple) and `x_tilde` the objective is weight * (E - rhs.x) + 1/2 (x - x_tilde)^T M (x - x_tilde)
    (one implicit time step: InertiaForm + weighted ElasticForm / BodyForm)."""
    om = mesh if isinstance(mesh, OMesh) else OMesh(mesh)
    mat, _keep_mat = _material_for(om, model, lam, mu)
    p = NewtonParams()
    lib().pfo_newton_default_params(C.byref(p))
    for k, v in params.items():
        setattr(p, k, v)
    x = np.array(x0, dtype=np.float64, copy=True)
    bn = np.ascontiguousarray(boundary_nodes, dtype=np.int32)
    r = None if rhs is None else np.ascontiguousarray(rhs, dtype=np.float64)
    info = NewtonInfo()
    if mass is None:
        rc = lib().pfo_newton_solve(C.byref(om.s), C.byref(mat), _dp(r) if r is not None else None, _ip(bn), C.c_int64(bn.size),
                                    _dp(x), C.byref(p), int(n_threads), C.byref(info))
    else:
        m, _keep = _numpy_to_csc(*mass)
        xt = np.ascontiguousarray(x_tilde, dtype=np.float64)
        rc = lib().pfo_newton_solve_inertia(C.byref(om.s), C.byref(mat), _dp(r) if r is not None else None, _ip(bn),
                                            C.c_int64(bn.size), _dp(x), C.byref(p), int(n_threads), C.byref(info),
                                            C.byref(m), _dp(xt), C.c_double(weight))
    assert rc == 0
    return x, {f[0]: getattr(info, f[0]) for f in NewtonInfo._fields_}


def implicit_euler_step(mesh, model, lam, mu, rhs, boundary_nodes, mass, dt, x_prev, v_prev, n_threads: int = 1, **params):
    """One step of the reference's transient nonlinear solve with the ImplicitEuler integrator
    (time_integrator/ImplicitEuler.cpp:5-33): x_tilde = x_prev + dt v_prev, Newton from x_prev on the weighted objective
    (acceleration_scaling = dt^2), then v = (x - x_prev)/dt, a = (v - v_prev)/dt.  Returns (x, v, a, info)."""
    x_prev = np.asarray(x_prev, dtype=np.float64)
    v_prev = np.asarray(v_prev, dtype=np.float64)
    x_tilde = x_prev + dt * v_prev
    x, info = newton_solve(mesh, model, lam, mu, rhs, boundary_nodes, x_prev, n_threads, mass=mass, x_tilde=x_tilde,
                           weight=dt * dt, **params)
    v = (x - x_prev) / dt
    a = (v - v_prev) / dt
    return x, v, a, info


def assemble_mass(mesh, size: int, density=1.0, n_threads: int = 1):
    """Assembler<Mass>::assemble(is_mass=true).  density: scalar or one value per element."""
    om = mesh if isinstance(mesh, OMesh) else OMesh(mesh)
    rho_e = None if np.ndim(density) == 0 else np.ascontiguousarray(density, dtype=np.float64)
    out = Csc()
    if rho_e is not None and rho_e.ndim == 2:  # one density per (element, mass quadrature point)
        assert rho_e.shape[0] == om.s.n_el
        rc = lib().pfo_assemble_mass_pointwise(C.byref(om.s), int(size), _dp(rho_e), int(n_threads), C.byref(out))
        assert rc == 0
        return _csc_to_numpy(out)
    rc = lib().pfo_assemble_mass(C.byref(om.s), int(size), C.c_double(float(density) if rho_e is None else float(rho_e[0])),
                                 _dp(rho_e) if rho_e is not None else None, int(n_threads), C.byref(out))
    assert rc == 0
    return _csc_to_numpy(out)


def assemble_rhs(mesh, size: int, force, density=1.0):
    """RhsAssembler::assemble for a constant body force."""
    om = mesh if isinstance(mesh, OMesh) else OMesh(mesh)
    rho_e = None if np.ndim(density) == 0 else np.ascontiguousarray(density, dtype=np.float64)
    f = np.ascontiguousarray(force, dtype=np.float64)
    assert f.size == size
    rhs = np.zeros(om.s.n_nodes * size)
    if rho_e is not None and rho_e.ndim == 2:  # one density per (element, quadrature point)
        assert lib().pfo_assemble_rhs_density_pointwise(C.byref(om.s), int(size), _dp(f), None, _dp(rho_e), _dp(rhs)) == 0
        return rhs
    rc = lib().pfo_assemble_rhs(C.byref(om.s), int(size), _dp(f), C.c_double(float(density) if rho_e is None else float(rho_e[0])),
                                _dp(rho_e) if rho_e is not None else None, _dp(rhs))
    assert rc == 0
    return rhs


def quadrature_points(mesh):
    """Physical positions of the stiffness-rule quadrature points, shape (n_el, n_pts, 3)."""
    om = mesh if isinstance(mesh, OMesh) else OMesh(mesh)
    buf = np.zeros(om.s.n_el * 125 * 3)
    lib().pfo_quadrature_points.restype = C.c_int
    n_pts = lib().pfo_quadrature_points(C.byref(om.s), _dp(buf))
    assert n_pts > 0
    d = kind_dim(om.s.kind)
    return buf[: om.s.n_el * n_pts * d].reshape(om.s.n_el, n_pts, d).copy()


def mass_quadrature_points(mesh):
    """Physical positions of the points of the mass quadrature rule, shape (n_el, n_pts, 3)."""
    om = mesh if isinstance(mesh, OMesh) else OMesh(mesh)
    buf = np.zeros(om.s.n_el * 125 * 3)
    lib().pfo_mass_quadrature_points.restype = C.c_int
    n_pts = lib().pfo_mass_quadrature_points(C.byref(om.s), _dp(buf))
    assert n_pts > 0
    d = kind_dim(om.s.kind)
    return buf[: om.s.n_el * n_pts * d].reshape(om.s.n_el, n_pts, d).copy()


def assemble_rhs_pointwise(mesh, size: int, force_pts, density=1.0):
    """RhsAssembler::assemble with the force density given per quadrature point: force_pts (n_el, n_pts, size)."""
    om = mesh if isinstance(mesh, OMesh) else OMesh(mesh)
    rho_e = None if np.ndim(density) == 0 else np.ascontiguousarray(density, dtype=np.float64)
    f = np.ascontiguousarray(force_pts, dtype=np.float64)
    assert f.shape[0] == om.s.n_el and f.shape[2] == size
    rhs = np.zeros(om.s.n_nodes * size)
    if rho_e is not None and rho_e.ndim == 2:
        assert lib().pfo_assemble_rhs_density_pointwise(C.byref(om.s), int(size), None, _dp(f), _dp(rho_e), _dp(rhs)) == 0
        return rhs
    rc = lib().pfo_assemble_rhs_pointwise(C.byref(om.s), int(size), _dp(f), C.c_double(float(density) if rho_e is None else float(rho_e[0])),
                                          _dp(rho_e) if rho_e is not None else None, _dp(rhs))
    assert rc == 0
    return rhs


def mass_quadrature(kind: int):
    q = Quad()
    assert lib().pfo_mass_quadrature(kind, C.byref(q)) == 0
    n = q.n_pts
    return np.ctypeslib.as_array(q.pts).reshape(MAX_PTS, 3)[:n].copy(), np.ctypeslib.as_array(q.w)[:n].copy()
