"""ctypes binding of libpolyfem_amd.so (the C ABI of include/polyfem_amd.h) plus a host-side mirror of the
reference's assembler / linear-solver interface for this path.

Names follow the reference:
  * `Assembler.assemble`                                 <- assembler::Assembler<L>::assemble        (Assembler.hpp:32-39)
  * `NLAssembler.assemble / assemble_grad / assemble_hessian`
                                                         <- assembler::NLAssembler<L>::...           (Assembler.hpp:80-113)
  * `LinearSolver.create / set_parameters / analyze_pattern / factorize / solve / get_info`
                                                         <- polysolve::LinearSolver                  (call sites
                                                            solver/SparseNewtonDescentSolver.tpp:128-155)
There is no CPU fallback: if the HIP library cannot be loaded every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PFX_LIBRARY selects another build of the same ABI (e.g. one with host-side sanitizers)
LIB_PATH = os.environ.get("PFX_LIBRARY") or os.path.join(_HERE, "libpolyfem_amd.so")

P1TET, P2TET, Q1HEX, Q2HEX, P1TRI, P2TRI = 0, 1, 2, 3, 4, 5
LAPLACIAN, LINEAR_ELASTICITY, NEOHOOKEAN, SAINT_VENANT, HOOKE_LINEAR, MULTI_MODEL = 0, 1, 2, 3, 4, 5
PRECOND_IDENTITY, PRECOND_JACOBI, PRECOND_AMG = 0, 1, 2
PFX_ERR_ARG, PFX_ERR_HIP, PFX_ERR_UNSUPPORTED, PFX_ERR_NUMERIC, PFX_ERR_SIZE = -1, -2, -3, -4, -5
MODEL_BY_NAME = {"Laplacian": LAPLACIAN, "LinearElasticity": LINEAR_ELASTICITY, "NeoHookean": NEOHOOKEAN,
                 "SaintVenant": SAINT_VENANT}

EXPORTED_SYMBOLS = [
    "pfx_create", "pfx_destroy", "pfx_last_error", "pfx_set_stream", "pfx_synchronize", "pfx_set_mesh",
    "pfx_set_material", "pfx_build_pattern", "pfx_get_pattern", "pfx_get_values", "pfx_assemble_linear",
    "pfx_get_linear_csr", "pfx_energy", "pfx_gradient", "pfx_hessian", "pfx_energy_dev", "pfx_gradient_dev",
    "pfx_hessian_dev", "pfx_set_dirichlet", "pfx_get_reduced_csr", "pfx_spmv", "pfx_spmv_dev", "pfx_cg",
    "pfx_cg_dev", "pfx_dirichlet_solve", "pfx_get_info", "pfx_timing_enable", "pfx_timing_get", "pfx_timing_reset",
    "pfx_set_owned", "pfx_cgs_set_workspace", "pfx_cgs_prepare", "pfx_cgs_residual", "pfx_cgs_start", "pfx_cgs_start2",
    "pfx_cgs_spmv", "pfx_cgs_update", "pfx_cgs_direction", "pfx_cgs_status",
    "pfx_newton_default_params", "pfx_newton_solve", "pfx_newton_solve_dev", "pfx_set_material_per_element",
    "pfx_assemble_mass", "pfx_assemble_rhs", "pfx_assemble_rhs_dev", "pfx_cgs_spmv_interior", "pfx_cgs_spmv_boundary", "pfx_cgs_fetch_product",
    "pfx_hessian_psd_dev", "pfx_set_diag_shift", "pfx_take_numeric_flag", "pfx_inertia_set_mass", "pfx_inertia_enable",
    "pfx_implicit_euler_step", "pfx_implicit_euler_step_dev", "pfx_mass_apply_dev", "pfx_hessian_add_mass_dev",
    "pfx_inertia_enable_host", "pfx_set_elasticity_tensor", "pfx_set_saint_venant_tensor", "pfx_set_element_models",
    "pfx_al_enable", "pfx_al_enable_host", "pfx_lumped_mass",
    "pfx_quadrature_points", "pfx_assemble_rhs_pointwise", "pfx_assemble_rhs_pointwise_dev", "pfx_set_material_per_point",
    "pfx_energy_dev_noalloc",
    "pfx_mass_quadrature_points", "pfx_assemble_mass_pointwise", "pfx_inertia_set_mass_pointwise", "pfx_assemble_rhs_density_pointwise",
    "pfx_amg_default_params", "pfx_amg_set_params", "pfx_amg_setup", "pfx_amg_get_info", "pfx_amg_get_aggregates", "pfx_amg_apply_dev",
    "pfx_solver_analyze", "pfx_solver_analyze64", "pfx_solver_factorize", "pfx_solver_factorize_dev", "pfx_solver_set_node_positions",
    "pfx_solver_get_layout", "pfx_matrix_generation",
    "pfx_comm_rccl_unique_id", "pfx_comm_create_rccl", "pfx_comm_create_host", "pfx_comm_destroy", "pfx_comm_last_error",
    "pfx_dist_create", "pfx_dist_destroy", "pfx_dist_last_error", "pfx_dist_halo_exchange_dev", "pfx_dist_all_reduce_dev",
    "pfx_dist_energy_dev", "pfx_dist_cg_dev", "pfx_dist_halo_bytes", "pfx_dist_amg_apply_dev", "pfx_dist_newton_dev", "pfx_dist_implicit_euler_step_dev", "pfx_tune_placement",
]


def elasticity_tensor(entries=None, lam=None, mu=None, young=None, nu=None, dim=3):
    """The 6 x 6 tensor of the reference's ElasticityTensor for size 3 (utils/ElasticityUtils.cpp): from the 21
    upper-triangular entries row by row (set_from_entries, :298-370), from (lambda, mu) (:372-413) or from
    (young, nu) (:416-437); the precedence is HookeLinearElasticity::add_multimaterial's (entries, young/nu, lambda/mu).
    dim = 2: the 3 x 3 tensor of size 2 -- 6 upper-triangular entries (:313-322) or the 4 orthotropic constants Ex Ey nuYX
    muXY (:457-466), (lambda, mu) (:374-384), (young, nu) (:418-424)."""
    if dim == 2:
        t = np.zeros((3, 3))
        if entries is not None:
            entries = np.asarray(entries, dtype=np.float64).reshape(-1)
            if entries.size == 4:
                ex, ey, nyx, mxy = entries
                t[0, :2] = 1.0 / ex, -nyx / ey
                t[1, 1] = 1.0 / ey
                t[2, 2] = 1.0 / mxy
            elif entries.size < 6:
                raise ValueError("6 upper-triangular entries (or the 4 orthotropic constants) expected")
            else:
                t[np.triu_indices(3)] = entries[:6]
            return np.triu(t) + np.triu(t, 1).T
        if young is not None:
            return np.array([[1.0, nu, 0.0], [nu, 1.0, 0.0], [0.0, 0.0, (1.0 - nu) / 2.0]]) * (young / (1.0 - nu * nu))
        return np.array([[2 * mu + lam, lam, 0.0], [lam, 2 * mu + lam, 0.0], [0.0, 0.0, mu]], dtype=np.float64)
    t = np.zeros((6, 6))
    if entries is not None:
        entries = np.asarray(entries, dtype=np.float64).reshape(-1)
        if entries.size == 9:
            # the orthotropic constants Ex Ey Ez nuYX nuZX nuZY muYZ muZX muXY, stored as set_orthotropic writes them (:439-455)
            ex, ey, ez, nyx, nzx, nzy, myz, mzx, mxy = entries
            t[0, :3] = 1.0 / ex, -nyx / ey, -nzx / ez
            t[1, 1:3] = 1.0 / ey, -nzy / ez
            t[2, 2] = 1.0 / ez
            t[3, 3], t[4, 4], t[5, 5] = 1.0 / myz, 1.0 / mzx, 1.0 / mxy
            return np.triu(t) + np.triu(t, 1).T
        if entries.size < 21:
            raise ValueError("21 upper-triangular entries (or the 9 orthotropic constants) expected")
        t[np.triu_indices(6)] = entries[:21]
    elif young is not None:
        f = young / ((1.0 + nu) * (1.0 - 2.0 * nu))
        t[:3, :3] = nu * f
        t[np.arange(3), np.arange(3)] = (1.0 - nu) * f
        t[np.arange(3, 6), np.arange(3, 6)] = ((1.0 - 2.0 * nu) / 2.0) * f
        return t
    else:
        t[:3, :3] = lam
        t[np.arange(3), np.arange(3)] = 2 * mu + lam
        t[np.arange(3, 6), np.arange(3, 6)] = mu
        return t
    return np.triu(t) + np.triu(t, 1).T


class PfxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"pfx error {code}: {msg}")
        self.code = code


class NewtonParams(C.Structure):
    """pfx_newton_params: solver/nonlinear/... of the reference's input-spec.json."""
    _fields_ = [("f_delta", C.c_double), ("grad_norm", C.c_double), ("first_grad_norm_tol", C.c_double),
                ("use_grad_norm_tol", C.c_double), ("max_iterations", C.c_int), ("use_grad_norm", C.c_int),
                ("relative_gradient", C.c_int), ("line_search", C.c_int), ("cg_precond", C.c_int),
                ("cg_max_iter", C.c_int), ("cg_tol", C.c_double)]


class NewtonInfo(C.Structure):
    _fields_ = [("status", C.c_int), ("iterations", C.c_int), ("line_search_iterations", C.c_int),
                ("descent_strategy", C.c_int), ("cg_iterations", C.c_int64), ("energy", C.c_double),
                ("grad_norm", C.c_double)]


NEWTON_STATUS = ["continue", "iteration_limit", "f_delta", "grad_norm", "not_started", "failed"]


class AmgParams(C.Structure):
    """pfx_amg_params (the options of polysolve's AMGCL block that have a counterpart here, input-spec.json:1397-1558)."""
    _fields_ = [("aggregate_passes", C.c_int), ("relax_degree", C.c_int), ("ncycle", C.c_int), ("max_levels", C.c_int),
                ("coarsest_dofs", C.c_int), ("power_iters", C.c_int), ("relax_lower", C.c_double), ("coarse_scale", C.c_double),
                ("mixed_precision", C.c_int)]


class AmgInfo(C.Structure):
    _fields_ = [("n_levels", C.c_int), ("n_vertices", C.c_int64 * 16), ("n_aggregates", C.c_int64 * 16),
                ("n_blocks", C.c_int64 * 16), ("lambda_max", C.c_double * 16), ("operator_complexity", C.c_double),
                ("symbolic_s", C.c_double), ("numeric_s", C.c_double)]


class Info(C.Structure):
    _fields_ = [("n_el", C.c_int64), ("n_nodes", C.c_int64), ("n_dofs", C.c_int64), ("n_blocks", C.c_int64),
                ("nnz", C.c_int64), ("n_loc", C.c_int), ("n_pts", C.c_int), ("dim", C.c_int),
                ("bytes_device", C.c_int64)]


# pfx_host_transport: the caller's own transport for host buffers (see include/polyfem_amd.h)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                          C.POINTER(C.c_void_p), C.POINTER(C.c_int64))
ALL_REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)


class HostTransport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("exchange", EXCHANGE_FN), ("all_reduce_sum", ALL_REDUCE_FN)]


_lib = None

_i64, _i32, _dbl, _vp = C.c_int64, C.c_int, C.c_double, C.c_void_p
_pi64, _pi32, _pd = C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_double)


def load():
    """Loads the HIP library; raises if it is missing (no CPU fallback exists on purpose)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: build it with `make -C polyfem_amd/csrc` (python -c 'import "
                           "__graft_entry__ as g; g.build()')")
    L = C.CDLL(LIB_PATH)
    sig = {
        "pfx_create": [C.POINTER(_vp), _i32],
        "pfx_destroy": [_vp],
        "pfx_set_stream": [_vp, _vp],
        "pfx_synchronize": [_vp],
        "pfx_set_mesh": [_vp, _i32, _i64, _i64, _i64, _pi32, _pi32, _pd, _i32],
        "pfx_set_material": [_vp, _i32, _i32, _dbl, _dbl],
        "pfx_set_material_per_element": [_vp, _i32, _i32, _pd, _pd, _i64],
        "pfx_set_material_per_point": [_vp, _i32, _i32, _pd, _pd, _i64, _i32],
        "pfx_set_elasticity_tensor": [_vp, _i32, _pd],
        "pfx_set_saint_venant_tensor": [_vp, _i32, _pd],
        "pfx_set_element_models": [_vp, _pi32, _i64],
        "pfx_al_enable": [_vp, C.c_double, _vp, _vp],
        "pfx_al_enable_host": [_vp, C.c_double, _pd, _pd],
        "pfx_lumped_mass": [_vp, _pd],
        "pfx_quadrature_points": [_vp, C.POINTER(C.c_int), _pd],
        "pfx_assemble_rhs_pointwise": [_vp, _pd, C.c_double, _pd, _pd],
        "pfx_assemble_rhs_pointwise_dev": [_vp, _vp, C.c_double, _pd, _vp],
        "pfx_build_pattern": [_vp, _pi64],
        "pfx_get_pattern": [_vp, _pi64, _pi32],
        "pfx_get_values": [_vp, _pd],
        "pfx_assemble_linear": [_vp],
        "pfx_get_linear_csr": [_vp, _pi64, _pi64, _pi32, _pd],
        "pfx_energy": [_vp, _pd, _pd],
        "pfx_gradient": [_vp, _pd, _pd],
        "pfx_hessian": [_vp, _pd, _i32, _pd],
        "pfx_energy_dev": [_vp, _vp, _vp],
        "pfx_energy_dev_noalloc": [_vp, _vp, _vp],
        "pfx_gradient_dev": [_vp, _vp, _vp],
        "pfx_hessian_dev": [_vp, _vp],
        "pfx_hessian_psd_dev": [_vp, _vp],
        "pfx_set_dirichlet": [_vp, _pi32, _i64],
        "pfx_get_reduced_csr": [_vp, _pi64, _pi64, _pi64, _pi32, _pd],
        "pfx_spmv": [_vp, _pd, _pd],
        "pfx_spmv_dev": [_vp, _vp, _vp],
        "pfx_cg": [_vp, _pd, _pd, _i32, _i32, _dbl, C.POINTER(C.c_int), _pd],
        "pfx_cg_dev": [_vp, _vp, _vp, _i32, _i32, _dbl, C.POINTER(C.c_int), _pd],
        "pfx_dirichlet_solve": [_vp, _pd, _pd, _i32, _i32, _dbl, C.POINTER(C.c_int), _pd],
        "pfx_get_info": [_vp, C.POINTER(Info)],
        "pfx_timing_enable": [_vp, _i32],
        "pfx_timing_get": [_vp, C.c_char_p, _pd, _pi64],
        "pfx_timing_reset": [_vp],
        "pfx_set_owned": [_vp, _i64, C.POINTER(C.c_uint8)],
        "pfx_cgs_set_workspace": [_vp, _vp, _i64],
        "pfx_cgs_prepare": [_vp, _vp, _i32],
        "pfx_cgs_residual": [_vp, _vp, _vp, _vp],
        "pfx_cgs_start": [_vp, _vp, _dbl, _vp],
        "pfx_cgs_start2": [_vp, _vp],
        "pfx_cgs_spmv": [_vp, _vp],
        "pfx_cgs_spmv_interior": [_vp],
        "pfx_cgs_spmv_boundary": [_vp, _vp],
        "pfx_cgs_fetch_product": [_vp],
        "pfx_cgs_update": [_vp, _i32, _vp, _vp],
        "pfx_cgs_direction": [_vp, _i32, _vp],
        "pfx_cgs_status": [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), _pd, _pd],
        "pfx_assemble_mass": [_vp, _dbl, _pd],
        "pfx_assemble_rhs": [_vp, _pd, _dbl, _pd, _pd],
        "pfx_assemble_rhs_dev": [_vp, _pd, _dbl, _pd, _vp],
        "pfx_inertia_set_mass": [_vp, _dbl, _pd],
        "pfx_inertia_enable": [_vp, _dbl, _vp],
        "pfx_inertia_enable_host": [_vp, _dbl, _pd],
        "pfx_implicit_euler_step": [_vp, _dbl, _pd, _pd, _pd, _pd, C.POINTER(NewtonParams), C.POINTER(NewtonInfo)],
        "pfx_implicit_euler_step_dev": [_vp, _dbl, _vp, _vp, _vp, _vp, C.POINTER(NewtonParams), C.POINTER(NewtonInfo)],
        "pfx_set_diag_shift": [_vp, _dbl],
        "pfx_mass_apply_dev": [_vp, _vp, _vp],
        "pfx_hessian_add_mass_dev": [_vp, _dbl],
        "pfx_take_numeric_flag": [_vp, C.POINTER(C.c_int)],
        "pfx_newton_solve": [_vp, _pd, _pd, C.POINTER(NewtonParams), C.POINTER(NewtonInfo)],
        "pfx_newton_solve_dev": [_vp, _vp, _vp, C.POINTER(NewtonParams), C.POINTER(NewtonInfo)],
        "pfx_mass_quadrature_points": [_vp, C.POINTER(C.c_int), _pd],
        "pfx_assemble_mass_pointwise": [_vp, _pd, _i64, _i32],
        "pfx_inertia_set_mass_pointwise": [_vp, _pd, _i64, _i32],
        "pfx_assemble_rhs_density_pointwise": [_vp, _pd, _pd, _pd, _i64, _i32, _pd],
        "pfx_amg_set_params": [_vp, C.POINTER(AmgParams)],
        "pfx_amg_setup": [_vp],
        "pfx_amg_get_info": [_vp, C.POINTER(AmgInfo)],
        "pfx_amg_get_aggregates": [_vp, _i32, _pi32],
        "pfx_amg_apply_dev": [_vp, _vp, _vp],
        "pfx_solver_analyze": [_vp, _i64, _pi32, _pi32, _i32],
        "pfx_solver_analyze64": [_vp, _i64, _pi64, _pi32, _i32],
        "pfx_solver_factorize": [_vp, _pd],
        "pfx_solver_factorize_dev": [_vp, _vp],
        "pfx_solver_set_node_positions": [_vp, _pd, _i64],
        "pfx_solver_get_layout": [_vp, C.POINTER(C.c_int), _pi64, _pi64],
        "pfx_matrix_generation": [_vp, _pi64],
        "pfx_comm_rccl_unique_id": [_vp],
        "pfx_comm_create_rccl": [C.POINTER(_vp), _vp, _i32, _i32, _i32],
        "pfx_comm_create_host": [C.POINTER(_vp), C.POINTER(HostTransport), _i32, _i32],
        "pfx_dist_create": [C.POINTER(_vp), _vp, _vp, _i32, _pi32, _pi64, _pi32, _pi64],
        "pfx_dist_halo_exchange_dev": [_vp, _vp],
        "pfx_dist_all_reduce_dev": [_vp, _vp, _i32],
        "pfx_dist_energy_dev": [_vp, _vp, _vp],
        "pfx_dist_cg_dev": [_vp, _vp, _vp, _i32, _i32, _dbl, _i32, C.POINTER(C.c_int), _pd],
        "pfx_dist_newton_dev": [_vp, _vp, _vp, C.POINTER(NewtonParams), C.POINTER(NewtonInfo)],
        "pfx_dist_implicit_euler_step_dev": [_vp, _dbl, _vp, _vp, _vp, _vp, C.POINTER(NewtonParams), C.POINTER(NewtonInfo)],
        "pfx_tune_placement": [_vp],
        "pfx_dist_halo_bytes": [_vp, _pi64],
        "pfx_dist_amg_apply_dev": [_vp, _vp, _vp],
    }
    for name, args in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = None if name == "pfx_destroy" else C.c_int
    L.pfx_newton_default_params.argtypes = [C.POINTER(NewtonParams)]
    L.pfx_newton_default_params.restype = None
    L.pfx_amg_default_params.argtypes = [C.POINTER(AmgParams)]
    L.pfx_amg_default_params.restype = None
    L.pfx_last_error.argtypes = [_vp]
    L.pfx_last_error.restype = C.c_char_p
    for name in ("pfx_comm_destroy", "pfx_dist_destroy"):
        getattr(L, name).argtypes = [_vp]
        getattr(L, name).restype = None
    for name in ("pfx_comm_last_error", "pfx_dist_last_error"):
        getattr(L, name).argtypes = [_vp]
        getattr(L, name).restype = C.c_char_p
    _lib = L
    return L


def _d(a):
    return a.ctypes.data_as(_pd)


def _i(a):
    return a.ctypes.data_as(_pi32)


def _l(a):
    return a.ctypes.data_as(_pi64)


class Engine:
    """One device context (one per process / GPU)."""

    def __init__(self, device: int = 0):
        self.L = load()
        self.h = _vp()
        rc = self.L.pfx_create(C.byref(self.h), device)
        if rc != 0:
            raise PfxError(rc, "pfx_create failed (no usable HIP device?)")
        self.mesh = None
        self.model = None

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.pfx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise PfxError(rc, self.L.pfx_last_error(self.h).decode())

    # ---- setup -------------------------------------------------------------------------------------------
    def set_stream(self, hip_stream: int | None):
        self._ck(self.L.pfx_set_stream(self.h, _vp(hip_stream) if hip_stream else None))

    def synchronize(self):
        self._ck(self.L.pfx_synchronize(self.h))

    def set_mesh(self, mesh):
        conn = np.ascontiguousarray(mesh.conn, dtype=np.int32)
        gconn = np.ascontiguousarray(mesh.gconn, dtype=np.int32)
        xyz = np.ascontiguousarray(mesh.xyz, dtype=np.float64)
        self._ck(self.L.pfx_set_mesh(self.h, mesh.kind, conn.shape[0], mesh.n_nodes, xyz.shape[0], _i(conn), _i(gconn),
                                     _d(xyz), mesh.quad_order))
        self.mesh = mesh

    def _mesh_dim(self) -> int:
        return 2 if self.mesh is not None and self.mesh.kind in (P1TRI, P2TRI) else 3

    def set_material(self, model: int, lam=1.0, mu=1.0, dim: int | None = None):
        """Scalars: uniform Lame parameters; 1-D arrays: one value per element (multi-material bodies); 2-D arrays
        (n_el, n_pts): one value per quadrature point (space-dependent materials).  dim: 3, or 2 on a triangle mesh
        (default: that of the mesh set before)."""
        if dim is None:
            dim = self._mesh_dim()
        if np.ndim(lam) == 0:
            self._ck(self.L.pfx_set_material(self.h, model, dim, float(lam), float(mu)))
        else:
            la = np.ascontiguousarray(lam, dtype=np.float64)
            ma = np.ascontiguousarray(mu, dtype=np.float64)
            assert la.shape == ma.shape and la.ndim in (1, 2)
            if la.ndim == 1:
                self._ck(self.L.pfx_set_material_per_element(self.h, model, dim, _d(la), _d(ma), la.size))
            else:
                self._ck(self.L.pfx_set_material_per_point(self.h, model, dim, _d(la), _d(ma), la.shape[0], la.shape[1]))
        self.model = model

    def set_element_models(self, models):
        """MultiModel: one of LINEAR_ELASTICITY / NEOHOOKEAN / SAINT_VENANT (ids or names) per element, after set_material."""
        ids = np.ascontiguousarray([MODEL_BY_NAME[m] if isinstance(m, str) else int(m) for m in models], dtype=np.int32)
        self._ck(self.L.pfx_set_element_models(self.h, ids.ctypes.data_as(_pi32), ids.size))

    # ---- augmented-Lagrangian Dirichlet conditions -----------------------------------------------------------
    def al_enable(self, weight: float, d_masked_lumped_mass: int, d_target: int):
        """Device pointers (kept by the engine until al_disable)."""
        self._ck(self.L.pfx_al_enable(self.h, weight, _vp(d_masked_lumped_mass), _vp(d_target)))

    def al_enable_host(self, weight: float, masked_lumped_mass, target):
        m = np.ascontiguousarray(masked_lumped_mass, dtype=np.float64)
        t = np.ascontiguousarray(target, dtype=np.float64)
        self._ck(self.L.pfx_al_enable_host(self.h, weight, _d(m), _d(t)))

    def al_disable(self):
        self._ck(self.L.pfx_al_enable(self.h, 0.0, None, None))

    def lumped_mass(self):
        out = np.zeros(self.info().n_dofs)
        self._ck(self.L.pfx_lumped_mass(self.h, _d(out)))
        return out

    def al_solve(self, x0, rhs, dirichlet_dofs, dirichlet_values, lumped_mass=None, initial_weight=1e6, max_weight=1e11,
                 force_al=False, **newton_params):
        """ALSolver::solve (solver/ALSolver.cpp:22-84).  While the state with the Dirichlet values imposed is invalid
        (non-finite energy) -- or once, with force_al -- solve the full-size problem with the penalty, doubling its
        weight; then one solve with the Dirichlet dofs held at their values.  lumped_mass: vector of n_dofs (None: the
        identity, the static problem's choice).  Returns (x, info of the final solve, list of penalty weights used)."""
        n = self.info().n_dofs
        dofs = np.ascontiguousarray(dirichlet_dofs, dtype=np.int32)
        target = np.zeros(n)
        target[dofs] = dirichlet_values
        mass = np.ones(n) if lumped_mass is None else np.asarray(lumped_mass, dtype=np.float64)
        masked = np.zeros(n)
        masked[dofs] = mass[dofs]
        x = np.array(x0, dtype=np.float64, copy=True)

        def projected_valid():
            xp = x.copy()
            xp[dofs] = target[dofs]
            try:
                return bool(np.isfinite(self.energy(xp)))
            except PfxError:
                self.take_numeric_flag()
                return False

        weight, used = float(initial_weight), []
        while force_al or not projected_valid():
            force_al = False
            self.set_dirichlet(np.zeros(0, dtype=np.int32))
            self.al_enable_host(weight, masked, target)
            try:
                x, _ = self.newton_solve(x, rhs, **newton_params)
            finally:
                self.al_disable()
            used.append(weight)
            weight *= 2
            if weight >= max_weight:
                raise PfxError(PFX_ERR_NUMERIC, f"Unable to solve AL problem, weight {weight} >= {max_weight}, stopping")
        self.set_dirichlet(dofs)
        x[dofs] = target[dofs]
        x, info = self.newton_solve(x, rhs, **newton_params)
        return x, info, used

    def _voigt(self, tensor, dim):
        dim = self._mesh_dim() if dim is None else int(dim)
        t = np.ascontiguousarray(tensor, dtype=np.float64).reshape(-1)
        if dim in (2, 3) and t.size != (36 if dim == 3 else 9):
            raise ValueError("the elasticity tensor is 6 x 6 for dim = 3 and 3 x 3 for dim = 2")
        return t, dim

    def set_elasticity_tensor(self, tensor, dim=None):
        """HookeLinearElasticity: symmetric 6 x 6 tensor (Voigt order xx yy zz yz xz xy), or 3 x 3 (xx yy xy) on a
        triangle mesh; linear assembly only.  dim defaults to the dimension of the mesh."""
        t, dim = self._voigt(tensor, dim)
        self._ck(self.L.pfx_set_elasticity_tensor(self.h, dim, _d(t)))

    def set_saint_venant_tensor(self, tensor, dim=None):
        """SaintVenantElasticity with its `elasticity_tensor` (same layout) instead of (lambda, mu)."""
        t, dim = self._voigt(tensor, dim)
        self._ck(self.L.pfx_set_saint_venant_tensor(self.h, dim, _d(t)))

    def build_pattern(self) -> int:
        nnz = C.c_int64()
        self._ck(self.L.pfx_build_pattern(self.h, C.byref(nnz)))
        return nnz.value

    def info(self) -> Info:
        inf = Info()
        self._ck(self.L.pfx_get_info(self.h, C.byref(inf)))
        return inf

    # ---- pattern / values --------------------------------------------------------------------------------
    def get_pattern(self):
        inf = self.info()
        row_ptr = np.zeros(inf.n_dofs + 1, dtype=np.int64)
        col_idx = np.zeros(max(inf.nnz, 1), dtype=np.int32)
        self._ck(self.L.pfx_get_pattern(self.h, _l(row_ptr), _i(col_idx)))
        return row_ptr, col_idx[: inf.nnz]

    def get_values(self):
        inf = self.info()
        v = np.zeros(max(inf.nnz, 1))
        self._ck(self.L.pfx_get_values(self.h, _d(v)))
        return v[: inf.nnz]

    # ---- assembly ----------------------------------------------------------------------------------------
    def assemble_linear(self):
        self._ck(self.L.pfx_assemble_linear(self.h))

    def get_linear_csr(self):
        nnz = C.c_int64()
        self._ck(self.L.pfx_get_linear_csr(self.h, C.byref(nnz), None, None, None))
        n = self.info().n_dofs
        row_ptr = np.zeros(n + 1, dtype=np.int64)
        col_idx = np.zeros(max(nnz.value, 1), dtype=np.int32)
        vals = np.zeros(max(nnz.value, 1))
        self._ck(self.L.pfx_get_linear_csr(self.h, C.byref(nnz), _l(row_ptr), _i(col_idx), _d(vals)))
        return row_ptr, col_idx[: nnz.value], vals[: nnz.value]

    def energy(self, u) -> float:
        u = np.ascontiguousarray(u, dtype=np.float64)
        e = C.c_double()
        self._ck(self.L.pfx_energy(self.h, _d(u), C.byref(e)))
        return e.value

    def gradient(self, u):
        u = np.ascontiguousarray(u, dtype=np.float64)
        g = np.zeros_like(u)
        self._ck(self.L.pfx_gradient(self.h, _d(u), _d(g)))
        return g

    def hessian(self, u, project_to_psd: bool = False, fetch: bool = True):
        u = np.ascontiguousarray(u, dtype=np.float64)
        if fetch:
            v = np.zeros(max(self.info().nnz, 1))
            self._ck(self.L.pfx_hessian(self.h, _d(u), int(project_to_psd), _d(v)))
            return v[: self.info().nnz]
        self._ck(self.L.pfx_hessian(self.h, _d(u), int(project_to_psd), None))
        return None

    # device-pointer variants (raw integer addresses, e.g. torch.Tensor.data_ptr())
    def energy_dev(self, d_u: int, d_e: int):
        self._ck(self.L.pfx_energy_dev(self.h, _vp(d_u), _vp(d_e)))

    def energy_dev_noalloc(self, d_u: int, d_e: int):
        """pfx_energy_dev without any device allocation (static scratch of the library, n_el <= 2^20)."""
        self._ck(self.L.pfx_energy_dev_noalloc(self.h, _vp(d_u), _vp(d_e)))

    def gradient_dev(self, d_u: int, d_g: int):
        self._ck(self.L.pfx_gradient_dev(self.h, _vp(d_u), _vp(d_g)))

    def hessian_dev(self, d_u: int, project_to_psd: bool = False):
        if project_to_psd:
            self._ck(self.L.pfx_hessian_psd_dev(self.h, _vp(d_u)))
        else:
            self._ck(self.L.pfx_hessian_dev(self.h, _vp(d_u)))

    # ---- Dirichlet ---------------------------------------------------------------------------------------
    def set_dirichlet(self, sorted_dofs):
        d = np.ascontiguousarray(sorted_dofs, dtype=np.int32)
        self._ck(self.L.pfx_set_dirichlet(self.h, _i(d) if d.size else None, d.size))

    def get_reduced_csr(self):
        nr, nnz = C.c_int64(), C.c_int64()
        self._ck(self.L.pfx_get_reduced_csr(self.h, C.byref(nr), C.byref(nnz), None, None, None))
        row_ptr = np.zeros(nr.value + 1, dtype=np.int64)
        col_idx = np.zeros(max(nnz.value, 1), dtype=np.int32)
        vals = np.zeros(max(nnz.value, 1))
        self._ck(self.L.pfx_get_reduced_csr(self.h, C.byref(nr), C.byref(nnz), _l(row_ptr), _i(col_idx), _d(vals)))
        return row_ptr, col_idx[: nnz.value], vals[: nnz.value]

    # ---- solve -------------------------------------------------------------------------------------------
    def spmv(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros_like(x)
        self._ck(self.L.pfx_spmv(self.h, _d(x), _d(y)))
        return y

    def spmv_dev(self, d_x: int, d_y: int):
        self._ck(self.L.pfx_spmv_dev(self.h, _vp(d_x), _vp(d_y)))

    def cg(self, b, x0=None, precond: int = PRECOND_JACOBI, max_iter: int = 1000, tol: float = 1e-12):
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.zeros_like(b) if x0 is None else np.array(x0, dtype=np.float64, copy=True)
        it, err = C.c_int(), C.c_double()
        self._ck(self.L.pfx_cg(self.h, _d(b), _d(x), precond, max_iter, tol, C.byref(it), C.byref(err)))
        return x, it.value, err.value

    def cg_dev(self, d_b: int, d_x: int, precond: int = PRECOND_JACOBI, max_iter: int = 1000, tol: float = 1e-12):
        it, err = C.c_int(), C.c_double()
        self._ck(self.L.pfx_cg_dev(self.h, _vp(d_b), _vp(d_x), precond, max_iter, tol, C.byref(it), C.byref(err)))
        return it.value, err.value

    # ---- AMG preconditioner (precond = PRECOND_AMG) -----------------------------------------------------------
    def amg_set_params(self, **params):
        """aggregate_passes, relax_degree, ncycle, max_levels, coarsest_dofs, power_iters, relax_lower, coarse_scale, mixed_precision (see pfx_amg_params)."""
        p = AmgParams()
        self.L.pfx_amg_default_params(C.byref(p))
        for k, v in params.items():
            if k not in dict(AmgParams._fields_):
                raise TypeError(f"unknown AMG parameter {k!r}")
            setattr(p, k, v)
        self._ck(self.L.pfx_amg_set_params(self.h, C.byref(p)))

    def amg_setup(self):
        self._ck(self.L.pfx_amg_setup(self.h))

    def amg_info(self) -> dict:
        a = AmgInfo()
        self._ck(self.L.pfx_amg_get_info(self.h, C.byref(a)))
        n = a.n_levels
        return dict(n_levels=n, n_vertices=list(a.n_vertices[:n]), n_aggregates=list(a.n_aggregates[:n]), n_blocks=list(a.n_blocks[:n]),
                    lambda_max=list(a.lambda_max[:n]), operator_complexity=a.operator_complexity, symbolic_s=a.symbolic_s,
                    numeric_s=a.numeric_s)

    def amg_aggregates(self, level: int):
        info = self.amg_info()
        n = self.info().n_nodes if level == 0 else info["n_vertices"][level]
        out = np.zeros(n, dtype=np.int32)
        self._ck(self.L.pfx_amg_get_aggregates(self.h, level, _i(out)))
        return out

    def amg_apply_dev(self, d_r: int, d_z: int):
        self._ck(self.L.pfx_amg_apply_dev(self.h, _vp(d_r), _vp(d_z)))

    def mass_quadrature_points(self):
        """Physical positions of the points of the mass quadrature rule, shape (n_el, n_pts, dim)."""
        n_pts = C.c_int()
        self._ck(self.L.pfx_mass_quadrature_points(self.h, C.byref(n_pts), None))
        out = np.zeros((self.info().n_el, n_pts.value, self._mesh_dim()))
        self._ck(self.L.pfx_mass_quadrature_points(self.h, C.byref(n_pts), _d(out)))
        return out

    def assemble_mass(self, density=1.0):
        """Assembler<Mass>::assemble(is_mass=True) into the resident pattern; density: scalar, one value per element, or
        (n_el, n_pts) values at the points of the mass quadrature rule (mass_quadrature_points)."""
        if np.ndim(density) == 2:
            rho = np.ascontiguousarray(density, dtype=np.float64)
            self._ck(self.L.pfx_assemble_mass_pointwise(self.h, _d(rho), rho.shape[0], rho.shape[1]))
            return
        if np.ndim(density) == 0:
            self._ck(self.L.pfx_assemble_mass(self.h, float(density), None))
        else:
            rho = np.ascontiguousarray(density, dtype=np.float64)
            self._ck(self.L.pfx_assemble_mass(self.h, float(rho[0]), _d(rho)))

    def quadrature_points(self):
        """Physical positions of the quadrature points of the stiffness rule, shape (n_el, n_pts, dim)."""
        n_pts = C.c_int()
        self._ck(self.L.pfx_quadrature_points(self.h, C.byref(n_pts), None))
        out = np.zeros((self.info().n_el, n_pts.value, self._mesh_dim()))
        self._ck(self.L.pfx_quadrature_points(self.h, C.byref(n_pts), _d(out)))
        return out

    def assemble_rhs_pointwise(self, force_pts, density=1.0):
        """Body-force vector for a force density given at every quadrature point, (n_el, n_pts, d)."""
        f = np.ascontiguousarray(force_pts, dtype=np.float64)
        inf = self.info()
        if f.ndim != 3 or f.shape[0] != inf.n_el or f.shape[1] != inf.n_pts:
            raise ValueError("force_pts must have shape (n_el, n_pts, d)")
        if np.ndim(density) == 2:
            rho = np.ascontiguousarray(density, dtype=np.float64)
            out = np.zeros(inf.n_dofs)
            self._ck(self.L.pfx_assemble_rhs_density_pointwise(self.h, None, _d(f), _d(rho), rho.shape[0], rho.shape[1], _d(out)))
            return out
        rho_e = None if np.ndim(density) == 0 else np.ascontiguousarray(density, dtype=np.float64)
        out = np.zeros(inf.n_dofs)
        self._ck(self.L.pfx_assemble_rhs_pointwise(self.h, _d(f), float(density) if rho_e is None else float(rho_e[0]),
                                                   _d(rho_e) if rho_e is not None else None, _d(out)))
        return out

    def assemble_rhs(self, force, density=1.0):
        """RhsAssembler::assemble for a constant body force (one component per dof of a node)."""
        f = np.zeros(3)
        f[:np.size(force)] = np.asarray(force, dtype=np.float64).reshape(-1)
        out = np.zeros(self.info().n_dofs)
        if np.ndim(density) == 2:  # (n_el, n_pts) values at the stiffness rule's points (quadrature_points)
            rho = np.ascontiguousarray(density, dtype=np.float64)
            self._ck(self.L.pfx_assemble_rhs_density_pointwise(self.h, _d(f), None, _d(rho), rho.shape[0], rho.shape[1], _d(out)))
            return out
        if np.ndim(density) == 0:
            self._ck(self.L.pfx_assemble_rhs(self.h, _d(f), float(density), None, _d(out)))
        else:
            rho = np.ascontiguousarray(density, dtype=np.float64)
            self._ck(self.L.pfx_assemble_rhs(self.h, _d(f), float(rho[0]), _d(rho), _d(out)))
        return out

    def newton_solve(self, x0, rhs=None, **params):
        """Minimises E(x) - rhs.x over the free dofs (see pfx_newton_solve).  Returns (x, info dict); raises PfxError
        where the reference's solver throws (the iterate reached is lost, as there)."""
        p = NewtonParams()
        self.L.pfx_newton_default_params(C.byref(p))
        for k, v in params.items():
            setattr(p, k, v)
        x = np.array(x0, dtype=np.float64, copy=True)
        r = None if rhs is None else np.ascontiguousarray(rhs, dtype=np.float64)
        info = NewtonInfo()
        self._ck(self.L.pfx_newton_solve(self.h, _d(r) if r is not None else None, _d(x), C.byref(p), C.byref(info)))
        return x, {f[0]: getattr(info, f[0]) for f in NewtonInfo._fields_}

    def newton_solve_dev(self, d_x: int, d_rhs: int = 0, **params):
        """Same with device pointers (x in/out, rhs may be 0); returns the info dict."""
        p = NewtonParams()
        self.L.pfx_newton_default_params(C.byref(p))
        for k, v in params.items():
            setattr(p, k, v)
        info = NewtonInfo()
        self._ck(self.L.pfx_newton_solve_dev(self.h, _vp(d_rhs) if d_rhs else None, _vp(d_x), C.byref(p), C.byref(info)))
        return {f[0]: getattr(info, f[0]) for f in NewtonInfo._fields_}

    # ---- transient steps -------------------------------------------------------------------------------------
    def inertia_set_mass(self, density=1.0):
        """Assembles the mass matrix (scalar density or one value per element) and keeps it for the inertia term."""
        if np.ndim(density) == 2:
            d = np.ascontiguousarray(density, dtype=np.float64)
            self._ck(self.L.pfx_inertia_set_mass_pointwise(self.h, _d(d), d.shape[0], d.shape[1]))
        elif np.ndim(density) == 0:
            self._ck(self.L.pfx_inertia_set_mass(self.h, float(density), None))
        else:
            d = np.ascontiguousarray(density, dtype=np.float64)
            assert d.size == self.info().n_el
            self._ck(self.L.pfx_inertia_set_mass(self.h, 0.0, _d(d)))

    def inertia_enable(self, acceleration_scaling: float, d_x_tilde: int):
        """Newton solves minimise acceleration_scaling * (E - rhs.x) + 1/2 (x - x_tilde)^T M (x - x_tilde) from now on
        (device pointer to x_tilde); inertia_disable() restores the static objective."""
        self._ck(self.L.pfx_inertia_enable(self.h, float(acceleration_scaling), _vp(d_x_tilde)))

    def inertia_enable_host(self, acceleration_scaling: float, x_tilde):
        xt = np.ascontiguousarray(x_tilde, dtype=np.float64)
        assert xt.size == self.info().n_dofs
        self._ck(self.L.pfx_inertia_enable_host(self.h, float(acceleration_scaling), _d(xt)))

    def inertia_disable(self):
        self._ck(self.L.pfx_inertia_enable(self.h, 0.0, None))

    def _newton_params(self, params):
        p = NewtonParams()
        self.L.pfx_newton_default_params(C.byref(p))
        for k, v in params.items():
            setattr(p, k, v)
        return p

    def implicit_euler_step(self, dt: float, x, v, rhs=None, **params):
        """One implicit Euler step of the transient nonlinear solve (see pfx_implicit_euler_step).  Returns
        (x, v, a, info dict)."""
        p = self._newton_params(params)
        x = np.array(x, dtype=np.float64, copy=True)
        v = np.array(v, dtype=np.float64, copy=True)
        a = np.zeros_like(x)
        r = None if rhs is None else np.ascontiguousarray(rhs, dtype=np.float64)
        info = NewtonInfo()
        self._ck(self.L.pfx_implicit_euler_step(self.h, float(dt), _d(r) if r is not None else None, _d(x), _d(v), _d(a),
                                                C.byref(p), C.byref(info)))
        return x, v, a, {f[0]: getattr(info, f[0]) for f in NewtonInfo._fields_}

    def implicit_euler_step_dev(self, dt: float, d_x: int, d_v: int, d_a: int = 0, d_rhs: int = 0, **params):
        p = self._newton_params(params)
        info = NewtonInfo()
        self._ck(self.L.pfx_implicit_euler_step_dev(self.h, float(dt), _vp(d_rhs) if d_rhs else None, _vp(d_x), _vp(d_v),
                                                    _vp(d_a) if d_a else None, C.byref(p), C.byref(info)))
        return {f[0]: getattr(info, f[0]) for f in NewtonInfo._fields_}

    def dirichlet_solve(self, b, precond: int = PRECOND_JACOBI, max_iter: int = 1000, tol: float = 1e-12):
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.zeros_like(b)
        it, err = C.c_int(), C.c_double()
        self._ck(self.L.pfx_dirichlet_solve(self.h, _d(b), _d(x), precond, max_iter, tol, C.byref(it), C.byref(err)))
        return x, it.value, err.value

    # ---- multi-GPU building blocks (device pointers as integers) ---------------------------------------------
    def set_owned(self, n_owned_nodes: int, elem_owned=None):
        if elem_owned is None:
            self._ck(self.L.pfx_set_owned(self.h, n_owned_nodes, None))
        else:
            f = np.ascontiguousarray(elem_owned, dtype=np.uint8)
            self._ck(self.L.pfx_set_owned(self.h, n_owned_nodes, f.ctypes.data_as(C.POINTER(C.c_uint8))))

    def set_diag_shift(self, shift: float):
        """The solver side works with A + shift * I on the free dofs until set back to 0."""
        self._ck(self.L.pfx_set_diag_shift(self.h, float(shift)))

    def mass_apply_dev(self, d_v: int, d_y: int):
        """y[owned rows] = M v with the mass matrix of inertia_set_mass (device pointers)."""
        self._ck(self.L.pfx_mass_apply_dev(self.h, _vp(d_v), _vp(d_y)))

    def hessian_add_mass_dev(self, weight: float):
        """Resident matrix := weight * matrix + M (after hessian_dev)."""
        self._ck(self.L.pfx_hessian_add_mass_dev(self.h, float(weight)))

    def take_numeric_flag(self) -> bool:
        """True if a *_dev evaluation met an inverted element since the last call; clears the flag."""
        f = C.c_int(0)
        self._ck(self.L.pfx_take_numeric_flag(self.h, C.byref(f)))
        return bool(f.value)

    def cgs_set_workspace(self, d_work: int, n_doubles: int = 0):
        """Caller-owned CG workspace of n_doubles >= 4 * n_dofs doubles; d_work = 0 returns to the engine's own."""
        if getattr(self, "h", None):
            self._ck(self.L.pfx_cgs_set_workspace(self.h, _vp(d_work) if d_work else None, int(n_doubles)))
            self.cgs_workspace = int(d_work) if d_work else 0  # what the context points at now (0: its own)

    def cgs_prepare(self, d_x: int, precond: int):
        self._ck(self.L.pfx_cgs_prepare(self.h, _vp(d_x), precond))

    def cgs_residual(self, d_b: int, d_x: int, d_scal: int):
        self._ck(self.L.pfx_cgs_residual(self.h, _vp(d_b), _vp(d_x), _vp(d_scal) if d_scal else None))

    def cgs_start(self, d_x: int, tol: float, d_scal: int):
        self._ck(self.L.pfx_cgs_start(self.h, _vp(d_x), tol, _vp(d_scal) if d_scal else None))

    def cgs_start2(self, d_scal: int):
        self._ck(self.L.pfx_cgs_start2(self.h, _vp(d_scal) if d_scal else None))

    def cgs_spmv(self, d_scal: int):
        self._ck(self.L.pfx_cgs_spmv(self.h, _vp(d_scal) if d_scal else None))

    def cgs_spmv_interior(self):
        self._ck(self.L.pfx_cgs_spmv_interior(self.h))

    def cgs_spmv_boundary(self, d_scal: int):
        self._ck(self.L.pfx_cgs_spmv_boundary(self.h, _vp(d_scal) if d_scal else None))

    def cgs_fetch_product(self):
        """tmp = A p in row order in the workspace (the update step reads it wherever it is; a caller that reads it does this first)."""
        self._ck(self.L.pfx_cgs_fetch_product(self.h))

    def cgs_update(self, it: int, d_x: int, d_scal: int):
        self._ck(self.L.pfx_cgs_update(self.h, it, _vp(d_x), _vp(d_scal) if d_scal else None))

    def cgs_direction(self, it: int, d_scal: int):
        self._ck(self.L.pfx_cgs_direction(self.h, it, _vp(d_scal) if d_scal else None))

    def cgs_status(self):
        done, it, rr, bb = C.c_int(), C.c_int(), C.c_double(), C.c_double()
        self._ck(self.L.pfx_cgs_status(self.h, C.byref(done), C.byref(it), C.byref(rr), C.byref(bb)))
        return bool(done.value), it.value, rr.value, bb.value

    # ---- a caller's matrix (the context is then a pure solver context) ----------------------------------------
    def solver_analyze(self, n: int, outer, inner, block_size: int = 0):
        """polysolve analyzePattern for a compressed-column matrix handed in by the caller (pfx_solver_analyze):
        outer int32 or int64 (n + 1), inner int32.  block_size 0 lets the library choose 3 x 3 node blocks or scalars."""
        inner = np.ascontiguousarray(inner, dtype=np.int32)
        outer = np.asarray(outer)
        if outer.dtype == np.int32:
            outer = np.ascontiguousarray(outer)
            self._ck(self.L.pfx_solver_analyze(self.h, int(n), _i(outer), _i(inner), int(block_size)))
        else:
            outer = np.ascontiguousarray(outer, dtype=np.int64)
            self._ck(self.L.pfx_solver_analyze64(self.h, int(n), _l(outer), _i(inner), int(block_size)))

    def solver_factorize(self, values):
        """Values in the order of `inner`; PfxError with code PFX_ERR_NUMERIC for a non-finite entry or a diagonal entry
        that is not positive."""
        v = np.ascontiguousarray(values, dtype=np.float64)
        self._ck(self.L.pfx_solver_factorize(self.h, _d(v)))

    def solver_factorize_dev(self, d_values: int):
        self._ck(self.L.pfx_solver_factorize_dev(self.h, _vp(d_values)))

    def solver_set_node_positions(self, xyz):
        if xyz is None:
            self._ck(self.L.pfx_solver_set_node_positions(self.h, None, 0))
            return
        x = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
        self._ck(self.L.pfx_solver_set_node_positions(self.h, _d(x), x.shape[0]))

    def solver_layout(self):
        """(block size, number of blocks, stored entries incl. explicit zeros) chosen by solver_analyze."""
        bs, nb, ne = C.c_int(), C.c_int64(), C.c_int64()
        self._ck(self.L.pfx_solver_get_layout(self.h, C.byref(bs), C.byref(nb), C.byref(ne)))
        return bs.value, nb.value, ne.value

    def tune_placement(self):
        """pfx_tune_placement: search a fast placement for the solver's value array and the single-precision copies of an
        existing AMG hierarchy now (the CG loops do it by themselves for the former once a solve has proven long)."""
        self._ck(self.L.pfx_tune_placement(self.h))

    def matrix_generation(self) -> int:
        g = C.c_int64()
        self._ck(self.L.pfx_matrix_generation(self.h, C.byref(g)))
        return g.value

    # ---- timing ------------------------------------------------------------------------------------------
    def timing_enable(self, on: bool = True):
        self._ck(self.L.pfx_timing_enable(self.h, int(on)))

    def timing_reset(self):
        self._ck(self.L.pfx_timing_reset(self.h))

    def timing_get(self, family: str):
        ms, n = C.c_double(), C.c_int64()
        self._ck(self.L.pfx_timing_get(self.h, family.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value


# --------------------------------------------------------------------------------------------------------
# Host-side mirror of the reference interface
# --------------------------------------------------------------------------------------------------------


class StiffnessMatrix:
    """Eigen::SparseMatrix<double, ColMajor, int> stand-in (utils/Types.hpp:16-20): compressed arrays of a
    structurally symmetric matrix, so (outer, inner) are both the CSC and the CSR index arrays."""

    def __init__(self, n: int, outer, inner, values, resident=None):
        self.n, self.outer, self.inner, self.values = n, outer, inner, values
        # (engine, matrix generation) when this object describes the matrix that engine holds on the device right now:
        # a LinearSolver handed such a matrix solves with the device copy instead of uploading the values again.  Code
        # that changes the values (adds another form's Hessian, a regularisation ...) must drop the tag: use
        # add_diagonal / add, or set .resident = None.
        self.resident = resident

    def add_diagonal(self, d):
        """A += diag(d) (d scalar or vector): e.g. the regularisation `hessian += reg_weight * sparse_identity` of
        solver/SparseNewtonDescentSolver.tpp:116-121, or a lumped mass / penalty form."""
        import scipy.sparse as sp
        self._assign(self.to_scipy_csc() + sp.diags(np.broadcast_to(np.asarray(d, dtype=np.float64), (self.n,))))

    def add(self, other):
        """A += B, the `hessian += tmp` of solver/FullNLProblem.cpp:131 (sum of the forms' second derivatives)."""
        self._assign(self.to_scipy_csc() + other.to_scipy_csc())

    def _assign(self, m):
        m = m.tocsc()
        m.sort_indices()
        self.outer, self.inner, self.values = m.indptr.astype(np.int64), m.indices.astype(np.int32), m.data.astype(np.float64)
        self.resident = None

    def to_scipy_csc(self):
        import scipy.sparse as sp
        return sp.csc_matrix((self.values, self.inner, self.outer), shape=(self.n, self.n))

    def rows(self):
        return self.n

    def cols(self):
        return self.n

    def non_zeros(self):
        return int(self.inner.size)

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.csr_matrix((self.values, self.inner, self.outer), shape=(self.n, self.n))


class SpareMatrixCache:
    """utils::SpareMatrixCache (utils/MatrixUtils.hpp:39-100) as seen by a caller of assemble_hessian: it owns
    the reusable pattern.  Here the pattern lives on the device; the object records that it has been built."""

    def __init__(self):
        self.engine = None
        self.outer = self.inner = None

    def non_zeros(self):
        return 0 if self.inner is None else int(self.inner.size)


class _AssemblerBase:
    def __init__(self, formulation: str, engine: Engine | None = None, device: int = 0):
        if formulation not in MODEL_BY_NAME:
            raise ValueError(f"unsupported formulation {formulation}")
        self.formulation = formulation
        self.engine = engine or Engine(device)
        self.lam, self.mu = 1.0, 1.0
        self.size = 3
        self._bound = None

    def set_size(self, size: int):
        """2 for triangle meshes (is_volume = False), 3 otherwise; call before add_multimaterial, as the reference does."""
        if size not in (2, 3):
            raise ValueError("elasticity is implemented for size 2 (triangles) and 3")
        self.size = size

    def add_multimaterial(self, index: int, params: dict):
        """LameParameters::add_multimaterial (utils/ElasticityUtils.cpp:519-560) for uniform parameters given
        as (lambda, mu) or (E, nu)."""
        if "lambda" in params and "mu" in params:
            self.lam, self.mu = float(params["lambda"]), float(params["mu"])
        elif "E" in params and "nu" in params:
            E, nu = float(params["E"]), float(params["nu"])
            # convert_to_lambda with is_volume = size() == 3 (utils/ElasticityUtils.cpp:203-214)
            self.lam = (E * nu) / ((1.0 + nu) * (1.0 - 2.0 * nu)) if self.size == 3 else (nu * E) / (1.0 - nu * nu)
            self.mu = E / (2.0 * (1.0 + nu))
        self._bound = None

    def _bind(self, mesh, is_volume=True):
        from . import mesh as _M
        dim = _M.KIND_DIM[mesh.kind]
        if bool(is_volume) != (dim == 3):
            raise ValueError("is_volume does not match the mesh (False for triangle meshes)")
        if self.formulation != "Laplacian" and self.size != dim:
            raise ValueError("the local assembler's size() does not match the mesh")
        key = (id(mesh), self.lam, self.mu)
        if self._bound != key:
            self.engine.set_mesh(mesh)
            self.engine.set_material(MODEL_BY_NAME[self.formulation], self.lam, self.mu)
            self.engine.build_pattern()
            self._bound = key


class Assembler(_AssemblerBase):
    """assembler::Assembler<LocalAssembler> for Laplacian / LinearElasticity."""

    def assemble(self, is_volume: bool, n_basis: int, bases, gbases=None, cache=None, is_mass: bool = False) -> StiffnessMatrix:
        if is_mass:
            raise NotImplementedError("is_mass = True belongs to Assembler<Mass>: Engine.assemble_mass")
        mesh = bases
        if n_basis != mesh.n_nodes:
            raise ValueError("n_basis does not match the bases")
        self._bind(mesh, is_volume)
        self.engine.assemble_linear()
        rp, ci, v = self.engine.get_linear_csr()
        return StiffnessMatrix(int(rp.size - 1), rp, ci, v, resident=(self.engine, self.engine.matrix_generation()))


class NLAssembler(_AssemblerBase):
    """assembler::NLAssembler<LocalAssembler> for NeoHookean, SaintVenant (and LinearElasticity)."""

    def assemble(self, is_volume, bases, gbases, cache, dt, displacement, displacement_prev=None) -> float:
        self._bind(bases, is_volume)
        return self.engine.energy(np.asarray(displacement).reshape(-1))

    def assemble_grad(self, is_volume, n_basis, bases, gbases, cache, dt, displacement, displacement_prev=None):
        self._bind(bases, is_volume)
        return self.engine.gradient(np.asarray(displacement).reshape(-1)).reshape(-1, 1)

    def assemble_hessian(self, is_volume, n_basis, project_to_psd, bases, gbases, cache, dt, displacement,
                         displacement_prev, mat_cache: SpareMatrixCache) -> StiffnessMatrix:
        self._bind(bases, is_volume)
        if mat_cache.engine is not self.engine or mat_cache.outer is None:
            mat_cache.engine = self.engine
            mat_cache.outer, mat_cache.inner = self.engine.get_pattern()
        v = self.engine.hessian(np.asarray(displacement).reshape(-1), project_to_psd)
        return StiffnessMatrix(int(mat_cache.outer.size - 1), mat_cache.outer, mat_cache.inner, v,
                               resident=(self.engine, self.engine.matrix_generation()))


class LinearSolver:
    """polysolve::LinearSolver ("Eigen::ConjugateGradient"; "AMGCL" / "Hypre" = CG with this library's AMG) as
    SparseNewtonDescentSolver and State::solve_linear use it (solver/SparseNewtonDescentSolver.tpp:128-155,
    state/StateSolveLinear.cpp:27-29,51-54): analyze_pattern(A) / factorize(A) / solve(b, x) work on the matrix they are
    HANDED -- the reduced sum of all forms' Hessians, any size -- which is uploaded into a solver context of its own
    (pfx_solver_analyze / pfx_solver_factorize).  Only a matrix that still carries the tag of the engine holding it on the
    device (StiffnessMatrix.resident, unchanged since the assembly that produced it) is solved in place, without a copy
    crossing PCIe; the constrained dofs of such a full-size solve come from set_dirichlet()."""

    def __init__(self, engine: Engine | None, precond: str, solver: str = "Eigen::ConjugateGradient", device: int = 0):
        self.engine = engine          # the assembling engine (may be None: every matrix is then uploaded)
        self.device = device
        self.own = None               # solver context for matrices handed in by value
        self.active = None            # the context the next solve runs on
        self.block_size = 0           # pfx_solver_analyze's block_size argument (0: choose)
        self.node_positions = None    # for the AMG preconditioner of a handed-in block-3 matrix
        self.solver = solver or "Eigen::ConjugateGradient"
        self._amg = {}
        if self.solver in ("AMGCL", "Hypre"):
            # the slot of polysolve's AMG solvers: CG preconditioned by this library's own aggregation AMG (not AMGCL's
            # or Hypre's algorithm); their defaults tol 1e-10, 1000 iterations (input-spec.json:1381-1396, 1423-1434)
            self.precond = PRECOND_AMG
            self.max_iter, self.tolerance = 1000, 1e-10
        else:
            self.precond = PRECOND_JACOBI if precond in ("", "Eigen::DiagonalPreconditioner", "Jacobi") else PRECOND_IDENTITY
            self.max_iter, self.tolerance = 1000, 1e-12  # input-spec.json:1326-1336
        self._info = {}
        self._analyzed = None

    @staticmethod
    def create(solver: str, precond: str, engine: Engine | None = None) -> "LinearSolver":
        if solver not in ("Eigen::ConjugateGradient", "", "AMGCL", "Hypre"):
            raise RuntimeError(f"Unrecognized solver type: {solver}")
        return LinearSolver(engine, precond, solver)

    def close(self):
        if self.own is not None:
            self.own.close()
            self.own = None

    def name(self) -> str:
        return self.solver

    def set_parameters(self, params: dict):
        p = params.get(self.name(), params)
        if self.solver == "AMGCL":
            sp_, pp = p.get("solver", {}), p.get("precond", {})
            self.max_iter = int(sp_.get("maxiter", self.max_iter))
            self.tolerance = float(sp_.get("tol", self.tolerance))
            amg = {}
            if "ncycle" in pp:
                amg["ncycle"] = int(pp["ncycle"])
            if "max_levels" in pp:
                amg["max_levels"] = int(pp["max_levels"])
            if "degree" in pp.get("relax", {}):
                amg["relax_degree"] = int(pp["relax"]["degree"])
            if "power_iters" in pp.get("relax", {}):
                amg["power_iters"] = int(pp["relax"]["power_iters"])
            self._amg = amg
            for ctx in (self.engine, self.own):
                if amg and ctx is not None:
                    ctx.amg_set_params(**amg)
            return
        self.max_iter = int(p.get("max_iter", self.max_iter))
        self.tolerance = float(p.get("tolerance", self.tolerance))

    def _is_resident(self, A) -> bool:
        r = getattr(A, "resident", None)
        return (A is not None and r is not None and self.engine is not None and r[0] is self.engine
                and r[1] == self.engine.matrix_generation())

    def analyze_pattern(self, A, precond_num: int = 0):
        """A: StiffnessMatrix.  precond_num is polysolve's hint for block preconditioners (unused by CG)."""
        if A is None:
            raise ValueError("analyze_pattern needs the matrix (polysolve::LinearSolver::analyzePattern(A, precond_num))")
        if self._is_resident(A):
            self.active, self._analyzed = self.engine, ("resident", A.rows())
            return
        if self.own is None:
            self.own = Engine(self.device)
            if self._amg:
                self.own.amg_set_params(**self._amg)
        self.own.solver_analyze(A.rows(), A.outer, A.inner, self.block_size)
        if self.node_positions is not None:
            self.own.solver_set_node_positions(self.node_positions)
        self.active, self._analyzed = self.own, ("own", A.rows(), int(A.inner.size))

    def factorize(self, A):
        """Raises RuntimeError where polysolve's factorize does (the Newton solver catches it and regularises,
        solver/SparseNewtonDescentSolver.tpp:135-150)."""
        if self._analyzed is None:
            raise RuntimeError("factorize called before analyzePattern")
        if self._analyzed[0] == "resident":
            if not self._is_resident(A) or A.rows() != self._analyzed[1]:
                # the caller changed the matrix after analyzePattern: it is no longer the device copy
                self.analyze_pattern(A)
            else:
                return
        if A.rows() != self._analyzed[1] or int(A.inner.size) != self._analyzed[2]:
            raise RuntimeError("factorize called with a matrix of another pattern than analyzePattern")
        try:
            self.own.solver_factorize(A.values)
        except PfxError as e:
            if e.code == PFX_ERR_NUMERIC:
                raise RuntimeError(f"Factorization failed: {e}") from e
            raise

    def set_dirichlet(self, sorted_dofs):
        """Full-size solves on the resident matrix: the constrained dofs (the reduced matrix is never formed)."""
        (self.active or self.engine).set_dirichlet(sorted_dofs)

    def solve(self, b, x):
        if self.active is None:
            raise RuntimeError("solve called before analyzePattern / factorize")
        sol, it, err = self.active.cg(b, x, self.precond, self.max_iter, self.tolerance)
        self._info = {"solver_iter": it, "solver_error": err, "solver_tol": self.tolerance, "solver_maxiter": self.max_iter}
        if not np.isfinite(err):
            raise RuntimeError("CG broke down (non-finite residual)")
        x[...] = sol
        return x

    def get_info(self) -> dict:
        return dict(self._info)
