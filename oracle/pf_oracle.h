/* pf_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement, in plain C, of the reference's (xDarkLemon/polyfem) element-assembly + CG hot path.
 * It is the parity checker for the HIP library and the timed "cpu_baseline"; nothing in the product path
 * (polyfem_amd/, include/) may call, link or import it.
 *
 * Every function cites the reference file:line it follows (paths relative to the reference root).
 * Arithmetic that lives in un-vendored third-party code (Eigen 3.4.0 dense 3x3 inverse/determinant,
 * Eigen::ConjugateGradient, polysolve dirichlet_solve) is restated from its published algorithm and is
 * marked "parity unpinned": the reference holds no golden vectors for it.
 */
#ifndef PF_ORACLE_H
#define PF_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* element kinds: number of local bases 4 / 10 / 8 / 27, and the 2-D triangles 3 / 6 */
enum { PFO_P1TET = 0, PFO_P2TET = 1, PFO_Q1HEX = 2, PFO_Q2HEX = 3, PFO_P1TRI = 4, PFO_P2TRI = 5 };
/* local assemblers */
enum { PFO_LAPLACIAN = 0, PFO_LINEAR_ELASTICITY = 1, PFO_NEOHOOKEAN = 2, PFO_SAINT_VENANT = 3, PFO_HOOKE_LINEAR = 4, PFO_MULTI_MODEL = 5 };

#define PFO_MAX_PTS 125
#define PFO_MAX_LOC 27

typedef struct
{
	int n_pts;
	double pts[PFO_MAX_PTS][3];
	double w[PFO_MAX_PTS];
} pfo_quad;

/* per-element evaluation (reference: assembler::ElementAssemblyValues) */
typedef struct
{
	int n_pts, n_loc;
	int dim;                                       /* 3, or 2 for triangles: third components are zero, jac_it holds the 2x2 in its upper left */
	double val[PFO_MAX_LOC][PFO_MAX_PTS];          /* basis value at each point */
	double grad[PFO_MAX_LOC][PFO_MAX_PTS][3];      /* reference-space gradient  */
	double grad_t_m[PFO_MAX_LOC][PFO_MAX_PTS][3];  /* physical gradient          */
	double det[PFO_MAX_PTS];
	double jac_it[PFO_MAX_PTS][9];                 /* row-major 3x3 (J^{-1})^T   */
	pfo_quad quad;
} pfo_vals;

typedef struct
{
	int kind;      /* basis kind                       */
	int gkind;     /* geometry basis kind (P1TET for tets, Q1HEX for hexes unless isoparametric) */
	int64_t n_el;
	int64_t n_nodes;  /* number of global bases (scalar) */
	int64_t n_gnodes;
	const int32_t *conn;   /* n_el x n_loc  global basis index per local basis (Local2Global.index) */
	const int32_t *gconn;  /* n_el x n_gloc geometry node index                                      */
	const double *xyz;     /* n_gnodes x dim geometry node positions (Local2Global.node): 3 per node, 2 for triangles */
	int quad_order;        /* -1: reference default max(-1, 2(p-1)+1)                                 */
} pfo_mesh;

typedef struct
{
	int model;
	int dim;          /* problem dimension (3; 2 for triangle meshes) ; block size is 1 for the Laplacian, dim otherwise */
	double lambda, mu; /* uniform Lame parameters */
	const double *lambda_e, *mu_e; /* optional per-element values (LameParameters with one entry per element,
	                                  utils/ElasticityUtils.cpp:490-517); NULL = uniform */
	const double *tensor;          /* PFO_HOOKE_LINEAR, and PFO_SAINT_VENANT when not NULL (SaintVenantElasticity.cpp:64-68): the symmetric 6 x 6 elasticity tensor, row-major, Voigt order
	                                  xx yy zz yz xz xy (ElasticityTensor, utils/ElasticityUtils.cpp:298-413) */
	int per_point;                 /* > 0: lambda_e / mu_e hold one value per (element, quadrature point), per_point =
	                                  points per element (LameParameters::lambda_mu is evaluated at every point,
	                                  utils/ElasticityUtils.cpp:490-517); 0: one value per element */
	const int32_t *model_e;        /* PFO_MULTI_MODEL: the model of every element (MultiModel::multi_material_models_,
	                                  assembler/MultiModel.cpp:42-96): LINEAR_ELASTICITY, NEOHOOKEAN or SAINT_VENANT */
} pfo_material;

/* compressed column storage with int32 indices = Eigen::SparseMatrix<double, ColMajor, int> */
typedef struct
{
	int32_t n;
	int64_t nnz;
	int32_t *outer; /* n+1 */
	int32_t *inner; /* nnz, ascending inside a column */
	double *values;
} pfo_csc;

void pfo_csc_free(pfo_csc *m);

int pfo_n_loc(int kind);
int pfo_order_of(int kind);
double pfo_lambda_from_E_nu(double E, double nu); /* 3-D */
double pfo_lambda_from_E_nu_2d(double E, double nu);
int pfo_kind_dim(int kind);
double pfo_mu_from_E_nu(double E, double nu);

int pfo_tet_quadrature(int order, pfo_quad *q);
int pfo_tri_quadrature(int order, pfo_quad *q);
int pfo_line_quadrature(int order, pfo_quad *q);
int pfo_hex_quadrature(int order, pfo_quad *q);
int pfo_element_quadrature(int kind, int quad_order, pfo_quad *q);

void pfo_basis_value(int kind, int local_index, const double *pt, double *val);
void pfo_basis_grad(int kind, int local_index, const double *pt, double *g3);
void pfo_basis_nodes(int kind, double *nodes /* n_loc x 3 */);
void pfo_q2_tensor_basis(int local_index, const double *pt, double *val, double *g3); /* cross-check of the Q2 tables */

double pfo_determinant3(const double *m /* row-major 3x3 */);

void pfo_compute_vals(const pfo_mesh *mesh, int64_t e, const pfo_quad *quad, pfo_vals *vals);

/* local assemblers */
void pfo_local_laplacian(const pfo_vals *v, int i, int j, const double *da, double *res1);
void pfo_local_linear_elasticity(const pfo_vals *v, int i, int j, const double *da, double lambda, double mu, double *res9);
double pfo_local_neohookean_energy(const pfo_vals *v, const double *da, const double *local_disp, double lambda, double mu);
void pfo_local_neohookean_grad(const pfo_vals *v, const double *da, const double *local_disp, double lambda, double mu, double *G);
void pfo_local_neohookean_hessian(const pfo_vals *v, const double *da, const double *local_disp, double lambda, double mu, double *H);
double pfo_local_linear_elasticity_energy(const pfo_vals *v, const double *da, const double *local_disp, double lambda, double mu);
void pfo_local_linear_elasticity_grad(const pfo_vals *v, const double *da, const double *local_disp, double lambda, double mu, double *G);
void pfo_local_linear_elasticity_hessian(const pfo_vals *v, const double *da, double lambda, double mu, double *H);
/* 2-D (triangles): local_disp is n_loc x 2, G has n_loc*2 entries, H is (n_loc*2)^2 row-major */
double pfo_local_neohookean_energy_2d(const pfo_vals *v, const double *da, const double *local_disp, double lambda, double mu);
void pfo_local_neohookean_grad_2d(const pfo_vals *v, const double *da, const double *local_disp, double lambda, double mu, double *G);
void pfo_local_neohookean_hessian_2d(const pfo_vals *v, const double *da, const double *local_disp, double lambda, double mu, double *H);

/* global assemblers; n_threads > 1 uses thread-local storages merged like the reference does */
int pfo_assemble_linear(const pfo_mesh *mesh, const pfo_material *mat, int n_threads, pfo_csc *out);
double pfo_assemble_energy(const pfo_mesh *mesh, const pfo_material *mat, const double *x, int n_threads);
int pfo_assemble_gradient(const pfo_mesh *mesh, const pfo_material *mat, const double *x, int n_threads, double *rhs);
int pfo_assemble_hessian(const pfo_mesh *mesh, const pfo_material *mat, const double *x, int n_threads, pfo_csc *out);

/* Mass matrix (assembler/Mass.cpp:5-20 through Assembler<Mass>::assemble with is_mass = true, Assembler.cpp:135-139):
 * mass quadrature of order max(mass_quadrature_order = -1, 2p+1) (basis/FEBasis3d.cpp:1468,1507), rho * phi_i * phi_j on
 * the diagonal of the size x size block, same 1e-30 drop rule.  density_e: optional per-element density. */
int pfo_mass_quadrature(int kind, pfo_quad *q);
int pfo_assemble_mass(const pfo_mesh *mesh, int size, double density, const double *density_e, int n_threads, pfo_csc *out);
/* Body-force right-hand side (assembler/RhsAssembler.cpp:55-98) for a constant force density f (size components):
 * rhs[idx*size+d] += sum_q f_d rho det w phi_i(q), stiffness quadrature, serial element loop. */
int pfo_assemble_rhs(const pfo_mesh *mesh, int size, const double *f, double density, const double *density_e, double *rhs);
/* the same with the force density given per quadrature point (n_el x n_pts x size), and the physical positions of those
 * points (n_el x n_pts x 3; returns n_pts) for the caller to evaluate its force function at */
int pfo_assemble_rhs_pointwise(const pfo_mesh *mesh, int size, const double *force_pts, double density, const double *density_e, double *rhs);
/* Density evaluated per quadrature point (Density::operator(), utils/ElasticityUtils.cpp:561-570): mass matrix with one
 * density per (element, point of the mass rule); right-hand side with one per (element, point of the stiffness rule) and a
 * constant (f) or pointwise (force_pts) force; positions of the mass rule's points (returns n_pts). */
int pfo_assemble_mass_pointwise(const pfo_mesh *mesh, int size, const double *density_pts, int n_threads, pfo_csc *out);
int pfo_assemble_rhs_density_pointwise(const pfo_mesh *mesh, int size, const double *f, const double *force_pts, const double *density_pts, double *rhs);
int pfo_mass_quadrature_points(const pfo_mesh *mesh, double *pts);
int pfo_quadrature_points(const pfo_mesh *mesh, double *pts);

/* ipc::project_to_psd (ipc-toolkit, un-vendored; call site assembler/Assembler.cpp:554-555): symmetric eigendecomposition,
 * A returned unchanged if its smallest eigenvalue is >= 0, otherwise V max(D,0) V^T.  The eigenvalues come from cyclic
 * Jacobi rotations here (the library uses Eigen::SelfAdjointEigenSolver): same projection up to round-off, parity
 * unpinned.  a: n x n row-major, in place. */
void pfo_project_to_psd(double *a, int n);
/* NLAssembler::assemble_hessian with project_to_psd = true */
int pfo_assemble_hessian_psd(const pfo_mesh *mesh, const pfo_material *mat, const double *x, int n_threads, pfo_csc *out);

/* SpareMatrixCache restatement */
typedef struct pfo_cache pfo_cache;
pfo_cache *pfo_cache_create(int64_t size);
pfo_cache *pfo_cache_clone(const pfo_cache *other);
void pfo_cache_destroy(pfo_cache *c);
void pfo_cache_set_zero(pfo_cache *c);
void pfo_cache_add_value(pfo_cache *c, int e, int i, int j, double value);
void pfo_cache_prune(pfo_cache *c);
void pfo_cache_add(pfo_cache *c, const pfo_cache *o); /* operator+= */
int pfo_cache_get_matrix(pfo_cache *c, int compute_mapping, pfo_csc *out);
int64_t pfo_cache_mapping_size(const pfo_cache *c);
int pfo_assemble_hessian_cached(const pfo_mesh *mesh, const pfo_material *mat, const double *x, pfo_cache *cache, pfo_csc *out);

/* Dirichlet reduction */
int pfo_full_to_reduced_matrix(const pfo_csc *full, const int32_t *removed_vars, int64_t n_removed, pfo_csc *reduced);

/* sparse helpers */
void pfo_spmv(const pfo_csc *A, const double *x, double *y);

/* Eigen::ConjugateGradient with DiagonalPreconditioner (parity unpinned) */
int pfo_cg(const pfo_csc *A, const double *b, double *x, int precond_jacobi, int max_iter, double tol, int *iters, double *error);

/* polysolve::dirichlet_solve system modification (parity unpinned): A,b modified in place */
int pfo_dirichlet_modify(pfo_csc *A, double *b, const int32_t *boundary_nodes, int64_t n_boundary);

/* Newton solve of  min_x E(x) - rhs.x  with the dofs in boundary_nodes held at their values in x
 * (cppoptlib::NonlinearSolver::minimize + SparseNewtonDescentSolver + BacktrackingLineSearch on NLProblem).
 * cppoptlib's checkConvergence and polysolve's CG are un-vendored: parity unpinned. */
typedef struct
{
	double f_delta;             /* solver/nonlinear/f_delta, default 1e-10              */
	double grad_norm;           /* solver/nonlinear/grad_norm, default 1e-8             */
	double first_grad_norm_tol; /* default 1e-10                                         */
	double use_grad_norm_tol;   /* line_search/use_grad_norm_tol, default 1e-4          */
	int max_iterations;         /* default 1000                                          */
	int use_grad_norm;          /* default 1                                             */
	int relative_gradient;      /* default 0                                             */
	int line_search;            /* 1 = "backtracking" (default), 0 = none (rate 1)       */
	int cg_precond_jacobi;      /* linear solver: CG, Jacobi or identity preconditioner */
	int cg_max_iter;            /* default 1000                                          */
	double cg_tol;              /* default 1e-12                                         */
} pfo_newton_params;

enum
{
	PFO_NEWTON_CONTINUE = 0,
	PFO_NEWTON_ITERATION_LIMIT = 1,
	PFO_NEWTON_F_DELTA = 2,
	PFO_NEWTON_GRAD_NORM = 3,
	PFO_NEWTON_NOT_STARTED = 4, /* first gradient below first_grad_norm_tol */
	PFO_NEWTON_FAILED = 5       /* nan energy / gradient, or line search failed on gradient descent */
};

typedef struct
{
	int status;
	int iterations;
	int line_search_iterations; /* energy / gradient evaluations spent in the line searches */
	int descent_strategy;       /* strategy of the last accepted step */
	int64_t cg_iterations;
	double energy, grad_norm;
} pfo_newton_info;

void pfo_newton_default_params(pfo_newton_params *p);
int pfo_newton_solve(const pfo_mesh *mesh, const pfo_material *mat, const double *rhs, const int32_t *boundary_nodes, int64_t n_boundary,
					 double *x, const pfo_newton_params *params, int n_threads, pfo_newton_info *info);
/* The same solve for one implicit time step: objective  weight * (E(x) - rhs.x) + 1/2 (x - x_tilde)^T M (x - x_tilde),
 * i.e. ElasticForm and BodyForm weighted with the integrator's acceleration_scaling (state/StateSolveNonlinear.cpp:78-84)
 * plus InertiaForm (solver/forms/InertiaForm.cpp:14-31).  mass == NULL gives pfo_newton_solve. */
int pfo_newton_solve_inertia(const pfo_mesh *mesh, const pfo_material *mat, const double *rhs, const int32_t *boundary_nodes, int64_t n_boundary,
							 double *x, const pfo_newton_params *params, int n_threads, pfo_newton_info *info,
							 const pfo_csc *mass, const double *x_tilde, double weight);

#ifdef __cplusplus
}
#endif
#endif
