/* polyfem_amd.h -- C ABI of the MI355X (gfx950) element-assembly + CG engine.
 *
 * This is the drop-in boundary for PolyFEM's assembly / Krylov hot path.  The reference has no FFI for it;
 * the seam is the C++ member templates of `polyfem::assembler::Assembler<L>` / `NLAssembler<L>`
 * (src/polyfem/assembler/Assembler.hpp:26-113) and the polysolve `LinearSolver` interface
 * (call sites src/polyfem/solver/SparseNewtonDescentSolver.tpp:128-155, src/polyfem/state/StateSolveLinear.cpp:14-68).
 * Each entry point names the reference interface it replaces.  A C++ shim with the reference's own signatures
 * sits on top of this header (include/polyfem_amd/assembler.hpp).
 *
 * Conventions
 *   - plain pointers and sizes; no exceptions cross the ABI; every call returns PFX_OK (0) or a negative code,
 *     pfx_last_error() gives the text.
 *   - one context per GPU and per host thread; the context owns all device memory.
 *   - pointers are HOST pointers unless the function name ends in `_dev`; `_dev` functions take device
 *     pointers, enqueue on the context's stream and do not synchronise.
 *   - vectors are node-major interleaved: x[node*dim + c] (src/polyfem/assembler/NeoHookeanElasticity.cpp:380).
 *   - indices are int32 for columns (reference: utils/Types.hpp:16-20) and int64 for row offsets.
 */
#ifndef POLYFEM_AMD_H
#define POLYFEM_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pfx_ctx pfx_ctx;
typedef int pfx_status;

enum
{
	PFX_OK = 0,
	PFX_ERR_ARG = -1,     /* bad argument / call order          */
	PFX_ERR_HIP = -2,     /* HIP runtime error                  */
	PFX_ERR_UNSUPPORTED = -3,
	PFX_ERR_NUMERIC = -4, /* non-finite energy, J <= 0 ...      */
	PFX_ERR_SIZE = -5     /* index width / capacity exceeded    */
};

/* element kinds: local node order of autogen/auto_p_bases.cpp:1080-1087,1328-1341 and
 * autogen/auto_q_bases.cpp:1262-1275,1873-1903; geometry map is P1 (tets) / Q1 (hexes).
 * 2-D problems: triangles with the local node order of autogen/auto_p_bases.cpp:138-144, 275-284 and a P1 geometry
 * map (the (3,2) and (6,2) instantiations of NeoHookeanElasticity.cpp:47-115); dim = 2, two coordinates per geometry
 * node, two displacement components per node.  They support every local assembler below (the size-2 branches of
 * the reference; an elasticity tensor is then 3 x 3), uniform / per-element / per-point parameters, project_to_psd, mass
 * matrix and right-hand side, the Jacobi- and AMG-preconditioned CG, the Newton solver with its inertia and
 * augmented-Lagrangian terms, the implicit-Euler step and the partitioned solve (pfx_dist_*). */
enum
{
	PFX_P1TET = 0,
	PFX_P2TET = 1,
	PFX_Q1HEX = 2,
	PFX_Q2HEX = 3,
	PFX_P1TRI = 4,
	PFX_P2TRI = 5
};

/* local assemblers: Laplacian.cpp:7-18, LinearElasticity.cpp:21-54, NeoHookeanElasticity.cpp:250-565,
 * SaintVenantElasticity.cpp:104-143,215-312 (nonlinear interface only; isotropic elasticity tensor from lambda, mu as
 * ElasticityTensor::set_from_lambda_mu builds it, utils/ElasticityUtils.cpp:372-413) */
enum
{
	PFX_LAPLACIAN = 0,
	PFX_LINEAR_ELASTICITY = 1,
	PFX_NEOHOOKEAN = 2,
	PFX_SAINT_VENANT = 3,
	PFX_HOOKE_LINEAR = 4, /* HookeLinearElasticity.cpp:75-190: linear assembler with a general elasticity tensor */
	PFX_MULTI_MODEL = 5   /* MultiModel.cpp:42-96: LinearElasticity, NeoHookean or SaintVenant chosen per element */
};

enum
{
	PFX_PRECOND_IDENTITY = 0,
	PFX_PRECOND_JACOBI = 1, /* Eigen::DiagonalPreconditioner, polysolve's default for Eigen::ConjugateGradient */
	PFX_PRECOND_AMG = 2     /* aggregation multigrid (see the AMG section below): rigid-body modes for block sizes 3 (six) and 2 (three, in the plane), constants for block size 1 */
};

/* ---- context ------------------------------------------------------------------------------------------------ */

/* Creates a context on HIP device `device`.  Replaces State::sending_data_to_GPU's role of owning the device
 * copies (src/polyfem/state/StateSolveNonlinear.cpp:169-300). */
pfx_status pfx_create(pfx_ctx **ctx, int device);
void pfx_destroy(pfx_ctx *ctx);
const char *pfx_last_error(const pfx_ctx *ctx);
/* Use an existing hipStream_t (e.g. the caller's RCCL/compute stream); NULL = the context's own stream. */
pfx_status pfx_set_stream(pfx_ctx *ctx, void *hip_stream);
pfx_status pfx_synchronize(pfx_ctx *ctx);

/* ---- mesh / material ------------------------------------------------------------------------------------------ */

/* Flattened std::vector<ElementBases> bases / gbases (basis/ElementBases.hpp:16-104): conn[e*n_loc + j] =
 * bases[e].bases[j].global()[0].index, gconn[e*n_gloc + j] = geometry node of gbases[e].bases[j],
 * geom_xyz[dim*g + c] = Local2Global.node (dim = 3; 2 for the triangle kinds).  Only conforming bases
 * (global().size()==1, val==1) are accepted.
 * quad_order: -1 = the reference default max(-1, 2(p-1)+1) (basis/FEBasis3d.cpp:1467,1506, basis/FEBasis2d.cpp:797). */
pfx_status pfx_set_mesh(pfx_ctx *ctx, int elem_kind, int64_t n_el, int64_t n_nodes, int64_t n_gnodes,
						const int32_t *conn, const int32_t *gconn, const double *geom_xyz, int quad_order);

/* Uniform Lame parameters (LameParameters::lambda_mu with constant expressions, utils/ElasticityUtils.cpp:490-517).
 * dim = 3 (2 for the triangle kinds).  The block size of the system is 1 for the Laplacian and dim otherwise. */
pfx_status pfx_set_material(pfx_ctx *ctx, int model, int dim, double lambda, double mu);
/* One (lambda, mu) per element: LameParameters holding one entry per element id, which is how the reference stores
 * multi-material bodies (add_multimaterial, utils/ElasticityUtils.cpp:490-517, 521-560; AssemblerUtils.cpp:604-672).
 * Call after pfx_set_mesh; a later pfx_set_mesh or pfx_set_material drops the arrays. */
pfx_status pfx_set_material_per_element(pfx_ctx *ctx, int model, int dim, const double *lambda, const double *mu,
										int64_t n_el);
/* The same with one (lambda, mu) per element AND quadrature point of the stiffness rule (n_el x n_pts values each, point
 * fastest): LameParameters::lambda_mu is evaluated at every point in the reference (utils/ElasticityUtils.cpp:490-517;
 * space-dependent materials).  pfx_quadrature_points gives the positions to evaluate them at. */
pfx_status pfx_set_material_per_point(pfx_ctx *ctx, int model, int dim, const double *lambda, const double *mu, int64_t n_el, int n_pts);
/* HookeLinearElasticity with its `elasticity_tensor`: the symmetric 6 x 6 tensor in Voigt order xx yy zz yz xz xy,
 * row-major, as ElasticityTensor holds it after set_from_entries / set_from_lambda_mu / set_from_young_poisson
 * (utils/ElasticityUtils.cpp:298-413).  dim = 2 (triangle kinds): the symmetric 3 x 3 tensor in Voigt order xx yy xy,
 * 9 values row-major (ElasticityTensor::resize(2), :264-274; HookeLinearElasticity.cpp:92-122).  Selects the model
 * PFX_HOOKE_LINEAR (pfx_assemble_linear only). */
pfx_status pfx_set_elasticity_tensor(pfx_ctx *ctx, int dim, const double *tensor36);
/* SaintVenantElasticity with the material's `elasticity_tensor` entries instead of (lambda, mu)
 * (SaintVenantElasticity::add_multimaterial, assembler/SaintVenantElasticity.cpp:42-69; the stress of a point is
 * tensor * Voigt(Green strain), :77-86, :281-307).  Same layout as pfx_set_elasticity_tensor (6 x 6, or 3 x 3 for dim = 2); selects
 * PFX_SAINT_VENANT for the nonlinear interface (energy, gradient, Hessian, Newton, time steps).  A later
 * pfx_set_material call returns to (lambda, mu). */
pfx_status pfx_set_saint_venant_tensor(pfx_ctx *ctx, int dim, const double *tensor36);
/* MultiModel (assembler/MultiModel.cpp:42-96, `multi_material_models_[el_id]`): after pfx_set_material(_per_element) with
 * any of the three elastic models, gives every element its own model -- PFX_LINEAR_ELASTICITY, PFX_NEOHOOKEAN or
 * PFX_SAINT_VENANT -- for the nonlinear interface (energy, gradient, Hessian, Newton, time steps).  The Lame parameters
 * stay those of pfx_set_material(_per_element).  A later pfx_set_material call returns to one model. */
pfx_status pfx_set_element_models(pfx_ctx *ctx, const int32_t *model_per_element, int64_t n_el);

/* ---- sparsity pattern (utils::SpareMatrixCache, utils/MatrixUtils.cpp:108-264) --------------------------------- */

/* Builds, on the device, the structural pattern of the global matrix (every (n_loc*dim)^2 entry of every
 * element, which is what NLAssembler::assemble_hessian inserts, Assembler.cpp:573-616) and the element ->
 * entry map that later assemblies reuse.  nnz receives the number of scalar entries. */
pfx_status pfx_build_pattern(pfx_ctx *ctx, int64_t *nnz);
/* The pattern as scalar CSR (== Eigen's CSC arrays of the structurally symmetric matrix): row_ptr has
 * n_nodes*dim+1 entries, col_idx nnz entries ascending within a row. */
pfx_status pfx_get_pattern(pfx_ctx *ctx, int64_t *row_ptr, int32_t *col_idx);
/* Current values in the order of pfx_get_pattern (explicit zeros included). */
pfx_status pfx_get_values(pfx_ctx *ctx, double *values);

/* ---- assembly ------------------------------------------------------------------------------------------------- */

/* Assembler<L>::assemble (Assembler.cpp:100-317) for L = Laplacian / LinearElasticity: stiffness values are
 * left on the device in the structural pattern; entries the reference never inserts (|local value| < 1e-30 in
 * every element, Assembler.cpp:170-173) are remembered in a mask. */
pfx_status pfx_assemble_linear(pfx_ctx *ctx);
/* The matrix exactly as the reference's setFromTriplets leaves it: masked entries dropped. Call with
 * row_ptr == NULL to get only *nnz. */
pfx_status pfx_get_linear_csr(pfx_ctx *ctx, int64_t *nnz, int64_t *row_ptr, int32_t *col_idx, double *values);

/* NLAssembler<L>::assemble -> energy (Assembler.cpp:638-678) */
pfx_status pfx_energy(pfx_ctx *ctx, const double *u, double *energy);
/* NLAssembler<L>::assemble_grad (Assembler.cpp:427-504); g has n_nodes*dim entries */
pfx_status pfx_gradient(pfx_ctx *ctx, const double *u, double *g);
/* NLAssembler<L>::assemble_hessian (Assembler.cpp:506-636); values (may be NULL = keep on device) in
 * pfx_get_pattern order.  project_to_psd != 0 projects every element Hessian onto the positive semi-definite cone
 * before it is added (ipc::project_to_psd at assembler/Assembler.cpp:554-555). */
pfx_status pfx_hessian(pfx_ctx *ctx, const double *u, int project_to_psd, double *values);

/* Device-resident variants: d_u, d_energy (1 double), d_g are device pointers. */
pfx_status pfx_energy_dev(pfx_ctx *ctx, const double *d_u, double *d_energy);
pfx_status pfx_gradient_dev(pfx_ctx *ctx, const double *d_u, double *d_g);
pfx_status pfx_hessian_dev(pfx_ctx *ctx, const double *d_u);
pfx_status pfx_hessian_psd_dev(pfx_ctx *ctx, const double *d_u); /* project_to_psd = true */

/* Energy evaluation that allocates nothing.  No counterpart in the reference: it is for callers that cannot allocate
 * device memory at the call site -- code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc.  Same result as pfx_energy_dev (same kernels, same fixed summation order), but the per-element energies
 * and the partial sums live in a static device array of the library (8 MiB per GPU, held until the process ends)
 * instead of buffers of the context, and the call needs only pfx_set_mesh + pfx_set_material (no pattern).  Limit:
 * n_el <= 1048576, larger meshes are rejected with PFX_ERR_SIZE.  Every call on a GPU shares that GPU's array: calls
 * that overlap on one GPU (two contexts / streams) race; calls on one stream do not. */
pfx_status pfx_energy_dev_noalloc(pfx_ctx *ctx, const double *d_u, double *d_energy);

/* ---- Dirichlet handling --------------------------------------------------------------------------------------- */

/* Sorted list of constrained dofs.  Afterwards the solver works on the reduced system of
 * utils::full_to_reduced_matrix (utils/MatrixUtils.cpp:408-469) without re-building a matrix. n == 0 clears. */
pfx_status pfx_set_dirichlet(pfx_ctx *ctx, const int32_t *sorted_dofs, int64_t n);
/* The reduced matrix itself (for hand-back / parity); row_ptr == NULL returns only *nnz and *n_reduced. */
pfx_status pfx_get_reduced_csr(pfx_ctx *ctx, int64_t *n_reduced, int64_t *nnz, int64_t *row_ptr, int32_t *col_idx, double *values);

/* ---- Krylov solve (polysolve::LinearSolver, solver "Eigen::ConjugateGradient") -------------------------------- */

/* y = A x on the current device matrix (full size, Dirichlet rows/cols act as identity when set). */
pfx_status pfx_spmv(pfx_ctx *ctx, const double *x, double *y);
pfx_status pfx_spmv_dev(pfx_ctx *ctx, const double *d_x, double *d_y);

/* LinearSolver::analyzePattern + factorize + solve(b, x) for the CG solver: x_inout holds the initial guess
 * (solveWithGuess) and receives the solution.  b and x are FULL size; constrained dofs are ignored on input
 * and returned as 0.  Stopping rule of Eigen::ConjugateGradient: ||r||^2 < max(tol^2 ||b||^2, DBL_MIN);
 * iters/rel_res are what getInfo() reports ("solver_iter", "solver_error"). */
pfx_status pfx_cg(pfx_ctx *ctx, const double *b, double *x_inout, int precond, int max_iter, double tol,
				  int *iters, double *rel_res);
pfx_status pfx_cg_dev(pfx_ctx *ctx, const double *d_b, double *d_x_inout, int precond, int max_iter, double tol,
					  int *iters, double *rel_res);

/* polysolve::dirichlet_solve (call site state/StateSolveLinear.cpp:27-29): b holds the Dirichlet values at
 * the constrained dofs set with pfx_set_dirichlet; solves the modified system and returns the full-size x. */
pfx_status pfx_dirichlet_solve(pfx_ctx *ctx, const double *b, double *x, int precond, int max_iter, double tol,
							   int *iters, double *rel_res);

/* ---- linear solver for a matrix the caller hands in ------------------------------------------------------------------------
 *
 * polysolve::LinearSolver::analyzePattern(A, precond_num) / factorize(A) / solve(b, x) as SparseNewtonDescentSolver calls
 * them (solver/SparseNewtonDescentSolver.tpp:128-155) on `hessian`, the REDUCED SUM of every form's second derivative
 * (solver/FullNLProblem.cpp:115-135, solver/NLProblem.cpp:92-99, utils::full_to_reduced_matrix), and as
 * polysolve::dirichlet_solve does for linear problems (state/StateSolveLinear.cpp:27-29,51-54).  A is the reference's
 * StiffnessMatrix, Eigen::SparseMatrix<double, ColMajor, int> in compressed form (utils/Types.hpp:16-20): outer (n + 1
 * column offsets; the 64 variant takes POLYSOLVE_LARGE_INDEX's wider type), inner (row of every entry), values.
 *
 * These calls need a context of their own (pfx_create; a context that holds a mesh is refused): it becomes a pure solver
 * context with no mesh behind it.  Any square matrix with a positive diagonal is accepted -- no symmetry of the PATTERN is
 * assumed (the arrays are transposed to row order on the device) -- CG itself needs a symmetric positive definite matrix.
 *
 *   pfx_solver_analyze(64)  once per pattern.  block_size 3: entries are grouped into 3 x 3 node blocks (one column index per
 *                           nine values in the solver's block-SELL format; entries missing from a block are explicit
 *                           zeros), n must be a multiple of 3; block_size 2: the same with 2 x 2 node blocks (2-D
 *                           elasticity), n a multiple of 2; block_size 1: scalar format; block_size 0: 3 if n is a
 *                           multiple of 3 and the blocks hold at most 25 % more entries than A, else 1.  A reduced matrix
 *                           whose constrained dofs are whole nodes, or a stiffness matrix from which a few exact zeros
 *                           were dropped, is blocked.
 *   pfx_solver_factorize    once per matrix: values in the order of `inner` (host; _dev: device pointer).  Returns
 *                           PFX_ERR_NUMERIC if a value is not finite or a diagonal entry is missing, not finite or not
 *                           positive -- the signal on which the reference regularises (SparseNewtonDescentSolver.tpp:135-150).
 *   then                    pfx_cg(_dev) = solve(b, x) with vectors of n entries, pfx_spmv(_dev), pfx_set_dirichlet /
 *                           pfx_dirichlet_solve (polysolve::dirichlet_solve on the unmodified stiffness matrix),
 *                           pfx_set_diag_shift, pfx_amg_*, pfx_cgs_* work on this matrix as on an assembled one.
 *   pfx_solver_set_node_positions  xyz[3 * node + c], one position per node (= n / block size): the AMG preconditioner of a
 *                           block-3 matrix builds its coarse spaces from the rigid-body modes, which need them; for a
 *                           scalar matrix they only steer the aggregation and may be left out.  NULL drops them.
 *   pfx_solver_get_layout   what analyze chose (block size, number of blocks, stored entries including explicit zeros).
 *   pfx_matrix_generation   a counter that changes whenever the matrix values of the context change (any assembly,
 *                           pfx_solver_factorize, pfx_hessian_add_mass_dev ...): lets a host wrapper check that a matrix
 *                           object it handed out still describes what the device holds. */
pfx_status pfx_solver_analyze(pfx_ctx *ctx, int64_t n, const int32_t *outer, const int32_t *inner, int block_size);
pfx_status pfx_solver_analyze64(pfx_ctx *ctx, int64_t n, const int64_t *outer, const int32_t *inner, int block_size);
pfx_status pfx_solver_factorize(pfx_ctx *ctx, const double *values);
pfx_status pfx_solver_factorize_dev(pfx_ctx *ctx, const double *d_values);
pfx_status pfx_solver_set_node_positions(pfx_ctx *ctx, const double *xyz, int64_t n_nodes);
pfx_status pfx_solver_get_layout(pfx_ctx *ctx, int *block_size, int64_t *n_blocks, int64_t *stored_entries);
pfx_status pfx_matrix_generation(pfx_ctx *ctx, int64_t *generation);

/* ---- algebraic multigrid preconditioner -------------------------------------------------------------------------------
 *
 * The role of polysolve's AMG solvers ("AMGCL", "Hypre": CMakeLists.txt:101-102; options input-spec.json:1397-1558 --
 * CG with an AMG preconditioner, Chebyshev relaxation, `ncycle`, `max_levels`, tol 1e-10), which is what a PolyFEM user
 * selects when Jacobi-CG runs into max_iter.  Neither library is in the reference tree and this is not a restatement of
 * either: it is this library's own plain-aggregation AMG (aggregates by repeated pairwise matching; 3-D elasticity: six
 * rigid-body modes per aggregate; scalar problems: one constant per aggregate; Galerkin coarse matrices in the same
 * node-block format, Chebyshev relaxation), selected
 * with precond = PFX_PRECOND_AMG in pfx_cg(_dev), pfx_dirichlet_solve and pfx_newton_params.cg_precond.  The iteration
 * is still Eigen's ConjugateGradient (same stopping rule); only z = M^-1 r changes.  pfx_cg(_dev) runs it on an
 * unpartitioned mesh; on a partition every rank applies the hierarchy of its own block (pfx_amg_apply_dev below).
 * The hierarchy is rebuilt when the pattern or the Dirichlet set changes; coarse matrices at every new matrix. */
typedef struct
{
	int aggregate_passes; /* 4: rounds of pairwise matching per level, aggregates of up to 2^passes vertices */
	int relax_degree;     /* 1: Chebyshev degree of the relaxation before / after the coarse correction (AMGCL/precond/relax/degree) */
	int ncycle;           /* 2: W cycle (AMGCL/precond/ncycle, the reference's default); 1: V cycle */
	int max_levels;       /* 12 (AMGCL/precond/max_levels) */
	int coarsest_dofs;    /* 600: levels at or below this size are solved with a dense inverse */
	int power_iters;      /* 15: iterations of the estimate of the spectral radius of D^-1 A (AMGCL/precond/relax/power_iters); here Lanczos
	                         steps, whose largest Ritz value + 5 % ends the Chebyshev interval */
	double relax_lower;   /* 0.1: the relaxation targets the eigenvalues in [relax_lower * lmax, lmax] */
	double coarse_scale;  /* 1.6: factor on every coarse-grid correction, in [1, 2).  Piecewise rigid-body interpolation
	                         underestimates the smooth error it corrects; over-correcting makes up for most of it
	                         (1 M quadratic tetrahedra: 44 iterations at 1.7, 67 at 1) and keeps the preconditioner
	                         symmetric positive definite below 2 */
	int mixed_precision;  /* 1: the products inside the preconditioner read single-precision copies of the level matrices (half
	                         the bytes per product; vectors, accumulation, Galerkin products and the outer PCG stay double, so the
	                         solution is the same and the iteration count changes by a few at most); 0: everything double */
} pfx_amg_params;

typedef struct
{
	int n_levels;
	int64_t n_vertices[16];   /* vertices per level (level 0: owned mesh nodes; coarser: aggregates of the level above) */
	int64_t n_aggregates[16]; /* aggregates each level is cut into (0 on the coarsest) */
	int64_t n_blocks[16];     /* 3x3 blocks of each level's matrix */
	double lambda_max[16];    /* estimated spectral radius of D^-1 A per relaxed level */
	double operator_complexity; /* sum of n_blocks over the levels / n_blocks[0] */
	double symbolic_s, numeric_s; /* host wall time of the last hierarchy build / coarse-matrix refresh */
} pfx_amg_info;

void pfx_amg_default_params(pfx_amg_params *params);
pfx_status pfx_amg_set_params(pfx_ctx *ctx, const pfx_amg_params *params);
/* Builds / refreshes the hierarchy for the matrix the context holds now (solves with PFX_PRECOND_AMG do this themselves). */
pfx_status pfx_amg_setup(pfx_ctx *ctx);
pfx_status pfx_amg_get_info(pfx_ctx *ctx, pfx_amg_info *info);
/* agg[v] = aggregate of vertex v of `level` (-1: left out: constrained node), n_vertices[level] entries. */
pfx_status pfx_amg_get_aggregates(pfx_ctx *ctx, int level, int32_t *agg);
/* One application of the preconditioner, d_z = M^-1 d_r (device vectors of n_dofs; entries of constrained dofs of d_r
 * must be zero and come back zero).  On a partitioned context (pfx_set_owned with fewer owned than local nodes) the
 * hierarchy is that of the block of the matrix whose rows and columns the rank owns: the entries of halo nodes of d_r
 * are not read and those of d_z come back zero, no communication takes place, and the ranks together apply a
 * block-Jacobi preconditioner (polyfem_amd.distributed.DistributedCG drives the iteration; pfx_cg(_dev) with
 * PFX_PRECOND_AMG itself refuses a partitioned context).  Uses the workspace of pfx_cgs_set_workspace when one is set:
 * its third and fourth quarter are overwritten, and a refresh of the hierarchy also overwrites the second. */
pfx_status pfx_amg_apply_dev(pfx_ctx *ctx, const double *d_r, double *d_z);

/* ---- mass matrix and body-force right-hand side -------------------------------------------------------------------
 *
 * pfx_assemble_mass = Assembler<Mass>::assemble(..., is_mass = true) (assembler/Assembler.cpp:100-317, mass quadrature
 * :135-139 of order max(-1, 2p+1), basis/FEBasis3d.cpp:1468,1507; local kernel assembler/Mass.cpp:5-20): rho phi_i phi_j
 * on the diagonal of every block, into the resident pattern.  Afterwards the matrix of the context IS the mass matrix
 * (pfx_get_linear_csr returns the reference's value-dependent pattern, pfx_spmv / pfx_cg work with it) until the next
 * assembly.  density_e: one density per element (Density with one entry per body id, utils/ElasticityUtils.cpp:561-570)
 * or NULL for the uniform `density`.
 * pfx_assemble_rhs = RhsAssembler::assemble (assembler/RhsAssembler.cpp:55-98) for a constant force density
 * `force` (one component per dof of a node): rhs[node*d + c] = sum_e sum_q force_c rho det w phi_i(q), stiffness
 * quadrature rule. */
pfx_status pfx_assemble_mass(pfx_ctx *ctx, double density, const double *density_e);
pfx_status pfx_assemble_rhs(pfx_ctx *ctx, const double *force, double density, const double *density_e, double *rhs);
pfx_status pfx_assemble_rhs_dev(pfx_ctx *ctx, const double *force, double density, const double *density_e, double *d_rhs);
/* The same with a force density that varies in space: force_pts holds what problem.rhs() returns in the reference
 * (RhsAssembler.cpp:66-75), one vector per (element, quadrature point), n_el x n_pts x d doubles.  pfx_quadrature_points
 * gives the physical positions of those points (ElementAssemblyValues::val, ElementAssemblyValues.cpp:328-345;
 * n_el x n_pts x dim doubles, pts may be NULL to ask for n_pts only) for the caller to evaluate its function at. */
pfx_status pfx_quadrature_points(pfx_ctx *ctx, int *n_pts, double *pts);
pfx_status pfx_assemble_rhs_pointwise(pfx_ctx *ctx, const double *force_pts, double density, const double *density_e, double *rhs);
pfx_status pfx_assemble_rhs_pointwise_dev(pfx_ctx *ctx, const double *d_force_pts, double density, const double *density_e, double *d_rhs);

/* Density evaluated per quadrature point, as Density::operator() is in the reference (utils/ElasticityUtils.cpp:561-570,
 * called per point at assembler/Mass.cpp:10 and assembler/RhsAssembler.cpp:81): the host evaluates its density function at
 * the positions the library reports and passes one value per (element, point), point fastest.
 *   pfx_mass_quadrature_points        positions of the points of the MASS quadrature rule (n_el x n_pts x dim; pts may be NULL)
 *   pfx_assemble_mass_pointwise       pfx_assemble_mass with n_el x n_pts densities at those points
 *   pfx_inertia_set_mass_pointwise    pfx_inertia_set_mass likewise
 *   pfx_assemble_rhs_density_pointwise  the body-force vector with n_el x n_pts densities at the points of the STIFFNESS rule
 *                                     (pfx_quadrature_points) and either a constant force (force, one double per dof of a node) or a force per
 *                                     point (force_pts, n_el x n_pts x d); exactly one of the two is non-NULL */
pfx_status pfx_mass_quadrature_points(pfx_ctx *ctx, int *n_pts, double *pts);
pfx_status pfx_assemble_mass_pointwise(pfx_ctx *ctx, const double *density_pts, int64_t n_el, int n_pts);
pfx_status pfx_inertia_set_mass_pointwise(pfx_ctx *ctx, const double *density_pts, int64_t n_el, int n_pts);
pfx_status pfx_assemble_rhs_density_pointwise(pfx_ctx *ctx, const double *force, const double *force_pts, const double *density_pts, int64_t n_el,
											  int n_pts, double *rhs);

/* ---- Newton solve with the iterate resident on the device ------------------------------------------------------
 *
 * cppoptlib::NonlinearSolver::minimize (solver/NonlinearSolver.tpp:33-275) driving
 * SparseNewtonDescentSolver::compute_update_direction / check_direction (solver/SparseNewtonDescentSolver.tpp:63-193)
 * and BacktrackingLineSearch::line_search (solver/line_search/BacktrackingLineSearch.hpp:29-255,
 * LineSearch.tpp:90-145) on the objective of NLProblem (solver/NLProblem.cpp:79-99) with an ElasticForm and a
 * BodyForm:  E(x) - rhs.x, the dofs of pfx_set_dirichlet held at their incoming values in x.  x, the gradient and the
 * direction never leave the device; per iteration only a handful of scalars are read back.  Parameter names and
 * defaults follow input-spec.json:1587-1647 (solver/nonlinear/...).  Descent strategy 1 ("projected Newton") assembles
 * with project_to_psd and the regularisation weights, as the reference does. */
typedef struct
{
	double f_delta;             /* 1e-10 */
	double grad_norm;           /* 1e-8  */
	double first_grad_norm_tol; /* 1e-10 */
	double use_grad_norm_tol;   /* line_search/use_grad_norm_tol, 1e-4 */
	int max_iterations;         /* 1000  */
	int use_grad_norm;          /* 1     */
	int relative_gradient;      /* 0     */
	int line_search;            /* 1 = "backtracking", 0 = none (full steps) */
	int cg_precond;             /* PFX_PRECOND_JACOBI */
	int cg_max_iter;            /* 1000  */
	double cg_tol;              /* 1e-12 */
} pfx_newton_params;

enum
{
	PFX_NEWTON_CONTINUE = 0,
	PFX_NEWTON_ITERATION_LIMIT = 1, /* the reference throws "Reached iteration limit" */
	PFX_NEWTON_F_DELTA = 2,
	PFX_NEWTON_GRAD_NORM = 3,
	PFX_NEWTON_NOT_STARTED = 4, /* "Not even starting, grad is small enough" */
	PFX_NEWTON_FAILED = 5       /* nan energy / gradient or a failed line search on gradient descent (the reference throws) */
};

typedef struct
{
	int status;
	int iterations;
	int line_search_iterations; /* objective evaluations spent inside the line searches */
	int descent_strategy;       /* 0 Newton, 1 regularised Newton, 2 gradient descent: strategy of the last accepted step */
	int64_t cg_iterations;
	double energy, grad_norm;   /* objective and gradient norm at the start of the last iteration */
} pfx_newton_info;

void pfx_newton_default_params(pfx_newton_params *params);
/* rhs (may be NULL) and x are FULL size host vectors; x is the starting point on input, the minimiser on output.
 * Returns PFX_ERR_NUMERIC (and info->status = PFX_NEWTON_FAILED / ITERATION_LIMIT) where the reference throws. */
pfx_status pfx_newton_solve(pfx_ctx *ctx, const double *rhs, double *x_inout, const pfx_newton_params *params,
							pfx_newton_info *info);
pfx_status pfx_newton_solve_dev(pfx_ctx *ctx, const double *d_rhs, double *d_x_inout, const pfx_newton_params *params,
								pfx_newton_info *info);

/* ---- transient steps: inertia term and the implicit Euler integrator ----------------------------------------------
 *
 * The reference's transient nonlinear solve (state/StateSolveNonlinear.cpp:355-420) minimises, per time step,
 *   acceleration_scaling * (ElasticForm(x) + BodyForm(x)) + InertiaForm(x),
 *   InertiaForm(x) = 1/2 (x - x_tilde)^T M (x - x_tilde)          (solver/forms/InertiaForm.cpp:14-31)
 * with the weights set at StateSolveNonlinear.cpp:78-84 and x_tilde / acceleration_scaling from the time integrator
 * (ImplicitEuler: x_tilde = x_prev + dt v_prev, acceleration_scaling = dt^2, time_integrator/ImplicitEuler.cpp:13-33).
 *
 * pfx_inertia_set_mass  assembles the mass matrix (as pfx_assemble_mass does, leaving it as the context's matrix) and
 *                       keeps it for the inertia term: one double per node block, since Mass::assemble puts the same
 *                       rho phi_i phi_j on the three diagonal entries of a block.  Call again after pfx_build_pattern.
 * pfx_inertia_enable    from now on pfx_newton_solve(_dev) minimises acceleration_scaling * (E(x) - rhs.x) +
 *                       InertiaForm(x) with the device vector d_x_tilde (which must stay valid);
 *                       acceleration_scaling <= 0 or d_x_tilde == NULL switches back to the static objective.
 * pfx_implicit_euler_step(_dev)  one whole step: x_tilde from x (= x_prev on input) and v, Newton solve starting at
 *                       x_prev as the reference does, then ImplicitEuler::update_quantities: v = (x - x_prev)/dt,
 *                       a = (v - v_prev)/dt (a may be NULL).  rhs is the body-force vector of pfx_assemble_rhs
 *                       (may be NULL); the dofs of pfx_set_dirichlet keep their incoming values.  Errors as
 *                       pfx_newton_solve; on error v and a are left untouched. */
pfx_status pfx_inertia_set_mass(pfx_ctx *ctx, double density, const double *density_e);
pfx_status pfx_inertia_enable(pfx_ctx *ctx, double acceleration_scaling, const double *d_x_tilde);
/* Augmented-Lagrangian treatment of the Dirichlet conditions (solver/forms/ALForm.cpp:7-86, solver/ALSolver.cpp:22-108):
 * while enabled, pfx_newton_solve(_dev) adds  weight/2 (x - target)^T diag(masked_lumped_mass) (x - target)  to its
 * objective.  masked_lumped_mass is the lumped mass matrix (utils::lump_matrix: row sums; the identity for a static
 * problem) with the entries of the free dofs set to 0; target holds the Dirichlet values (RhsAssembler::set_bc).  The
 * caller drives ALSolver's loop: pfx_set_dirichlet with no dofs, pfx_al_enable(weight), solve, double the weight while
 * the projected state is invalid, then pfx_al_enable(0) + pfx_set_dirichlet(dofs) and the final solve.
 * weight <= 0 or a null pointer disables the term.  The _dev form keeps the caller's device pointers (n_dofs doubles
 * each, valid until disabled); the _host form copies.  pfx_lumped_mass returns the row sums of the mass matrix kept by
 * pfx_inertia_set_mass (host vector of n_dofs). */
pfx_status pfx_al_enable(pfx_ctx *ctx, double weight, const double *d_masked_lumped_mass, const double *d_target);
pfx_status pfx_al_enable_host(pfx_ctx *ctx, double weight, const double *masked_lumped_mass, const double *target);
pfx_status pfx_lumped_mass(pfx_ctx *ctx, double *lumped);

/* the same with x_tilde on the host: it is copied into a buffer of the context */
pfx_status pfx_inertia_enable_host(pfx_ctx *ctx, double acceleration_scaling, const double *x_tilde);
pfx_status pfx_implicit_euler_step(pfx_ctx *ctx, double dt, const double *rhs, double *x_inout, double *v_inout, double *a_out,
								   const pfx_newton_params *params, pfx_newton_info *info);
pfx_status pfx_implicit_euler_step_dev(pfx_ctx *ctx, double dt, const double *d_rhs, double *d_x_inout, double *d_v_inout,
									   double *d_a_out, const pfx_newton_params *params, pfx_newton_info *info);

/* ---- multi-GPU building blocks (no counterpart in the reference, which is single-process) -------------------------
 *
 * One process per GPU holds one part of a node partition: its mesh is every element that touches an owned node,
 * local node numbering is [owned nodes | halo nodes].  Assembly then needs no communication (owned rows are
 * complete); the solver needs (a) the halo entries of the search direction before every SpMV and (b) sums over the
 * ranks of three scalars per iteration.  Both are done by the caller with RCCL (torch.distributed) between the step
 * functions below, which only enqueue kernels on the context's stream.
 *
 *   pfx_cgs_prepare   mask x on constrained dofs, build the Jacobi diagonal              -> exchange halo of x
 *   pfx_cgs_residual  r = b - A x ; scal[0] = b.b, scal[1] = r.r (local sums)            -> all-reduce scal[0..1]
 *   pfx_cgs_start     threshold, early exit, p = M^-1 r ; scal[2] = r.p                  -> all-reduce scal[2]
 *   pfx_cgs_start2    stores r.p
 *   per iteration it: (exchange halo of p)  pfx_cgs_spmv: tmp = A p, scal[0] = p.tmp     -> all-reduce scal[0]
 *                       or, to hide the exchange: start it, pfx_cgs_spmv_interior (rows whose columns are all owned),
 *                       wait for it, pfx_cgs_spmv_boundary (remaining rows, then scal[0])
 *                     pfx_cgs_update: x += a p, r -= a tmp, scal[1] = r.r, scal[2] = r.z -> all-reduce scal[1..2]
 *                     pfx_cgs_direction: convergence test, p = M^-1 r + beta p
 *   pfx_cgs_status    synchronises and reports (done, iterations, ||r||^2, ||b||^2)
 *
 * tmp = A p may stay in the solver copy's own row order between pfx_cgs_spmv* and pfx_cgs_update (which reads it there);
 * a caller that reads tmp in d_work itself calls pfx_cgs_fetch_product first.
 *
 * d_scal is a device array of >= 4 doubles owned by the caller.  With d_scal == NULL the steps reduce locally and
 * form exactly the single-GPU pfx_cg_dev.  d_work (4 * n_local_dofs doubles: r, p, tmp, invdiag in this order) is the
 * caller-owned workspace, so that the caller can address the halo part of p. */
pfx_status pfx_set_owned(pfx_ctx *ctx, int64_t n_owned_nodes, const uint8_t *elem_owned /* n_el flags or NULL */);
/* n_doubles = size of the caller's buffer, at least 4 * n_local_dofs (rejected otherwise); d_work == NULL returns to the
 * context's own workspace.  The pointer is dropped by pfx_set_mesh / pfx_build_pattern (its size belonged to the old
 * pattern) and must be cleared by the caller before the buffer is released. */
pfx_status pfx_cgs_set_workspace(pfx_ctx *ctx, double *d_work, int64_t n_doubles);
pfx_status pfx_cgs_prepare(pfx_ctx *ctx, double *d_x, int precond);
pfx_status pfx_cgs_residual(pfx_ctx *ctx, const double *d_b, const double *d_x, double *d_scal);
pfx_status pfx_cgs_start(pfx_ctx *ctx, double *d_x, double tol, double *d_scal);
pfx_status pfx_cgs_start2(pfx_ctx *ctx, const double *d_scal);
pfx_status pfx_cgs_spmv(pfx_ctx *ctx, double *d_scal);
pfx_status pfx_cgs_spmv_interior(pfx_ctx *ctx);
pfx_status pfx_cgs_spmv_boundary(pfx_ctx *ctx, double *d_scal);
pfx_status pfx_cgs_fetch_product(pfx_ctx *ctx);
pfx_status pfx_cgs_update(pfx_ctx *ctx, int it, double *d_x, double *d_scal);
pfx_status pfx_cgs_direction(pfx_ctx *ctx, int it, const double *d_scal);
pfx_status pfx_cgs_status(pfx_ctx *ctx, int *done, int *iters, double *rr, double *bb);

/* Pieces a Newton iteration over the partition needs beside the step functions (the serial counterpart is
 * pfx_newton_solve_dev; the whole iteration over a partition is pfx_dist_newton_dev, these pieces are for callers that
 * drive it themselves, as polyfem_amd.distributed.DistributedNewton does for transient steps):
 * pfx_set_diag_shift: from now on pfx_spmv*, pfx_cg* and the step functions work with A + shift * I on the free dofs --
 *   the regularised Hessian `hessian += reg_weight * sparse_identity` of SparseNewtonDescentSolver.tpp:116-121 without
 *   touching the assembled values.  shift = 0 switches it off.
 * pfx_take_numeric_flag: *flag = 1 if an element had a non-positive deformation-gradient determinant in any
 *   pfx_*_dev evaluation since the last call (the host variants report that as PFX_ERR_NUMERIC themselves); clears
 *   the flag.  Synchronises the stream. */
pfx_status pfx_set_diag_shift(pfx_ctx *ctx, double shift);
/* The inertia term for a driver that runs the Newton loop itself (on a partition: the rows this rank owns):
 * pfx_mass_apply_dev       d_y[owned rows] = M d_v with the mass matrix kept by pfx_inertia_set_mass -- InertiaForm's
 *                          first derivative M (x - x_tilde) when d_v holds x - x_tilde (solver/forms/InertiaForm.cpp:23-26);
 * pfx_hessian_add_mass_dev matrix := weight * matrix + M, the weighted sum of ElasticForm's and InertiaForm's second
 *                          derivatives (:28-31, weights at state/StateSolveNonlinear.cpp:78-84); call it after
 *                          pfx_hessian_dev / pfx_hessian_psd_dev. */
pfx_status pfx_mass_apply_dev(pfx_ctx *ctx, const double *d_v, double *d_y);
pfx_status pfx_hessian_add_mass_dev(pfx_ctx *ctx, double weight);
pfx_status pfx_take_numeric_flag(pfx_ctx *ctx, int *flag);

/* ---- the partitioned solve inside the library (no counterpart in the reference, which is single-process) -----------------
 *
 * The step functions above leave the communication to the caller.  The calls below run the whole distributed iteration --
 * halo exchange of the search direction hidden behind the interior rows, all-reduce of the scalars, convergence flag on the
 * device looked at every `check_every` iterations -- inside the library, over a communicator object with two
 * implementations:
 *
 *   pfx_comm_create_rccl   RCCL over xGMI: grouped ncclSend / ncclRecv between neighbour ranks on a communication stream of
 *                          its own (ordered with the context's stream by events, so the interior rows of a product run while
 *                          the halo travels), ncclAllReduce on the context's stream.  One process per GPU; rank 0 makes the id
 *                          with pfx_comm_rccl_unique_id and the caller hands it to the other ranks (MPI_Bcast, a file, ...).
 *   pfx_comm_create_host   the caller's own transport for HOST buffers (MPI_Sendrecv / MPI_Allreduce, gloo, pipes): the
 *                          library stages device data through pinned host memory and synchronises the stream around every
 *                          callback.  For tests and for machines without a device-to-device path.
 *
 * A communicator belongs to one host thread; its calls are collective (every rank makes the same sequence of calls). */
typedef struct pfx_comm pfx_comm;
#define PFX_RCCL_ID_BYTES 128

typedef struct
{
	void *user;
	/* For k = 0 .. n-1: send send_bytes[k] bytes at send[k] to rank peer[k] and receive recv_bytes[k] bytes into recv[k] from
	 * it (either count may be 0).  Blocking; all pointers are host memory; returns 0 on success. */
	int (*exchange)(void *user, int n, const int *peer, const void *const *send, const int64_t *send_bytes, void *const *recv, const int64_t *recv_bytes);
	/* buf[0 .. count) := sum over all ranks, in place, the same bits on every rank.  Blocking; returns 0 on success. */
	int (*all_reduce_sum)(void *user, double *buf, int count);
} pfx_host_transport;

pfx_status pfx_comm_rccl_unique_id(void *id /* PFX_RCCL_ID_BYTES bytes */);
pfx_status pfx_comm_create_rccl(pfx_comm **comm, const void *id, int rank, int world, int device);
pfx_status pfx_comm_create_host(pfx_comm **comm, const pfx_host_transport *transport, int rank, int world);
void pfx_comm_destroy(pfx_comm *comm);
const char *pfx_comm_last_error(const pfx_comm *comm);

/* A partitioned context (pfx_set_mesh of the rank's part, pfx_set_owned) bound to a communicator and its halo lists.
 * Neighbour k = rank peer[k]: this rank SENDS it the entries of its owned nodes send_nodes[send_ptr[k] .. send_ptr[k+1])
 * (local ids < n_owned, in the order the neighbour stores them) and RECEIVES from it the halo nodes n_owned + recv_ptr[k] ..
 * n_owned + recv_ptr[k+1] - 1 (halo nodes are numbered neighbour by neighbour).  The context and the communicator must
 * outlive the object; pfx_set_mesh / pfx_build_pattern on the context invalidate it. */
typedef struct pfx_dist pfx_dist;
pfx_status pfx_dist_create(pfx_dist **dist, pfx_ctx *ctx, pfx_comm *comm, int n_neigh, const int32_t *peer, const int64_t *send_ptr,
						   const int32_t *send_nodes, const int64_t *recv_ptr);
void pfx_dist_destroy(pfx_dist *dist);
const char *pfx_dist_last_error(const pfx_dist *dist);
/* d_vec[(n_owned + h) * bs + c] := the owner's value, for every halo node h (device vector of n_local * bs doubles) */
pfx_status pfx_dist_halo_exchange_dev(pfx_dist *dist, double *d_vec);
/* d_buf[0 .. count) := sum over the ranks (device pointer, count <= 64) */
pfx_status pfx_dist_all_reduce_dev(pfx_dist *dist, double *d_buf, int count);
/* NLAssembler::assemble -> energy over the partition: every rank sums the elements pfx_set_owned flagged, one all-reduce */
pfx_status pfx_dist_energy_dev(pfx_dist *dist, const double *d_u, double *d_energy);
/* pfx_cg_dev over the partition: d_b, d_x are this rank's local vectors (n_local * bs doubles; the halo entries of d_b are
 * ignored, those of d_x are current on return).  Same iteration, stopping rule and iteration count convention as pfx_cg_dev;
 * precond PFX_PRECOND_IDENTITY / _JACOBI / _AMG (see pfx_dist_amg_* below).  check_every <= 0: 50. */
pfx_status pfx_dist_cg_dev(pfx_dist *dist, const double *d_b, double *d_x, int precond, int max_iter, double tol, int check_every,
						   int *iters, double *rel_res);
/* PFX_PRECOND_AMG over a partition: ONE hierarchy that spans the ranks.  Every rank aggregates the vertices it owns, the
 * Galerkin products include the couplings across the partition boundaries (aggregate ids and offsets of the halo vertices
 * come from their owners once per level at set-up), every coarse level is again a row-distributed matrix whose products are
 * preceded by a halo exchange of that level, and the coarsest matrix is gathered and inverted on every rank.  It is the
 * preconditioner a single context would build from the union of the ranks' aggregates.  pfx_dist_amg_apply_dev applies it once
 * (d_r, d_z local vectors; collective); pfx_amg_get_info / pfx_amg_get_aggregates on the context then describe this rank's
 * share (vertices and aggregates it owns; the ids of level >= 1 vertices are local to the rank).  Parameters:
 * pfx_amg_set_params on the context.  Limit: 32 ranks. */
pfx_status pfx_dist_amg_apply_dev(pfx_dist *dist, const double *d_r, double *d_z);
/* pfx_newton_solve_dev over the partition (SparseNewtonDescentSolver::minimize with NLProblem = ElasticForm + BodyForm,
 * src/polyfem/solver/SparseNewtonDescentSolver.tpp:63-193, src/polyfem/solver/NonlinearSolver.tpp:33-275): the same control
 * flow on every rank, every decision taken on all-reduced scalars (energy of the owned elements, sums and maxima over the
 * owned dofs, the inverted-element flag), the linear solve is pfx_dist_cg_dev with params->cg_precond.  d_rhs (may be NULL)
 * and d_x are this rank's local vectors; the halo entries of d_rhs are ignored, those of d_x are made current on entry and
 * are current on return.  Statuses, errors and counters as pfx_newton_solve_dev, identical on all ranks.  The inertia term
 * (pfx_inertia_set_mass + pfx_inertia_enable on every rank, d_x_tilde a local vector with a current halo) is honoured; the
 * augmented-Lagrangian term (pfx_al_enable) is not available over a partition.  Limit: 38 ranks.
 * pfx_dist_implicit_euler_step_dev is pfx_implicit_euler_step_dev over the partition (ImplicitEuler.cpp:5-33 around that
 * solve): d_x, d_v local vectors with current halos on entry and on return, d_a (may be NULL) likewise on return. */
pfx_status pfx_dist_newton_dev(pfx_dist *dist, const double *d_rhs, double *d_x_inout, const pfx_newton_params *params,
							   pfx_newton_info *info);
pfx_status pfx_dist_implicit_euler_step_dev(pfx_dist *dist, double dt, const double *d_rhs, double *d_x_inout, double *d_v_inout,
											double *d_a_out, const pfx_newton_params *params, pfx_newton_info *info);
/* bytes one halo exchange of a vector moves through this rank (sent + received) */
pfx_status pfx_dist_halo_bytes(pfx_dist *dist, int64_t *bytes);

/* ---- placement of the solver's value arrays (no counterpart in the reference) --------------------------------------
 *
 * The speed of the matrix-vector product depends on which physical pages its value array lies in (measured: +-5 % for a
 * 30 GB array at unchanged virtual addresses).  The CG loops (pfx_cg*, pfx_dist_cg_dev with Jacobi / no preconditioner)
 * therefore try a few allocations of the array once a solve has run 192 iterations with at least 256 to go, and keep the
 * fastest.  pfx_tune_placement does the same on request, at once, and also for the single-precision copies of an existing
 * AMG hierarchy (whose solves are too short to trigger a search): call it after the first assembly / pfx_amg_setup of a
 * simulation that will solve many times on this pattern.  Cost: one copy of the array + four products per try (default
 * six tries, PFX_PLACEMENT_TRIES in the environment; 0 = never), room for a second copy of the array while it runs, and a
 * driver allocator that answers slowly for a few seconds afterwards (about a second per 12 GB the search freed).  Arrays
 * below 1 GiB are left alone.  Results do not change. */
pfx_status pfx_tune_placement(pfx_ctx *ctx);

/* ---- introspection for benchmarks ----------------------------------------------------------------------------- */

typedef struct
{
	int64_t n_el, n_nodes, n_dofs, n_blocks, nnz;
	int n_loc, n_pts, dim;
	int64_t bytes_device; /* device memory held by the context */
} pfx_info;
pfx_status pfx_get_info(pfx_ctx *ctx, pfx_info *info);

/* Average duration (ms) of the launches of one named kernel family since the last reset, measured with HIP events
 * on the context's stream when timing is enabled.  Families: "local", "gather" (matrix assembly), "local_grad",
 * "gather_vec", "local_energy", "spmv", "spmv_boundary", "cg_update", "cg_dir", "sell_convert", "mass".  on = 1 times every launch; on = k > 1
 * times every k-th launch of each family (two event records per timed launch cost about 3 us of stream time each,
 * which is visible in a loop of sub-millisecond kernels). */
pfx_status pfx_timing_enable(pfx_ctx *ctx, int on);
pfx_status pfx_timing_get(pfx_ctx *ctx, const char *family, double *avg_ms, int64_t *launches);
pfx_status pfx_timing_reset(pfx_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
