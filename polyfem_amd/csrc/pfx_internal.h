// pfx_internal.h -- context layout and launch helpers shared by the HIP translation units of libpolyfem_amd.so
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstring>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "../../include/polyfem_amd.h"

#define PFX_MAX_LOC 27
#define PFX_MAX_GLOC 8
#define PFX_MAX_PTS 125 // 5^3: mass quadrature of Q2 hexes

// Reference-element tables of one element kind, resident in device memory.
struct pfx_tables
{
	int n_loc, n_gloc, n_pts;
	double w[PFX_MAX_PTS];                        // quadrature weights (already /6 for tets)
	double grad[PFX_MAX_PTS][PFX_MAX_LOC][3];     // d(basis j)/d(ref coord) at point k
	double ggrad[PFX_MAX_PTS][PFX_MAX_GLOC][3];   // same for the geometry bases
	double val[PFX_MAX_PTS][PFX_MAX_LOC];         // basis j at point k
	double gval[PFX_MAX_PTS][PFX_MAX_GLOC];       // geometry basis j at point k
};

// Work split over the XCDs.  Workgroups are dealt to the eight XCDs round robin (workgroup b runs on XCD b % 8) and every
// XCD has its own L2: kernels whose neighbouring work units share data give each XCD one CONTIGUOUS range of units,
// units first[x] .. first[x+1]-1 to XCD x, cut so that the ranges carry equal work (rows differ in length: with vertex
// nodes numbered before edge nodes, equal COUNTS would hand one XCD all the long rows).
#define PFX_XCDS 8
struct pfx_xcd_split
{
	int32_t first[PFX_XCDS + 1];
	int32_t widest; // largest range: the launch has PFX_XCDS * widest workgroups
};
// prefix(k) = work of the units before unit k (non-decreasing, prefix(0) = 0), k = 0 .. n
template <class Prefix>
inline pfx_xcd_split pfx_make_xcd_split(int64_t n, Prefix prefix)
{
	pfx_xcd_split sp;
	const double total = (double)prefix(n);
	sp.first[0] = 0;
	for (int x = 1; x < PFX_XCDS; ++x)
	{
		const double target = total * x / PFX_XCDS;
		int64_t lo = sp.first[x - 1], hi = n; // smallest k with prefix(k) >= target
		while (lo < hi)
		{
			const int64_t mid = (lo + hi) >> 1;
			if ((double)prefix(mid) < target)
				lo = mid + 1;
			else
				hi = mid;
		}
		sp.first[x] = (int32_t)lo;
	}
	sp.first[PFX_XCDS] = (int32_t)n;
	sp.widest = 1;
	for (int x = 0; x < PFX_XCDS; ++x)
		if (sp.first[x + 1] - sp.first[x] > sp.widest)
			sp.widest = sp.first[x + 1] - sp.first[x];
	return sp;
}

// window descriptor (uint4): x = index of the block in the value array, y = index of its mirror image below the diagonal
// (0xFFFFFFFF: a diagonal block), z = c0 | c1 << 8 | flags: its contributions are the window's [c0, c1)
#define PFX_DESC_CONT_IN 0x10000u  // the block started in an earlier window: continue from the carried sums
#define PFX_DESC_CONT_OUT 0x20000u // the block goes on in the next window: carry the sums instead of storing them
#ifndef PFX_ROWS_PER_GROUP
#define PFX_ROWS_PER_GROUP 16
#endif

#define PFX_NOALLOC_MAX_EL (1 << 20) // element limit of pfx_energy_dev_noalloc (size of its static device scratch)
#define PFX_MAX_DEVICES 64

enum
{
	PFX_ESTATE_NONE = 0,
	PFX_ESTATE_LINEAR = 1, // J^{-T}, det per point (pfx_local_linear.hip)
	PFX_ESTATE_NL = 2      // J^{-T}, F, coefficients per point (pfx_local_nl.hip)
};

struct pfx_amg; // hierarchy of the aggregation-AMG preconditioner (pfx_amg.hip)

struct pfx_timer_family
{
	double total_ms = 0;
	int64_t launches = 0;
};

struct pfx_ctx
{
	int device = 0;
	hipStream_t stream = nullptr;
	bool own_stream = false;
	std::string err;

	// mesh
	int kind = -1, gkind = -1, n_loc = 0, n_gloc = 0, n_pts = 0, quad_order = -1;
	int mesh_dim = 3; // 2 for the triangle kinds (the device copy of the geometry nodes still has three coordinates, the third zero)
	int64_t n_el = 0, n_nodes = 0, n_gnodes = 0;
	int64_t n_owned = 0; // rows [0, n_owned) belong to this rank (== n_nodes unless the mesh is one part of a partition)
	int32_t *d_conn = nullptr, *d_gconn = nullptr;
	double *d_xyz = nullptr;
	pfx_tables *d_tab = nullptr;
	pfx_tables h_tab;
	pfx_tables *d_tab_mass = nullptr; // same for the mass quadrature rule (order 2p+1)
	pfx_tables h_tab_mass;
	double *d_density_e = nullptr;    // per-element density of the last mass / rhs assembly
	int64_t density_count = 0;        // doubles allocated in d_density_e

	// material
	int model = -1, dim = 3, bs = 0; // bs = block size (1 or dim)
	double lambda = 0, mu = 0;
	uint8_t *d_model_e = nullptr; // PFX_MULTI_MODEL: 0 LinearElasticity, 1 NeoHookean, 2 SaintVenant per element (the element kernels' numbering)
	double *d_tensor = nullptr; // PFX_HOOKE_LINEAR, PFX_SAINT_VENANT with stvk_tensor: symmetric 6 x 6 elasticity tensor (36 doubles)
	bool stvk_tensor = false;   // PFX_SAINT_VENANT reads d_tensor instead of (lambda, mu)

	// pattern: node-block structure; the scalar CSR is implied (row gi*bs+m holds, for every neighbour slot s,
	// the bs columns node_col[s]*bs+n) and the values array IS that CSR's value array.
	int64_t n_blocks = 0, nnz = 0;
	int64_t *d_node_ptr = nullptr; // n_nodes+1
	int32_t *d_node_col = nullptr; // n_blocks, ascending per node
	// Only the blocks on and above the diagonal (column node >= row node, "upper" blocks, n_ublocks of them, numbered in
	// block order) are assembled from contributions; the lane that finishes an entry of an upper block also writes it,
	// transposed, into the block's mirror image below the diagonal.
	int64_t n_ublocks = 0, n_upair = 0;
	// contribution lists (n_upair entries: n_loc(n_loc+1)/2 per element), in block order, ascending element inside a
	// block.  The assembly kernel walks that list in windows of 64 per group of PFX_ROWS_PER_GROUP node rows; what a
	// window needs to know about the blocks it intersects is precomputed (pfx_pattern.hip):
	int64_t n_groups = 0, n_windows = 0;
	uint32_t *d_grp_c0 = nullptr;  // n_groups+1: first contribution of each group
	uint32_t *d_grp_win = nullptr; // n_groups+1: first window of each group
	uint2 *d_win = nullptr;        // n_windows: (first upper block bl, number of blocks nb) of the window
	uint4 *d_desc = nullptr;       // n_ublocks + n_windows: block bl+s of window W at bl + s + W, see PFX_DESC_*
	std::vector<uint32_t> h_grp_win; // host copy of d_grp_win
	// The assembly takes the row groups in the order d_grp_order (n_groups): along a Morton curve through the groups'
	// positions, so that the groups resident on an XCD at one time form a compact region and share their elements' states
	// in its L2 whatever the caller's node numbering is.  h_grp_work[k] = windows (+ 1 per group) of the first k groups
	// of that order: the XCD ranges are cut by it.
	uint32_t *d_grp_order = nullptr;
	std::vector<double> h_grp_work;
	pfx_xcd_split gv_split;          // XCD ranges of the vector gather's blocks (256 scalar rows each), cut by incidences
	// Inside each window the contributions are stored reordered by
	// element (lanes that read the same element state sit next to each other): d_blk_src[c] = e*n_loc^2 + i*n_loc + j of
	// the contribution computed by lane c - window start, d_blk_slot[c] = its position in the window in block order.
	uint32_t *d_blk_src = nullptr;
	uint8_t *d_blk_slot = nullptr;
	int64_t *d_ne_ptr = nullptr;   // n_nodes+1 -> incident (element, local index) list
	uint32_t *d_ne_src = nullptr;  // n_el*n_loc : e*n_loc + i, ascending
	double *d_values = nullptr;    // nnz, node-block storage: entry (m, n) of block b at b*bs*bs + m*bs + n
	uint8_t *d_mask = nullptr;     // nnz (same layout), 1 = entry exists in the reference's value-dependent pattern
	bool have_mask = false;

	// scratch for the two-pass (local blocks, then per-row sums) assembly
	double *d_lambda_e = nullptr, *d_mu_e = nullptr; // per-element or per-point Lame parameters (null: uniform ctx->lambda, ctx->mu)
	int mat_per_point = 0; // 1: the two arrays hold one value per (element, quadrature point)
	double *d_kloc = nullptr;   // projected element Hessians (project_to_psd only): n_el * n_loc(n_loc+1)/2 * 9
	int64_t kloc_count = 0;
	double *d_estate = nullptr; // per-element state the row assembly computes its blocks from (n_el * 22 * n_pts doubles)
	int64_t estate_count = 0;   // doubles allocated in d_estate
	int estate_kind = 0;        // which element kernel wrote it last (PFX_ESTATE_*)
	double *d_gloc = nullptr; // n_el * n_loc*bs
	double *d_eloc = nullptr; // n_el
	int *d_flag = nullptr;    // numeric failure flag
	uint8_t *d_elem_owned = nullptr; // optional: 1 = this rank sums the element's energy (partitioned meshes)

	// Dirichlet
	int64_t n_dirichlet = 0;
	uint8_t *d_fixed = nullptr; // n_dofs, 1 = constrained
	int32_t *d_dirichlet = nullptr;

	// solver work vectors
	double *d_work = nullptr;     // 4 * n_dofs: r, p, tmp, invdiag
	double *d_work_ext = nullptr; // caller-owned workspace of the same layout (multi-GPU driver)
	double diag_shift = 0.0;      // the solver works with A + diag_shift * I on the free dofs (Newton regularisation)
	double *d_partials = nullptr; // room for 3 x 2048 per-block partial sums of the vector kernels (3 x PFX_NPART used), then 8 scalars at 3 * PFX_NPART (first: p.Ap of the current iteration)
	double *d_part0 = nullptr;    // per-block partials of the SpMV (one per SpMV block)
	int64_t part0_count = 0;
	double *d_newton = nullptr;   // workspace of the Newton driver (pfx_newton.hip)
	int64_t newton_count = 0;
	// inertia term of a transient step: 1/2 (x - x_tilde)^T M (x - x_tilde) beside inertia_weight * (E - rhs.x)
	double *d_mass_blk = nullptr; // n_blocks: rho phi_i phi_j of every node block (the mass matrix is that times I_3)
	int64_t mass_count = 0;       // 0 = no mass matrix captured
	double inertia_weight = 0.0;  // acceleration_scaling of the time integrator; 0 = static problem
	const double *d_x_tilde = nullptr;
	double *d_tstep = nullptr;    // x_prev and x_tilde of pfx_implicit_euler_step
	// augmented-Lagrangian penalty al_weight/2 (x - target)^T diag(al_mass) (x - target) beside the other terms (0: off)
	double al_weight = 0.0;
	const double *d_al_mass = nullptr, *d_al_target = nullptr; // n_dofs each: masked lumped mass, Dirichlet target
	double *d_al_store = nullptr; // the two vectors when they came from the host (pfx_al_enable_host)
	int64_t al_count = 0;
	int64_t tstep_count = 0;

	// solver copy of the matrix (block-SELL-64, pfx_sell.hip)
	int32_t *d_sell_perm = nullptr, *d_sell_cols = nullptr;
	int32_t *d_sell_pos = nullptr;   // row -> position (slice * 64 + lane) in the slice order
	double *d_sell_ys = nullptr;     // product in slice order, before k_sell_unpermute
	bool cg_tmp_sliced = false;      // the CG's last A p sits in d_sell_ys (slice order), not in the work vector
	int32_t *d_sell_order = nullptr; // slice list, interior slices first (null: natural order, no halo)
	int64_t sell_interior = 0;       // number of slices that read owned entries of x only
	pfx_xcd_split sell_split[3];     // positions of the slice list per XCD for PFX_SPMV_ALL / _INTERIOR / _BOUNDARY, cut by stored slots
	int64_t *d_sell_base = nullptr;
	double *d_sell_vals = nullptr;
	float *d_sell_vals32 = nullptr;  // single-precision copy for the products inside the AMG preconditioner
	bool sell_placed = false;        // the placement search has run for the current d_sell_vals (pfx_sell_place)
	bool sell32_placed = false;      // ... for the current d_sell_vals32 (pfx_sell_place32)
	int64_t sell32_gen = -1;
	int64_t sell_slices = 0, sell_rows = 0, sell_owned = -1;
	int64_t values_gen = 0, sell_gen = -1; // assembly counter / counter value the solver copy was made from
	int *d_cgstate = nullptr;
	double *d_u = nullptr, *d_vec = nullptr; // staging for host-pointer entry points

	// AMG preconditioner: parameters, hierarchy, and a counter of everything its symbolic part depends on (pattern,
	// Dirichlet set, owned range)
	pfx_amg_params amg_params = {4, 1, 2, 12, 600, 15, 0.1, 1.6, 1};
	pfx_amg *amg = nullptr;
	struct pfx_dist *amg_dist = nullptr; // set while a solve over a partition runs: the hierarchy then spans the ranks (pfx_dist.hip)
	int64_t struct_gen = 0;
	int64_t pattern_gen = 0; // bumped whenever the pattern is released / rebuilt (objects bound to the pattern check it)

	// a pattern and values handed in by the caller (pfx_solver_analyze / pfx_solver_factorize, pfx_extmat.hip) instead of
	// built from a mesh: the context is then a pure solver context (kind < 0)
	bool ext = false;
	bool ext_factorized = false;      // values are in place and passed the checks of pfx_solver_factorize
	int64_t ext_nnz = 0;              // entries of the caller's matrix
	uint32_t *d_ext_map = nullptr;    // nnz: position in the caller's value array of every stored entry (0xFFFFFFFF: explicit zero)
	double *d_ext_stage = nullptr;    // ext_nnz: device copy of the caller's host values
	double *d_ext_pos = nullptr;      // 3 * n_nodes: node positions for the AMG aggregation (optional)

	int64_t bytes_device = 0;                   // device memory currently held through pfx_alloc
	std::map<const void *, int64_t> alloc_bytes; // size of every live pfx_alloc allocation

	bool timing = false;
	int timing_stride = 1;                             // every timing_stride-th launch of a family is timed
	std::map<std::string, int64_t> timing_seen;        // launches of each family since the last reset
	std::map<std::string, pfx_timer_family> timers;
	struct pending_pair
	{
		const char *family;
		hipEvent_t e0, e1;
	};
	std::vector<hipEvent_t> ev_free;
	std::vector<pending_pair> ev_pending;

	int64_t n_dofs() const { return n_nodes * bs; }
};

#define PFX_HIP(call)                                                                                        \
	do                                                                                                       \
	{                                                                                                        \
		hipError_t e_ = (call);                                                                              \
		if (e_ != hipSuccess)                                                                                \
		{                                                                                                    \
			ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                                   \
			return PFX_ERR_HIP;                                                                              \
		}                                                                                                    \
	} while (0)

#define PFX_FAIL(code, msg)  \
	do                       \
	{                        \
		ctx->err = (msg);    \
		return (code);       \
	} while (0)

// hipFree + null (pointers owned by the context); keeps pfx_info.bytes_device equal to what is currently held
template <typename T>
inline void free_dev(pfx_ctx *ctx, T *&p)
{
	if (p)
	{
		auto it = ctx->alloc_bytes.find((const void *)p);
		if (it != ctx->alloc_bytes.end())
		{
			ctx->bytes_device -= it->second;
			ctx->alloc_bytes.erase(it);
		}
		(void)hipFree(p);
	}
	p = nullptr;
}

template <typename T>
pfx_status pfx_alloc(pfx_ctx *ctx, T **p, int64_t count)
{
	free_dev(ctx, *p);
	if (count <= 0)
		count = 1;
	PFX_HIP(hipMalloc((void **)p, sizeof(T) * (size_t)count));
	ctx->bytes_device += (int64_t)sizeof(T) * count;
	ctx->alloc_bytes[(const void *)*p] = (int64_t)sizeof(T) * count;
	return PFX_OK;
}

// grid of a grid-stride kernel over n items: enough blocks to fill the device, capped at 16 per CU
inline int grid_for(int64_t n, int block = 256)
{
	int64_t g = (n + block - 1) / block;
	if (g > 256 * 16)
		g = 256 * 16;
	if (g < 1)
		g = 1;
	return (int)g;
}

// Temporary device buffer of the host-side sequencing code: released when it goes out of scope, i.e. also on the early
// returns of PFX_HIP / PFX_FAIL.  Kernel launches take the raw pointer (.p).
template <typename T>
struct pfx_tmp
{
	T *p = nullptr;
	pfx_tmp() = default;
	pfx_tmp(const pfx_tmp &) = delete;
	pfx_tmp &operator=(const pfx_tmp &) = delete;
	~pfx_tmp() { reset(); }
	hipError_t alloc(size_t count)
	{
		reset();
		return hipMalloc((void **)&p, sizeof(T) * (count ? count : 1));
	}
	void reset()
	{
		if (p)
			(void)hipFree(p);
		p = nullptr;
	}
};

// Times one kernel family with a pair of HIP events recorded on the context's stream; nothing synchronises
// until pfx_timing_get()/pfx_timing_reset() resolves the pending pairs.
struct pfx_scoped_timer
{
	pfx_ctx *ctx;
	const char *family;
	hipEvent_t e0 = nullptr, e1 = nullptr;
	pfx_scoped_timer(pfx_ctx *c, const char *f) : ctx(c), family(f)
	{
		if (!ctx->timing)
			return;
		if (ctx->timing_stride > 1 && (ctx->timing_seen[f]++ % ctx->timing_stride) != 0)
			return;
		if (ctx->ev_free.size() < 2)
		{
			hipEvent_t a, b;
			if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess)
				return;
			ctx->ev_free.push_back(a);
			ctx->ev_free.push_back(b);
		}
		e0 = ctx->ev_free.back();
		ctx->ev_free.pop_back();
		e1 = ctx->ev_free.back();
		ctx->ev_free.pop_back();
		(void)hipEventRecord(e0, ctx->stream);
	}
	~pfx_scoped_timer()
	{
		if (!e0)
			return;
		(void)hipEventRecord(e1, ctx->stream);
		ctx->ev_pending.push_back({family, e0, e1});
	}
};

// translation-unit entry points
void pfx_fill_tables(int kind, int quad_order, pfx_tables *t);
void pfx_fill_tables_mass(int kind, pfx_tables *t);
pfx_status pfx_pattern_build(pfx_ctx *ctx);
pfx_status pfx_sell_place32(pfx_ctx *ctx);
pfx_status pfx_sell_place(pfx_ctx *ctx); // placement search for the solver's value array; the CG loops call it once a solve has proven long
#define PFX_PLACE_AFTER 192             // CG iterations of one solve before the search
#define PFX_PLACE_WORTH 256             // ... and only if at least this many iterations may still follow
pfx_status pfx_launch_local_linear(pfx_ctx *ctx);                                                   // strict arithmetic
// d_eloc_out: where the per-element energies go (null: the context's buffer)
pfx_status pfx_launch_local_nl(pfx_ctx *ctx, const double *d_u, bool energy, bool grad, bool hess, double *d_eloc_out = nullptr);
pfx_status pfx_launch_rows_linear(pfx_ctx *ctx, bool write_mask); // strict arithmetic
pfx_status pfx_launch_rows_nl(pfx_ctx *ctx);
pfx_status pfx_launch_rows_nl_psd(pfx_ctx *ctx);
pfx_status pfx_values_to_csr(pfx_ctx *ctx, double *host_values);
pfx_status pfx_newton_run(pfx_ctx *ctx, const double *d_rhs, double *d_x, const pfx_newton_params *params, pfx_newton_info *info);
// energy + gradient + Hessian state of one displacement in ONE element-kernel launch, and the Hessian assembled from
// that state afterwards (the Newton driver evaluates all three at every iterate)
pfx_status pfx_eval_all_dev(pfx_ctx *ctx, const double *d_u, double *d_energy, double *d_g);
pfx_status pfx_hessian_from_state(pfx_ctx *ctx, bool project_to_psd);
pfx_status pfx_capture_mass(pfx_ctx *ctx);
pfx_status pfx_lumped_mass_run(pfx_ctx *ctx, double *d_out);
pfx_status pfx_mass_apply_run(pfx_ctx *ctx, const double *d_v, double *d_y);
pfx_status pfx_add_mass_run(pfx_ctx *ctx, double w);
pfx_status pfx_implicit_euler_run(pfx_ctx *ctx, double dt, const double *d_rhs, double *d_x, double *d_v, double *d_a,
								  const pfx_newton_params *params, pfx_newton_info *info);
pfx_status pfx_launch_gather_vector(pfx_ctx *ctx, double *d_g, bool drop_tiny = true);
pfx_status pfx_launch_mass(pfx_ctx *ctx, double density, const double *d_density_e, const double *d_density_pts = nullptr);                       // strict arithmetic
pfx_status pfx_launch_local_rhs(pfx_ctx *ctx, const double *force, const double *d_force_pts, double density, const double *d_density_e, const double *d_density_pts = nullptr);
pfx_status pfx_launch_mass_quadrature_points(pfx_ctx *ctx, double *d_pts); // strict arithmetic
pfx_status pfx_launch_quadrature_points(pfx_ctx *ctx, double *d_pts);
pfx_status pfx_launch_reduce_energy(pfx_ctx *ctx, double *d_energy, const double *d_eloc = nullptr, double *d_partials = nullptr);
pfx_status pfx_noalloc_init(pfx_ctx *ctx);
pfx_status pfx_energy_noalloc_run(pfx_ctx *ctx, const double *d_u, double *d_energy);
pfx_status pfx_launch_spmv(pfx_ctx *ctx, const double *d_x, double *d_y);
void pfx_sell_release(pfx_ctx *ctx);
pfx_status pfx_sell_refresh(pfx_ctx *ctx);
int pfx_sell_grid(const pfx_ctx *ctx);
enum
{
	PFX_SPMV_ALL = 0,
	PFX_SPMV_INTERIOR = 1,
	PFX_SPMV_BOUNDARY = 2
};
// sliced: where given, a product that went through the slice-ordered buffer is LEFT there (*sliced = true; d_y untouched):
// the caller reads entry r of the product as ys[((pos[r / bs] >> 6) * bs + r % bs) * 64 + (pos[r / bs] & 63)]
pfx_status pfx_sell_spmv(pfx_ctx *ctx, const double *d_x, double *d_y, double *part, const int *done, int phase = PFX_SPMV_ALL, bool *sliced = nullptr, bool fp32 = false);
pfx_status pfx_sell_refresh32(pfx_ctx *ctx);
pfx_status pfx_sell_unpermute(pfx_ctx *ctx, double *d_y);
pfx_status pfx_cgs_fetch_product_impl(pfx_ctx *ctx);
void pfx_amg_release(pfx_ctx *ctx);
void pfx_ext_release(pfx_ctx *ctx);
void pfx_release_pattern(pfx_ctx *ctx);
pfx_status pfx_pcg_amg_run(pfx_ctx *ctx, const double *d_b, double *d_x, int max_iter, double tol, int *iters, double *rel_res);
pfx_status pfx_launch_invdiag(pfx_ctx *ctx, int jacobi, double *d_invdiag);
void pfx_launch_sum_partials(pfx_ctx *ctx, const double *part, int count, double *out);
void pfx_local_node_positions(int kind, int *n_loc, int *n_gloc, double (*pos)[3], double (*gval)[PFX_MAX_GLOC]);
pfx_status pfx_cg_run(pfx_ctx *ctx, const double *d_b, double *d_x, int precond, int max_iter, double tol, int *iters, double *rel_res);
pfx_status pfx_cgs_spmv_phase_impl(pfx_ctx *ctx, int phase, double *gscal);
pfx_status pfx_cgs_prepare_impl(pfx_ctx *ctx, double *d_x, int precond);
pfx_status pfx_cgs_residual_impl(pfx_ctx *ctx, const double *d_b, const double *d_x, double *gscal);
pfx_status pfx_cgs_start_impl(pfx_ctx *ctx, double *d_x, double tol, double *gscal);
pfx_status pfx_cgs_start2_impl(pfx_ctx *ctx, const double *gscal);
pfx_status pfx_cgs_spmv_impl(pfx_ctx *ctx, double *gscal);
pfx_status pfx_cgs_update_impl(pfx_ctx *ctx, int it, double *d_x, double *gscal);
pfx_status pfx_cgs_direction_impl(pfx_ctx *ctx, int it, const double *gscal);
pfx_status pfx_cgs_status_impl(pfx_ctx *ctx, int *done, int *iters, double *rr, double *bb);
pfx_status pfx_cg_direction_vector(pfx_ctx *ctx, double **p); // the search direction of the step functions (n_local dofs: its halo part is the caller's to fill)
pfx_status pfx_export_compressed(pfx_ctx *ctx, bool use_mask, bool drop_fixed, int64_t *n_rows, int64_t *nnz, int64_t *row_ptr, int32_t *col_idx, double *values);
pfx_status pfx_export_pattern(pfx_ctx *ctx, int64_t *row_ptr, int32_t *col_idx);
