// pfx_api.hip -- the extern "C" entry points declared in include/polyfem_amd.h: context management, uploads and
// the host-side sequencing of the kernels.  Reference call sites are cited in the header.
#include "pfx_internal.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>

namespace
{
	__global__ void k_set_fixed(const int32_t *__restrict__ dofs, int64_t n, uint8_t *__restrict__ fixed)
	{
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
			fixed[dofs[k]] = 1;
	}

	// g = b on constrained dofs, 0 elsewhere
	__global__ void k_dirichlet_lift(const double *__restrict__ b, const uint8_t *__restrict__ fixed, int64_t n, double *__restrict__ g)
	{
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
			g[k] = fixed[k] ? b[k] : 0.0;
	}

	// rhs = b - A g on free dofs (constrained entries are ignored by the solver)
	__global__ void k_dirichlet_rhs(const double *__restrict__ b, const double *__restrict__ Ag, const uint8_t *__restrict__ fixed, int64_t n, double *__restrict__ rhs)
	{
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
			rhs[k] = fixed[k] ? 0.0 : b[k] - Ag[k];
	}

	__global__ void k_dirichlet_fill(const double *__restrict__ b, const uint8_t *__restrict__ fixed, int64_t n, double *__restrict__ x)
	{
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
			if (fixed[k])
				x[k] = b[k];
	}

} // namespace

// everything that belongs to the current pattern (the mesh stays)
void pfx_release_pattern(pfx_ctx *ctx)
{
	{
		pfx_ext_release(ctx);
		free_dev(ctx, ctx->d_node_ptr);
		free_dev(ctx, ctx->d_node_col);
		free_dev(ctx, ctx->d_blk_src);
		free_dev(ctx, ctx->d_blk_slot);
		free_dev(ctx, ctx->d_grp_c0);
		free_dev(ctx, ctx->d_grp_win);
		free_dev(ctx, ctx->d_grp_order);
		ctx->h_grp_work.clear();
		free_dev(ctx, ctx->d_win);
		free_dev(ctx, ctx->d_desc);
		ctx->n_ublocks = ctx->n_upair = ctx->n_groups = ctx->n_windows = 0;
		ctx->h_grp_win.clear();
		free_dev(ctx, ctx->d_ne_ptr);
		free_dev(ctx, ctx->d_ne_src);
		free_dev(ctx, ctx->d_values);
		free_dev(ctx, ctx->d_mask);
		free_dev(ctx, ctx->d_estate);
		free_dev(ctx, ctx->d_kloc);
		ctx->kloc_count = 0;
		ctx->estate_count = 0;
		ctx->estate_kind = PFX_ESTATE_NONE;
		free_dev(ctx, ctx->d_gloc);
		free_dev(ctx, ctx->d_eloc);
		free_dev(ctx, ctx->d_fixed);
		free_dev(ctx, ctx->d_dirichlet);
		free_dev(ctx, ctx->d_work);
		free_dev(ctx, ctx->d_u);
		free_dev(ctx, ctx->d_vec);
		free_dev(ctx, ctx->d_part0);
		ctx->part0_count = 0;
		free_dev(ctx, ctx->d_newton);
		ctx->newton_count = 0;
		free_dev(ctx, ctx->d_mass_blk);
		free_dev(ctx, ctx->d_tstep);
		free_dev(ctx, ctx->d_al_store);
		ctx->al_count = 0;
		ctx->al_weight = 0.0;
		ctx->d_al_mass = ctx->d_al_target = nullptr;
		ctx->mass_count = ctx->tstep_count = 0;
		ctx->inertia_weight = 0.0;
		ctx->d_x_tilde = nullptr;
		pfx_sell_release(ctx);
		pfx_amg_release(ctx);
		ctx->struct_gen++;
		ctx->pattern_gen++;
		ctx->d_work_ext = nullptr; // a caller-owned workspace was sized for the pattern that is going away
		ctx->n_blocks = ctx->nnz = 0;
		ctx->n_dirichlet = 0;
		ctx->have_mask = false;
	}
}

namespace
{
	inline void release_pattern(pfx_ctx *ctx) { pfx_release_pattern(ctx); }

	// a matrix to work with: an assembled pattern, or a caller's matrix whose values are in place
	pfx_status need_pattern(pfx_ctx *ctx)
	{
		if (!ctx->d_node_ptr)
			PFX_FAIL(PFX_ERR_ARG, "pfx_build_pattern must be called first");
		return PFX_OK;
	}

	// entry points that run element kernels: the pattern must come from a mesh, not from pfx_solver_analyze
	pfx_status need_mesh(pfx_ctx *ctx)
	{
		if (need_pattern(ctx))
			return PFX_ERR_ARG;
		if (ctx->ext || ctx->kind < 0)
			PFX_FAIL(PFX_ERR_ARG, "this context holds a caller's matrix (pfx_solver_analyze), not a mesh");
		return PFX_OK;
	}

	// entry points that read the matrix values
	pfx_status need_matrix(pfx_ctx *ctx)
	{
		if (need_pattern(ctx))
			return PFX_ERR_ARG;
		if (ctx->ext && !ctx->ext_factorized)
			PFX_FAIL(PFX_ERR_ARG, "pfx_solver_factorize must succeed before the matrix is used");
		return PFX_OK;
	}

	pfx_status ensure_staging(pfx_ctx *ctx)
	{
		const int64_t n = ctx->n_dofs();
		if (!ctx->d_u && pfx_alloc(ctx, &ctx->d_u, n))
			return PFX_ERR_HIP;
		if (!ctx->d_vec && pfx_alloc(ctx, &ctx->d_vec, n + 1))
			return PFX_ERR_HIP;
		return PFX_OK;
	}

	// per-element state buffer of at least `count` doubles
	pfx_status ensure_estate(pfx_ctx *ctx, int64_t count)
	{
		if (ctx->d_estate && ctx->estate_count >= count)
			return PFX_OK;
		if (pfx_alloc(ctx, &ctx->d_estate, count))
			return PFX_ERR_HIP;
		ctx->estate_count = count;
		ctx->estate_kind = PFX_ESTATE_NONE;
		return PFX_OK;
	}

	// host array of n_el densities -> device (null: uniform)
	pfx_status upload_density(pfx_ctx *ctx, const double *density_e, const double **d_out)
	{
		*d_out = nullptr;
		if (!density_e)
			return PFX_OK;
		if (ctx->density_count < ctx->n_el)
		{
			if (pfx_alloc(ctx, &ctx->d_density_e, ctx->n_el))
				return PFX_ERR_HIP;
			ctx->density_count = ctx->n_el;
		}
		PFX_HIP(hipMemcpyAsync(ctx->d_density_e, density_e, sizeof(double) * (size_t)ctx->n_el, hipMemcpyHostToDevice, ctx->stream));
		*d_out = ctx->d_density_e;
		return PFX_OK;
	}

	pfx_status check_flag(pfx_ctx *ctx)
	{
		int h = 0;
		PFX_HIP(hipMemcpyAsync(&h, ctx->d_flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
		PFX_HIP(hipStreamSynchronize(ctx->stream));
		if (h)
		{
			PFX_HIP(hipMemsetAsync(ctx->d_flag, 0, sizeof(int), ctx->stream));
			PFX_FAIL(PFX_ERR_NUMERIC, "non-positive deformation-gradient determinant in at least one element (energy is not finite)");
		}
		return PFX_OK;
	}

	void resolve_timers(pfx_ctx *ctx)
	{
		if (ctx->ev_pending.empty())
			return;
		(void)hipStreamSynchronize(ctx->stream);
		for (auto &pp : ctx->ev_pending)
		{
			float ms = 0;
			if (hipEventElapsedTime(&ms, pp.e0, pp.e1) == hipSuccess)
			{
				auto &t = ctx->timers[pp.family];
				t.total_ms += ms;
				t.launches += 1;
			}
			ctx->ev_free.push_back(pp.e0);
			ctx->ev_free.push_back(pp.e1);
		}
		ctx->ev_pending.clear();
	}
} // namespace

extern "C"
{

	pfx_status pfx_create(pfx_ctx **out, int device)
	{
		if (!out)
			return PFX_ERR_ARG;
		*out = nullptr;
		int count = 0;
		if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count)
			return PFX_ERR_HIP;
		if (hipSetDevice(device) != hipSuccess)
			return PFX_ERR_HIP;
		pfx_ctx *ctx = new (std::nothrow) pfx_ctx();
		if (!ctx)
			return PFX_ERR_ARG;
		ctx->device = device;
		if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess)
		{
			delete ctx;
			return PFX_ERR_HIP;
		}
		ctx->own_stream = true;
		if (hipMalloc((void **)&ctx->d_tab, sizeof(pfx_tables)) != hipSuccess || hipMalloc((void **)&ctx->d_tab_mass, sizeof(pfx_tables)) != hipSuccess || hipMalloc((void **)&ctx->d_flag, sizeof(int)) != hipSuccess || hipMalloc((void **)&ctx->d_partials, sizeof(double) * (3 * 2048 + 8)) != hipSuccess || hipMemset(ctx->d_partials, 0, sizeof(double) * (3 * 2048 + 8)) != hipSuccess || hipMemset(ctx->d_flag, 0, sizeof(int)) != hipSuccess)
		{
			pfx_destroy(ctx); // releases whatever was allocated so far, and the stream
			return PFX_ERR_HIP;
		}
		if (pfx_noalloc_init(ctx) != PFX_OK)
		{
			pfx_destroy(ctx);
			return PFX_ERR_HIP;
		}
		*out = ctx;
		return PFX_OK;
	}

	void pfx_destroy(pfx_ctx *ctx)
	{
		if (!ctx)
			return;
		(void)hipSetDevice(ctx->device);
		(void)hipStreamSynchronize(ctx->stream);
		release_pattern(ctx);
		free_dev(ctx, ctx->d_conn);
		free_dev(ctx, ctx->d_gconn);
		free_dev(ctx, ctx->d_xyz);
		free_dev(ctx, ctx->d_tab);
		free_dev(ctx, ctx->d_tab_mass);
		free_dev(ctx, ctx->d_density_e);
		free_dev(ctx, ctx->d_flag);
		free_dev(ctx, ctx->d_elem_owned);
		free_dev(ctx, ctx->d_lambda_e);
		free_dev(ctx, ctx->d_mu_e);
		free_dev(ctx, ctx->d_model_e);
		free_dev(ctx, ctx->d_tensor);
		free_dev(ctx, ctx->d_partials);
		free_dev(ctx, ctx->d_cgstate);
		resolve_timers(ctx);
		for (auto e : ctx->ev_free)
			(void)hipEventDestroy(e);
		if (ctx->own_stream)
			(void)hipStreamDestroy(ctx->stream);
		delete ctx;
	}

	const char *pfx_last_error(const pfx_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

	pfx_status pfx_set_stream(pfx_ctx *ctx, void *hip_stream)
	{
		if (!ctx)
			return PFX_ERR_ARG;
		PFX_HIP(hipStreamSynchronize(ctx->stream));
		if (hip_stream)
		{
			if (ctx->own_stream)
				(void)hipStreamDestroy(ctx->stream);
			ctx->stream = (hipStream_t)hip_stream;
			ctx->own_stream = false;
		}
		else if (!ctx->own_stream)
		{
			PFX_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
			ctx->own_stream = true;
		}
		return PFX_OK;
	}

	pfx_status pfx_synchronize(pfx_ctx *ctx)
	{
		if (!ctx)
			return PFX_ERR_ARG;
		PFX_HIP(hipStreamSynchronize(ctx->stream));
		return PFX_OK;
	}

	pfx_status pfx_set_mesh(pfx_ctx *ctx, int elem_kind, int64_t n_el, int64_t n_nodes, int64_t n_gnodes,
							const int32_t *conn, const int32_t *gconn, const double *geom_xyz, int quad_order)
	{
		if (!ctx)
			return PFX_ERR_ARG;
		if (elem_kind < PFX_P1TET || elem_kind > PFX_P2TRI)
			PFX_FAIL(PFX_ERR_ARG, "unknown element kind");
		const bool tri = elem_kind == PFX_P1TRI || elem_kind == PFX_P2TRI;
		const int mesh_dim = tri ? 2 : 3;
		if (n_el <= 0 || n_nodes <= 0 || n_gnodes <= 0 || !conn || !gconn || !geom_xyz)
			PFX_FAIL(PFX_ERR_ARG, "empty mesh or null pointer");
		if (n_nodes >= ((int64_t)1 << 31) / 3)
			PFX_FAIL(PFX_ERR_SIZE, "n_nodes*dim must fit the reference's 32-bit column index");
		PFX_HIP(hipSetDevice(ctx->device));
		// the tables go into the context only once every argument has been accepted: a rejected call leaves the
		// previous mesh fully intact
		static thread_local pfx_tables tab, tab_mass;
		pfx_fill_tables(elem_kind, quad_order, &tab);
		pfx_fill_tables_mass(elem_kind, &tab_mass);
		if (tab.n_pts == 0)
			PFX_FAIL(PFX_ERR_UNSUPPORTED, "quadrature order not available for this element kind");
		const int n_loc = tab.n_loc, n_gloc = tab.n_gloc;
		// the kernels index device arrays with these values: validate them on the host
		for (int64_t k = 0; k < n_el * n_loc; ++k)
			if (conn[k] < 0 || conn[k] >= n_nodes)
				PFX_FAIL(PFX_ERR_ARG, "conn entry out of range");
		for (int64_t k = 0; k < n_el * n_gloc; ++k)
			if (gconn[k] < 0 || gconn[k] >= n_gnodes)
				PFX_FAIL(PFX_ERR_ARG, "gconn entry out of range");
		release_pattern(ctx);
		ctx->h_tab = tab;
		ctx->h_tab_mass = tab_mass;
		ctx->kind = elem_kind;
		ctx->gkind = (elem_kind == PFX_Q1HEX || elem_kind == PFX_Q2HEX) ? PFX_Q1HEX : tri ? PFX_P1TRI : PFX_P1TET;
		ctx->mesh_dim = mesh_dim;
		ctx->n_loc = n_loc;
		ctx->n_gloc = n_gloc;
		ctx->n_pts = ctx->h_tab.n_pts;
		ctx->quad_order = quad_order;
		ctx->n_el = n_el;
		ctx->n_nodes = n_nodes;
		ctx->n_owned = n_nodes;
		free_dev(ctx, ctx->d_elem_owned);
		free_dev(ctx, ctx->d_lambda_e);
		free_dev(ctx, ctx->d_mu_e);
		free_dev(ctx, ctx->d_model_e);
		ctx->n_gnodes = n_gnodes;
		if (pfx_alloc(ctx, &ctx->d_conn, n_el * n_loc) || pfx_alloc(ctx, &ctx->d_gconn, n_el * n_gloc) || pfx_alloc(ctx, &ctx->d_xyz, n_gnodes * 3))
			return PFX_ERR_HIP;
		PFX_HIP(hipMemcpyAsync(ctx->d_conn, conn, sizeof(int32_t) * (size_t)(n_el * n_loc), hipMemcpyHostToDevice, ctx->stream));
		PFX_HIP(hipMemcpyAsync(ctx->d_gconn, gconn, sizeof(int32_t) * (size_t)(n_el * n_gloc), hipMemcpyHostToDevice, ctx->stream));
		// the device copy holds three coordinates per geometry node; triangle meshes get a zero third one (the kernels then
		// work with the 2 x 2 Jacobian sitting in the upper left of a 3 x 3 matrix whose last diagonal entry is one)
		std::vector<double> padded;
		if (tri)
		{
			padded.resize((size_t)n_gnodes * 3);
			for (int64_t g = 0; g < n_gnodes; ++g)
				padded[3 * g] = geom_xyz[2 * g], padded[3 * g + 1] = geom_xyz[2 * g + 1], padded[3 * g + 2] = 0.0;
			geom_xyz = padded.data();
		}
		PFX_HIP(hipMemcpyAsync(ctx->d_xyz, geom_xyz, sizeof(double) * (size_t)(n_gnodes * 3), hipMemcpyHostToDevice, ctx->stream));
		PFX_HIP(hipMemcpyAsync(ctx->d_tab, &ctx->h_tab, sizeof(pfx_tables), hipMemcpyHostToDevice, ctx->stream));
		PFX_HIP(hipMemcpyAsync(ctx->d_tab_mass, &ctx->h_tab_mass, sizeof(pfx_tables), hipMemcpyHostToDevice, ctx->stream));
		PFX_HIP(hipStreamSynchronize(ctx->stream));
		return PFX_OK;
	}

	pfx_status pfx_set_material(pfx_ctx *ctx, int model, int dim, double lambda, double mu)
	{
		if (!ctx)
			return PFX_ERR_ARG;
		if (model < PFX_LAPLACIAN || model > PFX_SAINT_VENANT)
			PFX_FAIL(PFX_ERR_ARG, "unknown model (PFX_HOOKE_LINEAR is selected by pfx_set_elasticity_tensor)");
		if (dim != 3 && dim != 2)
			PFX_FAIL(PFX_ERR_UNSUPPORTED, "dim is 3, or 2 for triangle meshes");
		if (ctx->kind >= 0 && model != PFX_LAPLACIAN && dim != ctx->mesh_dim)
			PFX_FAIL(PFX_ERR_ARG, "dim does not match the mesh (2 for triangles, 3 otherwise)");
		const int bs = (model == PFX_LAPLACIAN) ? 1 : dim;
		if (ctx->d_node_ptr && bs != ctx->bs)
			release_pattern(ctx); // block size changes the layout
		ctx->model = model;
		ctx->dim = dim;
		ctx->bs = bs;
		ctx->lambda = lambda;
		ctx->mu = mu;
		free_dev(ctx, ctx->d_lambda_e);
		free_dev(ctx, ctx->d_mu_e);
		free_dev(ctx, ctx->d_model_e);
		ctx->mat_per_point = 0;
		ctx->stvk_tensor = false;
		return PFX_OK;
	}

	pfx_status pfx_set_saint_venant_tensor(pfx_ctx *ctx, int dim, const double *tensor36)
	{
		// same upload as for HookeLinearElasticity; the model stays the nonlinear one
		pfx_status rc = pfx_set_elasticity_tensor(ctx, dim, tensor36);
		if (rc)
			return rc;
		ctx->model = PFX_SAINT_VENANT;
		ctx->stvk_tensor = true;
		ctx->estate_kind = PFX_ESTATE_NONE;
		return PFX_OK;
	}

	pfx_status pfx_set_elasticity_tensor(pfx_ctx *ctx, int dim, const double *tensor36)
	{
		if (!ctx || !tensor36)
			return PFX_ERR_ARG;
		if (dim != 3 && dim != 2)
			PFX_FAIL(PFX_ERR_UNSUPPORTED, "dim is 3, or 2 for triangle meshes");
		const int nv = dim == 3 ? 6 : 3;
		for (int a = 0; a < nv; ++a)
			for (int b = 0; b < a; ++b)
				if (tensor36[a * nv + b] != tensor36[b * nv + a])
					PFX_FAIL(PFX_ERR_ARG, "the elasticity tensor must be symmetric");
		pfx_status rc = pfx_set_material(ctx, PFX_LINEAR_ELASTICITY, dim, 0.0, 0.0);
		if (rc)
			return rc;
		// the kernels read a 6 x 6 tensor; the 3 x 3 one of a 2-D problem (xx yy xy) goes into its xx, yy, xy rows and
		// columns, zeros elsewhere: the out-of-plane strains of the embedded 2-D problem are zero
		double full[36] = {};
		if (dim == 3)
			std::copy(tensor36, tensor36 + 36, full);
		else
		{
			const int at[3] = {0, 1, 5};
			for (int a = 0; a < 3; ++a)
				for (int b = 0; b < 3; ++b)
					full[at[a] * 6 + at[b]] = tensor36[a * 3 + b];
		}
		PFX_HIP(hipSetDevice(ctx->device));
		if (!ctx->d_tensor && pfx_alloc(ctx, &ctx->d_tensor, 36))
			return PFX_ERR_HIP;
		PFX_HIP(hipMemcpyAsync(ctx->d_tensor, full, sizeof(double) * 36, hipMemcpyHostToDevice, ctx->stream));
		PFX_HIP(hipStreamSynchronize(ctx->stream));
		ctx->model = PFX_HOOKE_LINEAR;
		return PFX_OK;
	}

	pfx_status pfx_set_element_models(pfx_ctx *ctx, const int32_t *model_per_element, int64_t n_el)
	{
		if (!ctx || !model_per_element)
			return PFX_ERR_ARG;
		if (ctx->n_el <= 0 || n_el != ctx->n_el)
			PFX_FAIL(PFX_ERR_ARG, "one model per element of the current mesh expected");
		if (ctx->bs < 2)
			PFX_FAIL(PFX_ERR_ARG, "pfx_set_material(_per_element) with an elastic model must come first");
		if (ctx->stvk_tensor || ctx->model == PFX_HOOKE_LINEAR)
			PFX_FAIL(PFX_ERR_UNSUPPORTED, "element models take their (lambda, mu) from pfx_set_material(_per_element), not an elasticity tensor");
		std::vector<uint8_t> m((size_t)n_el);
		for (int64_t e = 0; e < n_el; ++e)
			switch (model_per_element[e])
			{
			case PFX_LINEAR_ELASTICITY: m[e] = 0; break;
			case PFX_NEOHOOKEAN: m[e] = 1; break;
			case PFX_SAINT_VENANT: m[e] = 2; break;
			default: PFX_FAIL(PFX_ERR_ARG, "element models are PFX_LINEAR_ELASTICITY, PFX_NEOHOOKEAN or PFX_SAINT_VENANT");
			}
		PFX_HIP(hipSetDevice(ctx->device));
		free_dev(ctx, ctx->d_model_e);
		if (pfx_alloc(ctx, &ctx->d_model_e, n_el))
			return PFX_ERR_HIP;
		PFX_HIP(hipMemcpyAsync(ctx->d_model_e, m.data(), (size_t)n_el, hipMemcpyHostToDevice, ctx->stream));
		PFX_HIP(hipStreamSynchronize(ctx->stream));
		ctx->model = PFX_MULTI_MODEL;
		ctx->estate_kind = PFX_ESTATE_NONE;
		return PFX_OK;
	}

	pfx_status pfx_set_material_per_point(pfx_ctx *ctx, int model, int dim, const double *lambda, const double *mu, int64_t n_el, int n_pts)
	{
		if (!ctx || !lambda || !mu)
			return PFX_ERR_ARG;
		if (ctx->kind < 0 || n_el != ctx->n_el || n_pts != ctx->n_pts)
			PFX_FAIL(PFX_ERR_ARG, "pfx_set_material_per_point: call pfx_set_mesh first and pass one value per element and quadrature point");
		if (model == PFX_LAPLACIAN)
			PFX_FAIL(PFX_ERR_ARG, "the Laplacian has no material parameters");
		pfx_status rc = pfx_set_material(ctx, model, dim, n_el > 0 ? lambda[0] : 0.0, n_el > 0 ? mu[0] : 0.0);
		if (rc)
			return rc;
		PFX_HIP(hipSetDevice(ctx->device));
		const int64_t count = n_el * n_pts;
		if (pfx_alloc(ctx, &ctx->d_lambda_e, count) || pfx_alloc(ctx, &ctx->d_mu_e, count))
			return PFX_ERR_HIP;
		PFX_HIP(hipMemcpyAsync(ctx->d_lambda_e, lambda, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, ctx->stream));
		PFX_HIP(hipMemcpyAsync(ctx->d_mu_e, mu, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, ctx->stream));
		PFX_HIP(hipStreamSynchronize(ctx->stream));
		ctx->mat_per_point = 1;
		return PFX_OK;
	}

	pfx_status pfx_set_material_per_element(pfx_ctx *ctx, int model, int dim, const double *lambda, const double *mu, int64_t n_el)
	{
		if (!ctx || !lambda || !mu)
			return PFX_ERR_ARG;
		if (ctx->kind < 0 || n_el != ctx->n_el)
			PFX_FAIL(PFX_ERR_ARG, "pfx_set_material_per_element: call pfx_set_mesh first and pass one value per element");
		if (model == PFX_LAPLACIAN)
			PFX_FAIL(PFX_ERR_ARG, "the Laplacian has no material parameters");
		pfx_status rc = pfx_set_material(ctx, model, dim, n_el > 0 ? lambda[0] : 0.0, n_el > 0 ? mu[0] : 0.0);
		if (rc)
			return rc;
		PFX_HIP(hipSetDevice(ctx->device));
		if (pfx_alloc(ctx, &ctx->d_lambda_e, n_el) || pfx_alloc(ctx, &ctx->d_mu_e, n_el))
			return PFX_ERR_HIP;
		PFX_HIP(hipMemcpyAsync(ctx->d_lambda_e, lambda, sizeof(double) * (size_t)n_el, hipMemcpyHostToDevice, ctx->stream));
		PFX_HIP(hipMemcpyAsync(ctx->d_mu_e, mu, sizeof(double) * (size_t)n_el, hipMemcpyHostToDevice, ctx->stream));
		PFX_HIP(hipStreamSynchronize(ctx->stream));
		return PFX_OK;
	}

	pfx_status pfx_build_pattern(pfx_ctx *ctx, int64_t *nnz)
	{
		if (!ctx)
			return PFX_ERR_ARG;
		if (ctx->kind < 0 || ctx->model < 0)
			PFX_FAIL(PFX_ERR_ARG, "pfx_set_mesh and pfx_set_material must be called first");
		if (ctx->model != PFX_LAPLACIAN && ctx->dim != ctx->mesh_dim)
			PFX_FAIL(PFX_ERR_ARG, "the material's dim does not match the mesh (2 for triangles, 3 otherwise)");
		PFX_HIP(hipSetDevice(ctx->device));
		release_pattern(ctx);
		pfx_status rc = pfx_pattern_build(ctx);
		if (rc)
			return rc;
		const int N = ctx->n_loc * ctx->bs;
		if (pfx_alloc(ctx, &ctx->d_values, ctx->nnz + 2) || pfx_alloc(ctx, &ctx->d_mask, ctx->nnz) || pfx_alloc(ctx, &ctx->d_gloc, ctx->n_el * N) || pfx_alloc(ctx, &ctx->d_eloc, ctx->n_el) || pfx_alloc(ctx, &ctx->d_fixed, ctx->n_dofs()))
			return PFX_ERR_HIP;
		ctx->estate_count = 0;
		if (ensure_estate(ctx, ctx->n_el * (int64_t)(22 * ctx->n_pts)))
			return PFX_ERR_HIP;
		PFX_HIP(hipMemsetAsync(ctx->d_values, 0, sizeof(double) * (size_t)(ctx->nnz + 2), ctx->stream));
		PFX_HIP(hipMemsetAsync(ctx->d_fixed, 0, (size_t)ctx->n_dofs(), ctx->stream));
		PFX_HIP(hipStreamSynchronize(ctx->stream));
		if (nnz)
			*nnz = ctx->nnz;
		return PFX_OK;
	}

	pfx_status pfx_get_pattern(pfx_ctx *ctx, int64_t *row_ptr, int32_t *col_idx)
	{
		if (!ctx || !row_ptr || !col_idx)
			return PFX_ERR_ARG;
		if (need_pattern(ctx))
			return PFX_ERR_ARG;
		return pfx_export_pattern(ctx, row_ptr, col_idx);
	}

	pfx_status pfx_get_values(pfx_ctx *ctx, double *values)
	{
		if (!ctx || !values)
			return PFX_ERR_ARG;
		if (need_matrix(ctx))
			return PFX_ERR_ARG;
		return pfx_values_to_csr(ctx, values);
	}

	pfx_status pfx_assemble_linear(pfx_ctx *ctx)
	{
		if (!ctx)
			return PFX_ERR_ARG;
		if (need_mesh(ctx))
			return PFX_ERR_ARG;
		if (ctx->model != PFX_LAPLACIAN && ctx->model != PFX_LINEAR_ELASTICITY && ctx->model != PFX_HOOKE_LINEAR)
			PFX_FAIL(PFX_ERR_ARG, "pfx_assemble_linear needs the Laplacian, LinearElasticity or HookeLinearElasticity model");
		{
			pfx_scoped_timer tm(ctx, "local");
			pfx_status rc = pfx_launch_local_linear(ctx);
			if (rc)
				return rc;
		}
		{
			pfx_scoped_timer tm(ctx, "gather");
			pfx_status rc = pfx_launch_rows_linear(ctx, true);
			if (rc)
				return rc;
		}
		ctx->have_mask = true;
		ctx->values_gen++;
		return PFX_OK;
	}

	pfx_status pfx_assemble_mass(pfx_ctx *ctx, double density, const double *density_e)
	{
		if (!ctx)
			return PFX_ERR_ARG;
		if (need_mesh(ctx))
			return PFX_ERR_ARG;
		PFX_HIP(hipSetDevice(ctx->device));
		const bool hex = ctx->kind == PFX_Q1HEX || ctx->kind == PFX_Q2HEX;
		if (ensure_estate(ctx, ctx->n_el * (int64_t)(hex ? 10 * ctx->h_tab_mass.n_pts : 10)))
			return PFX_ERR_HIP;
		const double *d_rho = nullptr;
		pfx_status rc = upload_density(ctx, density_e, &d_rho);
		if (rc)
			return rc;
		{
			pfx_scoped_timer tm(ctx, "mass");
			if ((rc = pfx_launch_mass(ctx, density, d_rho)))
				return rc;
		}
		ctx->have_mask = true;
		ctx->values_gen++;
		return PFX_OK;
	}

	// mass matrix with the density evaluated at every point of the mass quadrature rule
	static pfx_status assemble_mass_pointwise_impl(pfx_ctx *ctx, const double *density_pts, int64_t n_el, int n_pts)
	{
		if (need_mesh(ctx))
			return PFX_ERR_ARG;
		if (n_el != ctx->n_el || n_pts != ctx->h_tab_mass.n_pts)
			PFX_FAIL(PFX_ERR_ARG, "one density per element and point of the mass quadrature rule expected (see pfx_mass_quadrature_points)");
		PFX_HIP(hipSetDevice(ctx->device));
		const bool hex = ctx->kind == PFX_Q1HEX || ctx->kind == PFX_Q2HEX;
		if (ensure_estate(ctx, ctx->n_el * (int64_t)(hex ? 10 * ctx->h_tab_mass.n_pts : 10)))
			return PFX_ERR_HIP;
		pfx_tmp<double> d_rho;
		const size_t count = (size_t)(n_el * n_pts);
		PFX_HIP(d_rho.alloc(count));
		PFX_HIP(hipMemcpyAsync(d_rho.p, density_pts, sizeof(double) * count, hipMemcpyHostToDevice, ctx->stream));
		pfx_status rc;
		{
			pfx_scoped_timer tm(ctx, "mass");
			rc = pfx_launch_mass(ctx, 0.0, nullptr, d_rho.p);
		}
		PFX_HIP(hipStreamSynchronize(ctx->stream)); // the temporary goes away on return
		if (rc)
			return rc;
		ctx->have_mask = true;
		ctx->values_gen++;
		return PFX_OK;
	}

	pfx_status pfx_assemble_mass_pointwise(pfx_ctx *ctx, const double *density_pts, int64_t n_el, int n_pts)
	{
		if (!ctx || !density_pts)
			return PFX_ERR_ARG;
		return assemble_mass_pointwise_impl(ctx, density_pts, n_el, n_pts);
	}

	pfx_status pfx_inertia_set_mass_pointwise(pfx_ctx *ctx, const double *density_pts, int64_t n_el, int n_pts)
	{
		if (!ctx || !density_pts)
			return PFX_ERR_ARG;
		if (ctx->bs < 2)
			PFX_FAIL(PFX_ERR_UNSUPPORTED, "the inertia term needs a vector-valued model (block size 2 or 3)");
		pfx_status rc = assemble_mass_pointwise_impl(ctx, density_pts, n_el, n_pts);
		if (rc)
			return rc;
		return pfx_capture_mass(ctx);
	}

	// the kernels write three coordinates per point; a 2-D mesh hands out two
	static pfx_status copy_points_out(pfx_ctx *ctx, const double *d_p, size_t n_points, double *pts)
	{
		if (ctx->mesh_dim == 3)
		{
			PFX_HIP(hipMemcpyAsync(pts, d_p, sizeof(double) * n_points * 3, hipMemcpyDeviceToHost, ctx->stream));
			PFX_HIP(hipStreamSynchronize(ctx->stream));
			return PFX_OK;
		}
		std::vector<double> h(n_points * 3);
		PFX_HIP(hipMemcpyAsync(h.data(), d_p, sizeof(double) * n_points * 3, hipMemcpyDeviceToHost, ctx->stream));
		PFX_HIP(hipStreamSynchronize(ctx->stream));
		for (size_t k = 0; k < n_points; ++k)
			pts[2 * k] = h[3 * k], pts[2 * k + 1] = h[3 * k + 1];
		return PFX_OK;
	}

	pfx_status pfx_mass_quadrature_points(pfx_ctx *ctx, int *n_pts, double *pts)
	{
		if (!ctx || !n_pts)
			return PFX_ERR_ARG;
		if (ctx->kind < 0 || ctx->n_el <= 0)
			PFX_FAIL(PFX_ERR_ARG, "pfx_set_mesh must be called first");
		*n_pts = ctx->h_tab_mass.n_pts;
		if (!pts)
			return PFX_OK;
		PFX_HIP(hipSetDevice(ctx->device));
		const size_t count = (size_t)(ctx->n_el * ctx->h_tab_mass.n_pts * 3);
		pfx_tmp<double> d_p;
		PFX_HIP(d_p.alloc(count));
		pfx_status rc = pfx_launch_mass_quadrature_points(ctx, d_p.p);
		if (rc)
			return rc;
		return copy_points_out(ctx, d_p.p, count / 3, pts);
	}

	pfx_status pfx_assemble_rhs_density_pointwise(pfx_ctx *ctx, const double *force, const double *force_pts, const double *density_pts, int64_t n_el,
												  int n_pts, double *rhs)
	{
		if (!ctx || !density_pts || !rhs || (!force == !force_pts))
			return PFX_ERR_ARG;
		if (need_mesh(ctx) || ensure_staging(ctx))
			return PFX_ERR_ARG;
		if (n_el != ctx->n_el || n_pts != ctx->n_pts)
			PFX_FAIL(PFX_ERR_ARG, "one density per element and quadrature point of the stiffness rule expected (see pfx_quadrature_points)");
		PFX_HIP(hipSetDevice(ctx->device));
		pfx_tmp<double> d_rho, d_f;
		const size_t count = (size_t)(n_el * n_pts);
		PFX_HIP(d_rho.alloc(count));
		PFX_HIP(hipMemcpyAsync(d_rho.p, density_pts, sizeof(double) * count, hipMemcpyHostToDevice, ctx->stream));
		if (force_pts)
		{
			PFX_HIP(d_f.alloc(count * (size_t)ctx->bs));
			PFX_HIP(hipMemcpyAsync(d_f.p, force_pts, sizeof(double) * count * (size_t)ctx->bs, hipMemcpyHostToDevice, ctx->stream));
		}
		pfx_status rc = pfx_launch_local_rhs(ctx, force, force_pts ? d_f.p : nullptr, 0.0, nullptr, d_rho.p);
		if (!rc)
			rc = pfx_launch_gather_vector(ctx, ctx->d_vec, false);
		if (rc)
		{
			(void)hipStreamSynchronize(ctx->stream);
			return rc;
		}
		PFX_HIP(hipMemcpyAsync(rhs, ctx->d_vec, sizeof(double) * (size_t)ctx->n_dofs(), hipMemcpyDeviceToHost, ctx->stream));
		PFX_HIP(hipStreamSynchronize(ctx->stream));
		return PFX_OK;
	}

	pfx_status pfx_assemble_rhs_dev(pfx_ctx *ctx, const double *force, double density, const double *density_e, double *d_rhs)
	{
		if (!ctx || !force || !d_rhs)
			return PFX_ERR_ARG;
		if (need_mesh(ctx))
			return PFX_ERR_ARG;
		PFX_HIP(hipSetDevice(ctx->device));
		const double *d_rho = nullptr;
		pfx_status rc = upload_density(ctx, density_e, &d_rho);
		if (rc)
			return rc;
		if ((rc = pfx_launch_local_rhs(ctx, force, nullptr, density, d_rho)))
			return rc;
		return pfx_launch_gather_vector(ctx, d_rhs, false);
	}

	pfx_status pfx_assemble_rhs(pfx_ctx *ctx, const double *force, double density, const double *density_e, double *rhs)
	{
		if (!ctx || !force || !rhs)
			return PFX_ERR_ARG;
		if (need_mesh(ctx) || ensure_staging(ctx))
			return PFX_ERR_ARG;
		pfx_status rc = pfx_assemble_rhs_dev(ctx, force, density, density_e, ctx->d_vec);
		if (rc)
			return rc;
		PFX_HIP(hipMemcpyAsync(rhs, ctx->d_vec, sizeof(double) * (size_t)ctx->n_dofs(), hipMemcpyDeviceToHost, ctx->stream));
		PFX_HIP(hipStreamSynchronize(ctx->stream));
		return PFX_OK;
	}

	pfx_status pfx_assemble_rhs_pointwise_dev(pfx_ctx *ctx, const double *d_force_pts, double density, const double *density_e, double *d_rhs)
	{
		if (!ctx || !d_force_pts || !d_rhs)
			return PFX_ERR_ARG;
		if (need_mesh(ctx))
			return PFX_ERR_ARG;
		PFX_HIP(hipSetDevice(ctx->device));
		const double *d_rho = nullptr;
		pfx_status rc = upload_density(ctx, density_e, &d_rho);
		if (rc)
			return rc;
		if ((rc = pfx_launch_local_rhs(ctx, nullptr, d_force_pts, density, d_rho)))
			return rc;
		return pfx_launch_gather_vector(ctx, d_rhs, false);
	}

	pfx_status pfx_assemble_rhs_pointwise(pfx_ctx *ctx, const double *force_pts, double density, const double *density_e, double *rhs)
	{
		if (!ctx || !force_pts || !rhs)
			return PFX_ERR_ARG;
		if (need_mesh(ctx) || ensure_staging(ctx))
			return PFX_ERR_ARG;
		const size_t count = (size_t)(ctx->n_el * ctx->n_pts * ctx->bs);
		pfx_tmp<double> d_f;
		PFX_HIP(d_f.alloc(count > 0 ? count : 1));
		PFX_HIP(hipMemcpyAsync(d_f.p, force_pts, sizeof(double) * count, hipMemcpyHostToDevice, ctx->stream));
		pfx_status rc = pfx_assemble_rhs_pointwise_dev(ctx, d_f.p, density, density_e, ctx->d_vec);
		if (rc)
			return rc;
		PFX_HIP(hipMemcpyAsync(rhs, ctx->d_vec, sizeof(double) * (size_t)ctx->n_dofs(), hipMemcpyDeviceToHost, ctx->stream));
		PFX_HIP(hipStreamSynchronize(ctx->stream));
		return PFX_OK;
	}

	pfx_status pfx_quadrature_points(pfx_ctx *ctx, int *n_pts, double *pts)
	{
		if (!ctx || !n_pts)
			return PFX_ERR_ARG;
		if (ctx->kind < 0 || ctx->n_el <= 0)
			PFX_FAIL(PFX_ERR_ARG, "pfx_set_mesh must be called first");
		*n_pts = ctx->n_pts;
		if (!pts)
			return PFX_OK;
		PFX_HIP(hipSetDevice(ctx->device));
		const size_t count = (size_t)(ctx->n_el * ctx->n_pts * 3);
		pfx_tmp<double> d_p;
		PFX_HIP(d_p.alloc(count));
		pfx_status rc = pfx_launch_quadrature_points(ctx, d_p.p);
		if (rc)
			return rc;
		return copy_points_out(ctx, d_p.p, count / 3, pts);
	}

	pfx_status pfx_get_linear_csr(pfx_ctx *ctx, int64_t *nnz, int64_t *row_ptr, int32_t *col_idx, double *values)
	{
		if (!ctx || !nnz)
			return PFX_ERR_ARG;
		if (need_mesh(ctx))
			return PFX_ERR_ARG;
		if (!ctx->have_mask)
			PFX_FAIL(PFX_ERR_ARG, "pfx_assemble_linear must be called first");
		int64_t n_rows = 0;
		return pfx_export_compressed(ctx, true, false, &n_rows, nnz, row_ptr, col_idx, values);
	}

	pfx_status pfx_energy_dev(pfx_ctx *ctx, const double *d_u, double *d_energy)
	{
		if (!ctx || !d_u || !d_energy)
			return PFX_ERR_ARG;
		if (need_mesh(ctx))
			return PFX_ERR_ARG;
		PFX_HIP(hipSetDevice(ctx->device));
		{
			pfx_scoped_timer tm(ctx, "local_energy");
			pfx_status rc = pfx_launch_local_nl(ctx, d_u, true, false, false);
			if (rc)
				return rc;
		}
		return pfx_launch_reduce_energy(ctx, d_energy);
	}

	pfx_status pfx_energy_dev_noalloc(pfx_ctx *ctx, const double *d_u, double *d_energy)
	{
		if (!ctx || !d_u || !d_energy)
			return PFX_ERR_ARG;
		if (ctx->kind < 0 || ctx->model < 0 || ctx->n_el <= 0)
			PFX_FAIL(PFX_ERR_ARG, "pfx_set_mesh and pfx_set_material must be called first");
		return pfx_energy_noalloc_run(ctx, d_u, d_energy);
	}

	pfx_status pfx_gradient_dev(pfx_ctx *ctx, const double *d_u, double *d_g)
	{
		if (!ctx || !d_u || !d_g)
			return PFX_ERR_ARG;
		if (need_mesh(ctx))
			return PFX_ERR_ARG;
		{
			pfx_scoped_timer tm(ctx, "local_grad");
			pfx_status rc = pfx_launch_local_nl(ctx, d_u, false, true, false);
			if (rc)
				return rc;
		}
		pfx_scoped_timer tm(ctx, "gather_vec");
		return pfx_launch_gather_vector(ctx, d_g);
	}

} // extern "C"

pfx_status pfx_eval_all_dev(pfx_ctx *ctx, const double *d_u, double *d_energy, double *d_g)
{
	{
		pfx_scoped_timer tm(ctx, "local");
		pfx_status rc = pfx_launch_local_nl(ctx, d_u, true, true, true);
		if (rc)
			return rc;
	}
	pfx_status rc = pfx_launch_reduce_energy(ctx, d_energy);
	if (rc)
		return rc;
	pfx_scoped_timer tm(ctx, "gather_vec");
	return pfx_launch_gather_vector(ctx, d_g);
}

pfx_status pfx_hessian_from_state(pfx_ctx *ctx, bool project_to_psd)
{
	ctx->have_mask = false;
	ctx->values_gen++;
	pfx_scoped_timer tm(ctx, project_to_psd ? "gather_psd" : "gather");
	return project_to_psd ? pfx_launch_rows_nl_psd(ctx) : pfx_launch_rows_nl(ctx);
}

extern "C"
{
	static pfx_status hessian_dev_impl(pfx_ctx *ctx, const double *d_u, bool project_to_psd)
	{
		if (!ctx || !d_u)
			return PFX_ERR_ARG;
		if (need_mesh(ctx))
			return PFX_ERR_ARG;
		{
			pfx_scoped_timer tm(ctx, "local");
			pfx_status rc = pfx_launch_local_nl(ctx, d_u, false, false, true);
			if (rc)
				return rc;
		}
		ctx->have_mask = false;
		ctx->values_gen++;
		pfx_scoped_timer tm(ctx, project_to_psd ? "gather_psd" : "gather");
		return project_to_psd ? pfx_launch_rows_nl_psd(ctx) : pfx_launch_rows_nl(ctx);
	}

	pfx_status pfx_hessian_dev(pfx_ctx *ctx, const double *d_u)
	{
		return hessian_dev_impl(ctx, d_u, false);
	}

	pfx_status pfx_hessian_psd_dev(pfx_ctx *ctx, const double *d_u)
	{
		return hessian_dev_impl(ctx, d_u, true);
	}

	pfx_status pfx_energy(pfx_ctx *ctx, const double *u, double *energy)
	{
		if (!ctx || !u || !energy)
			return PFX_ERR_ARG;
		if (need_mesh(ctx) || ensure_staging(ctx))
			return ctx->err.empty() ? PFX_ERR_ARG : PFX_ERR_ARG;
		const int64_t n = ctx->n_dofs();
		PFX_HIP(hipMemcpyAsync(ctx->d_u, u, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
		pfx_status rc = pfx_energy_dev(ctx, ctx->d_u, ctx->d_vec + n);
		if (rc)
			return rc;
		PFX_HIP(hipMemcpyAsync(energy, ctx->d_vec + n, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
		rc = check_flag(ctx);
		if (rc)
		{
			*energy = NAN;
			return rc;
		}
		return PFX_OK;
	}

	pfx_status pfx_gradient(pfx_ctx *ctx, const double *u, double *g)
	{
		if (!ctx || !u || !g)
			return PFX_ERR_ARG;
		if (need_mesh(ctx) || ensure_staging(ctx))
			return PFX_ERR_ARG;
		const int64_t n = ctx->n_dofs();
		PFX_HIP(hipMemcpyAsync(ctx->d_u, u, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
		pfx_status rc = pfx_gradient_dev(ctx, ctx->d_u, ctx->d_vec);
		if (rc)
			return rc;
		PFX_HIP(hipMemcpyAsync(g, ctx->d_vec, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
		return check_flag(ctx);
	}

	pfx_status pfx_hessian(pfx_ctx *ctx, const double *u, int project_to_psd, double *values)
	{
		if (!ctx || !u)
			return PFX_ERR_ARG;
		if (need_mesh(ctx) || ensure_staging(ctx))
			return PFX_ERR_ARG;
		const int64_t n = ctx->n_dofs();
		PFX_HIP(hipMemcpyAsync(ctx->d_u, u, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
		pfx_status rc = hessian_dev_impl(ctx, ctx->d_u, project_to_psd != 0);
		if (rc)
			return rc;
		if (values && (rc = pfx_values_to_csr(ctx, values)))
			return rc;
		return check_flag(ctx);
	}

	pfx_status pfx_set_dirichlet(pfx_ctx *ctx, const int32_t *sorted_dofs, int64_t n)
	{
		if (!ctx)
			return PFX_ERR_ARG;
		if (need_pattern(ctx))
			return PFX_ERR_ARG;
		const int64_t nd = ctx->n_dofs();
		if (n < 0 || (n > 0 && !sorted_dofs))
			PFX_FAIL(PFX_ERR_ARG, "bad Dirichlet list");
		for (int64_t k = 0; k < n; ++k)
		{
			if (sorted_dofs[k] < 0 || sorted_dofs[k] >= nd)
				PFX_FAIL(PFX_ERR_ARG, "Dirichlet dof out of range");
			if (k > 0 && sorted_dofs[k] <= sorted_dofs[k - 1])
				PFX_FAIL(PFX_ERR_ARG, "Dirichlet dofs must be strictly ascending");
		}
		PFX_HIP(hipMemsetAsync(ctx->d_fixed, 0, (size_t)nd, ctx->stream));
		ctx->n_dirichlet = n;
		ctx->struct_gen++;
		if (n > 0)
		{
			if (pfx_alloc(ctx, &ctx->d_dirichlet, n))
				return PFX_ERR_HIP;
			PFX_HIP(hipMemcpyAsync(ctx->d_dirichlet, sorted_dofs, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
			hipLaunchKernelGGL(k_set_fixed, dim3(grid_for(n)), dim3(256), 0, ctx->stream, ctx->d_dirichlet, n, ctx->d_fixed);
		}
		PFX_HIP(hipStreamSynchronize(ctx->stream));
		return PFX_OK;
	}

	pfx_status pfx_get_reduced_csr(pfx_ctx *ctx, int64_t *n_reduced, int64_t *nnz, int64_t *row_ptr, int32_t *col_idx, double *values)
	{
		if (!ctx || !n_reduced || !nnz)
			return PFX_ERR_ARG;
		if (need_matrix(ctx))
			return PFX_ERR_ARG;
		return pfx_export_compressed(ctx, ctx->have_mask, true, n_reduced, nnz, row_ptr, col_idx, values);
	}

	pfx_status pfx_spmv_dev(pfx_ctx *ctx, const double *d_x, double *d_y)
	{
		if (!ctx || !d_x || !d_y)
			return PFX_ERR_ARG;
		if (need_matrix(ctx))
			return PFX_ERR_ARG;
		return pfx_launch_spmv(ctx, d_x, d_y);
	}

	pfx_status pfx_spmv(pfx_ctx *ctx, const double *x, double *y)
	{
		if (!ctx || !x || !y)
			return PFX_ERR_ARG;
		if (need_matrix(ctx) || ensure_staging(ctx))
			return PFX_ERR_ARG;
		const int64_t n = ctx->n_dofs();
		PFX_HIP(hipMemcpyAsync(ctx->d_u, x, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
		pfx_status rc = pfx_launch_spmv(ctx, ctx->d_u, ctx->d_vec);
		if (rc)
			return rc;
		PFX_HIP(hipMemcpyAsync(y, ctx->d_vec, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
		PFX_HIP(hipStreamSynchronize(ctx->stream));
		return PFX_OK;
	}

	pfx_status pfx_inertia_set_mass(pfx_ctx *ctx, double density, const double *density_e)
	{
		if (!ctx)
			return PFX_ERR_ARG;
		if (need_mesh(ctx))
			return PFX_ERR_ARG;
		if (ctx->bs < 2)
			PFX_FAIL(PFX_ERR_UNSUPPORTED, "the inertia term needs a vector-valued model (block size 2 or 3)");
		pfx_status rc = pfx_assemble_mass(ctx, density, density_e);
		if (rc)
			return rc;
		return pfx_capture_mass(ctx);
	}

	pfx_status pfx_inertia_enable(pfx_ctx *ctx, double acceleration_scaling, const double *d_x_tilde)
	{
		if (!ctx)
			return PFX_ERR_ARG;
		if (!(acceleration_scaling > 0.0) || !d_x_tilde)
		{
			ctx->inertia_weight = 0.0;
			ctx->d_x_tilde = nullptr;
			return PFX_OK;
		}
		if (need_mesh(ctx))
			return PFX_ERR_ARG;
		if (ctx->mass_count == 0)
			PFX_FAIL(PFX_ERR_ARG, "pfx_inertia_set_mass must be called first");
		ctx->inertia_weight = acceleration_scaling;
		ctx->d_x_tilde = d_x_tilde;
		return PFX_OK;
	}

	pfx_status pfx_inertia_enable_host(pfx_ctx *ctx, double acceleration_scaling, const double *x_tilde)
	{
		if (!ctx)
			return PFX_ERR_ARG;
		if (!(acceleration_scaling > 0.0) || !x_tilde)
			return pfx_inertia_enable(ctx, 0.0, nullptr);
		if (need_mesh(ctx))
			return PFX_ERR_ARG;
		const int64_t n = ctx->n_dofs();
		if (ctx->tstep_count < 2 * n)
		{
			if (pfx_alloc(ctx, &ctx->d_tstep, 2 * n))
				return PFX_ERR_HIP;
			ctx->tstep_count = 2 * n;
		}
		PFX_HIP(hipMemcpyAsync(ctx->d_tstep + n, x_tilde, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
		PFX_HIP(hipStreamSynchronize(ctx->stream));
		return pfx_inertia_enable(ctx, acceleration_scaling, ctx->d_tstep + n);
	}

	pfx_status pfx_al_enable(pfx_ctx *ctx, double weight, const double *d_masked_lumped_mass, const double *d_target)
	{
		if (!ctx)
			return PFX_ERR_ARG;
		if (!(weight > 0.0) || !d_masked_lumped_mass || !d_target)
		{
			ctx->al_weight = 0.0;
			ctx->d_al_mass = ctx->d_al_target = nullptr;
			return PFX_OK;
		}
		if (need_mesh(ctx))
			return PFX_ERR_ARG;
		if (ctx->bs < 2)
			PFX_FAIL(PFX_ERR_UNSUPPORTED, "the augmented-Lagrangian term needs a vector-valued model (block size 2 or 3)");
		ctx->al_weight = weight;
		ctx->d_al_mass = d_masked_lumped_mass;
		ctx->d_al_target = d_target;
		return PFX_OK;
	}

	pfx_status pfx_al_enable_host(pfx_ctx *ctx, double weight, const double *masked_lumped_mass, const double *target)
	{
		if (!ctx)
			return PFX_ERR_ARG;
		if (!(weight > 0.0) || !masked_lumped_mass || !target)
			return pfx_al_enable(ctx, 0.0, nullptr, nullptr);
		if (need_mesh(ctx))
			return PFX_ERR_ARG;
		const int64_t n = ctx->n_dofs();
		for (int64_t k = 0; k < n; ++k)
			if (!(masked_lumped_mass[k] >= 0.0))
				PFX_FAIL(PFX_ERR_ARG, "the masked lumped mass must be non-negative");
		if (ctx->al_count < 2 * n)
		{
			if (pfx_alloc(ctx, &ctx->d_al_store, 2 * n))
				return PFX_ERR_HIP;
			ctx->al_count = 2 * n;
		}
		PFX_HIP(hipMemcpyAsync(ctx->d_al_store, masked_lumped_mass, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
		PFX_HIP(hipMemcpyAsync(ctx->d_al_store + n, target, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
		PFX_HIP(hipStreamSynchronize(ctx->stream));
		return pfx_al_enable(ctx, weight, ctx->d_al_store, ctx->d_al_store + n);
	}

	pfx_status pfx_lumped_mass(pfx_ctx *ctx, double *lumped)
	{
		if (!ctx || !lumped)
			return PFX_ERR_ARG;
		if (need_mesh(ctx))
			return PFX_ERR_ARG;
		if (ctx->mass_count == 0 || ctx->bs < 2)
			PFX_FAIL(PFX_ERR_ARG, "pfx_inertia_set_mass must be called first");
		const int64_t n = ctx->n_dofs();
		pfx_tmp<double> d;
		PFX_HIP(d.alloc((size_t)n));
		pfx_status rc = pfx_lumped_mass_run(ctx, d.p);
		if (rc)
			return rc;
		PFX_HIP(hipMemcpyAsync(lumped, d.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
		PFX_HIP(hipStreamSynchronize(ctx->stream));
		return PFX_OK;
	}

	pfx_status pfx_mass_apply_dev(pfx_ctx *ctx, const double *d_v, double *d_y)
	{
		if (!ctx || !d_v || !d_y)
			return PFX_ERR_ARG;
		if (need_mesh(ctx))
			return PFX_ERR_ARG;
		if (ctx->mass_count == 0)
			PFX_FAIL(PFX_ERR_ARG, "pfx_inertia_set_mass must be called first");
		return pfx_mass_apply_run(ctx, d_v, d_y);
	}

	pfx_status pfx_hessian_add_mass_dev(pfx_ctx *ctx, double weight)
	{
		if (!ctx || !(weight >= 0.0))
			return PFX_ERR_ARG;
		if (need_mesh(ctx))
			return PFX_ERR_ARG;
		if (ctx->mass_count == 0)
			PFX_FAIL(PFX_ERR_ARG, "pfx_inertia_set_mass must be called first");
		return pfx_add_mass_run(ctx, weight);
	}

	pfx_status pfx_implicit_euler_step_dev(pfx_ctx *ctx, double dt, const double *d_rhs, double *d_x, double *d_v, double *d_a,
										   const pfx_newton_params *params, pfx_newton_info *info)
	{
		if (!ctx || !d_x || !d_v || !params || !info)
			return PFX_ERR_ARG;
		if (!(dt > 0.0))
			PFX_FAIL(PFX_ERR_ARG, "pfx_implicit_euler_step: dt must be positive");
		if (need_mesh(ctx))
			return PFX_ERR_ARG;
		if (ctx->mass_count == 0)
			PFX_FAIL(PFX_ERR_ARG, "pfx_inertia_set_mass must be called first");
		if (ctx->model == PFX_LAPLACIAN)
			PFX_FAIL(PFX_ERR_ARG, "pfx_implicit_euler_step needs an elastic model");
		if (ctx->n_owned != ctx->n_nodes)
			PFX_FAIL(PFX_ERR_UNSUPPORTED, "pfx_implicit_euler_step works on an unpartitioned mesh");
		PFX_HIP(hipSetDevice(ctx->device));
		return pfx_implicit_euler_run(ctx, dt, d_rhs, d_x, d_v, d_a, params, info);
	}

	pfx_status pfx_implicit_euler_step(pfx_ctx *ctx, double dt, const double *rhs, double *x, double *v, double *a,
									   const pfx_newton_params *params, pfx_newton_info *info)
	{
		if (!ctx || !x || !v || !params || !info)
			return PFX_ERR_ARG;
		if (need_mesh(ctx))
			return PFX_ERR_ARG;
		PFX_HIP(hipSetDevice(ctx->device));
		const int64_t n = ctx->n_dofs();
		pfx_tmp<double> d;
		PFX_HIP(d.alloc((size_t)(4 * n)));
		double *d_x = d.p, *d_v = d.p + n, *d_a = d.p + 2 * n, *d_rhs = rhs ? d.p + 3 * n : nullptr;
		hipStream_t st = ctx->stream;
		pfx_status rc = PFX_OK;
		if (hipMemcpyAsync(d_x, x, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st) != hipSuccess || hipMemcpyAsync(d_v, v, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st) != hipSuccess || (rhs && hipMemcpyAsync(d_rhs, rhs, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st) != hipSuccess))
			rc = PFX_ERR_HIP;
		if (!rc)
			rc = pfx_implicit_euler_step_dev(ctx, dt, d_rhs, d_x, d_v, d_a, params, info);
		if (!rc)
		{
			if (hipMemcpyAsync(x, d_x, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, st) != hipSuccess || hipMemcpyAsync(v, d_v, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, st) != hipSuccess || (a && hipMemcpyAsync(a, d_a, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, st) != hipSuccess) || hipStreamSynchronize(st) != hipSuccess)
			{
				ctx->err = "copy of the time-step result failed";
				rc = PFX_ERR_HIP;
			}
		}
		(void)hipStreamSynchronize(st);
		return rc;
	}

	pfx_status pfx_set_diag_shift(pfx_ctx *ctx, double shift)
	{
		if (!ctx || !(shift >= 0.0))
			return PFX_ERR_ARG;
		ctx->diag_shift = shift;
		return PFX_OK;
	}

	pfx_status pfx_take_numeric_flag(pfx_ctx *ctx, int *flag)
	{
		if (!ctx || !flag)
			return PFX_ERR_ARG;
		PFX_HIP(hipMemcpyAsync(flag, ctx->d_flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
		PFX_HIP(hipStreamSynchronize(ctx->stream));
		if (*flag)
			PFX_HIP(hipMemsetAsync(ctx->d_flag, 0, sizeof(int), ctx->stream));
		return PFX_OK;
	}

	void pfx_newton_default_params(pfx_newton_params *p)
	{
		if (!p)
			return;
		p->f_delta = 1e-10;
		p->grad_norm = 1e-8;
		p->first_grad_norm_tol = 1e-10;
		p->use_grad_norm_tol = 1e-4;
		p->max_iterations = 1000;
		p->use_grad_norm = 1;
		p->relative_gradient = 0;
		p->line_search = 1;
		p->cg_precond = PFX_PRECOND_JACOBI;
		p->cg_max_iter = 1000;
		p->cg_tol = 1e-12;
	}

	pfx_status pfx_newton_solve_dev(pfx_ctx *ctx, const double *d_rhs, double *d_x, const pfx_newton_params *params, pfx_newton_info *info)
	{
		if (!ctx || !d_x || !params || !info)
			return PFX_ERR_ARG;
		if (need_mesh(ctx))
			return PFX_ERR_ARG;
		if (ctx->model == PFX_LAPLACIAN)
			PFX_FAIL(PFX_ERR_UNSUPPORTED, "pfx_newton_solve needs an elastic model");
		if (ctx->n_owned != ctx->n_nodes)
			PFX_FAIL(PFX_ERR_UNSUPPORTED, "pfx_newton_solve runs on an unpartitioned mesh");
		PFX_HIP(hipSetDevice(ctx->device));
		return pfx_newton_run(ctx, d_rhs, d_x, params, info);
	}

	pfx_status pfx_newton_solve(pfx_ctx *ctx, const double *rhs, double *x, const pfx_newton_params *params, pfx_newton_info *info)
	{
		if (!ctx || !x || !params || !info)
			return PFX_ERR_ARG;
		if (need_mesh(ctx))
			return PFX_ERR_ARG;
		const int64_t n = ctx->n_dofs();
		pfx_tmp<double> d;
		PFX_HIP(d.alloc((size_t)(2 * n)));
		PFX_HIP(hipMemcpyAsync(d.p, x, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
		if (rhs)
			PFX_HIP(hipMemcpyAsync(d.p + n, rhs, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
		const pfx_status rc = pfx_newton_solve_dev(ctx, rhs ? d.p + n : nullptr, d.p, params, info);
		(void)hipMemcpyAsync(x, d.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream);
		(void)hipStreamSynchronize(ctx->stream);
		return rc;
	}

	pfx_status pfx_cg_dev(pfx_ctx *ctx, const double *d_b, double *d_x, int precond, int max_iter, double tol, int *iters, double *rel_res)
	{
		if (!ctx || !d_b || !d_x || !iters || !rel_res)
			return PFX_ERR_ARG;
		if (need_matrix(ctx))
			return PFX_ERR_ARG;
		if (precond != PFX_PRECOND_IDENTITY && precond != PFX_PRECOND_JACOBI && precond != PFX_PRECOND_AMG)
			PFX_FAIL(PFX_ERR_ARG, "unknown preconditioner");
		return pfx_cg_run(ctx, d_b, d_x, precond, max_iter, tol, iters, rel_res);
	}

	pfx_status pfx_cg(pfx_ctx *ctx, const double *b, double *x, int precond, int max_iter, double tol, int *iters, double *rel_res)
	{
		if (!ctx || !b || !x || !iters || !rel_res)
			return PFX_ERR_ARG;
		if (need_matrix(ctx) || ensure_staging(ctx))
			return PFX_ERR_ARG;
		const int64_t n = ctx->n_dofs();
		PFX_HIP(hipMemcpyAsync(ctx->d_u, b, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
		PFX_HIP(hipMemcpyAsync(ctx->d_vec, x, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
		pfx_status rc = pfx_cg_dev(ctx, ctx->d_u, ctx->d_vec, precond, max_iter, tol, iters, rel_res);
		if (rc)
			return rc;
		PFX_HIP(hipMemcpyAsync(x, ctx->d_vec, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
		PFX_HIP(hipStreamSynchronize(ctx->stream));
		return PFX_OK;
	}

	pfx_status pfx_dirichlet_solve(pfx_ctx *ctx, const double *b, double *x, int precond, int max_iter, double tol, int *iters, double *rel_res)
	{
		if (!ctx || !b || !x || !iters || !rel_res)
			return PFX_ERR_ARG;
		if (need_matrix(ctx) || ensure_staging(ctx))
			return PFX_ERR_ARG;
		const int64_t n = ctx->n_dofs();
		hipStream_t st = ctx->stream;
		pfx_tmp<double> tb, tg;
		PFX_HIP(tb.alloc((size_t)n));
		PFX_HIP(tg.alloc((size_t)n));
		double *d_b = tb.p, *d_g = tg.p;
		PFX_HIP(hipMemcpyAsync(d_b, b, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st));
		// lift: g holds the prescribed values; A g with the UNMASKED operator moves them to the right-hand side
		hipLaunchKernelGGL(k_dirichlet_lift, dim3(grid_for(n)), dim3(256), 0, st, d_b, ctx->d_fixed, n, d_g);
		const int64_t nd_saved = ctx->n_dirichlet;
		ctx->n_dirichlet = 0;
		pfx_status rc = pfx_launch_spmv(ctx, d_g, ctx->d_vec);
		ctx->n_dirichlet = nd_saved;
		if (!rc)
		{
			hipLaunchKernelGGL(k_dirichlet_rhs, dim3(grid_for(n)), dim3(256), 0, st, d_b, ctx->d_vec, ctx->d_fixed, n, ctx->d_u);
			PFX_HIP(hipMemsetAsync(ctx->d_vec, 0, sizeof(double) * (size_t)n, st));
			rc = pfx_cg_run(ctx, ctx->d_u, ctx->d_vec, precond, max_iter, tol, iters, rel_res);
		}
		if (!rc)
		{
			hipLaunchKernelGGL(k_dirichlet_fill, dim3(grid_for(n)), dim3(256), 0, st, d_b, ctx->d_fixed, n, ctx->d_vec);
			PFX_HIP(hipMemcpyAsync(x, ctx->d_vec, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, st));
			PFX_HIP(hipStreamSynchronize(st));
		}
		(void)hipStreamSynchronize(st); // the temporaries are released on return
		return rc;
	}

	pfx_status pfx_set_owned(pfx_ctx *ctx, int64_t n_owned_nodes, const uint8_t *elem_owned)
	{
		if (!ctx)
			return PFX_ERR_ARG;
		if (ctx->kind < 0)
			PFX_FAIL(PFX_ERR_ARG, "pfx_set_mesh must be called first");
		if (n_owned_nodes < 0 || n_owned_nodes > ctx->n_nodes)
			PFX_FAIL(PFX_ERR_ARG, "n_owned_nodes out of range");
		ctx->n_owned = n_owned_nodes;
		ctx->struct_gen++;
		free_dev(ctx, ctx->d_elem_owned);
		if (elem_owned)
		{
			if (pfx_alloc(ctx, &ctx->d_elem_owned, ctx->n_el))
				return PFX_ERR_HIP;
			PFX_HIP(hipMemcpyAsync(ctx->d_elem_owned, elem_owned, (size_t)ctx->n_el, hipMemcpyHostToDevice, ctx->stream));
			PFX_HIP(hipStreamSynchronize(ctx->stream));
		}
		return PFX_OK;
	}

	pfx_status pfx_cgs_set_workspace(pfx_ctx *ctx, double *d_work, int64_t n_doubles)
	{
		if (!ctx)
			return PFX_ERR_ARG;
		if (!d_work)
		{
			ctx->d_work_ext = nullptr;
			return PFX_OK;
		}
		if (need_pattern(ctx))
			return PFX_ERR_ARG;
		if (n_doubles < 4 * ctx->n_dofs())
			PFX_FAIL(PFX_ERR_ARG, "pfx_cgs_set_workspace: the workspace must hold 4 * n_dofs doubles");
		ctx->d_work_ext = d_work;
		return PFX_OK;
	}

#define PFX_NEED(cond)         \
	if (!ctx || !(cond))       \
		return PFX_ERR_ARG;    \
	if (need_matrix(ctx))      \
		return PFX_ERR_ARG;

	pfx_status pfx_cgs_prepare(pfx_ctx *ctx, double *d_x, int precond)
	{
		PFX_NEED(d_x);
		return pfx_cgs_prepare_impl(ctx, d_x, precond);
	}
	pfx_status pfx_cgs_residual(pfx_ctx *ctx, const double *d_b, const double *d_x, double *d_scal)
	{
		PFX_NEED(d_b && d_x);
		return pfx_cgs_residual_impl(ctx, d_b, d_x, d_scal);
	}
	pfx_status pfx_cgs_start(pfx_ctx *ctx, double *d_x, double tol, double *d_scal)
	{
		PFX_NEED(d_x);
		return pfx_cgs_start_impl(ctx, d_x, tol, d_scal);
	}
	pfx_status pfx_cgs_start2(pfx_ctx *ctx, const double *d_scal)
	{
		PFX_NEED(true);
		return pfx_cgs_start2_impl(ctx, d_scal);
	}
	pfx_status pfx_cgs_spmv(pfx_ctx *ctx, double *d_scal)
	{
		PFX_NEED(true);
		return pfx_cgs_spmv_impl(ctx, d_scal);
	}
	pfx_status pfx_cgs_spmv_interior(pfx_ctx *ctx)
	{
		PFX_NEED(true);
		return pfx_cgs_spmv_phase_impl(ctx, PFX_SPMV_INTERIOR, nullptr);
	}
	pfx_status pfx_cgs_spmv_boundary(pfx_ctx *ctx, double *d_scal)
	{
		PFX_NEED(true);
		return pfx_cgs_spmv_phase_impl(ctx, PFX_SPMV_BOUNDARY, d_scal);
	}
	pfx_status pfx_cgs_fetch_product(pfx_ctx *ctx)
	{
		PFX_NEED(true);
		return pfx_cgs_fetch_product_impl(ctx);
	}
	pfx_status pfx_cgs_update(pfx_ctx *ctx, int it, double *d_x, double *d_scal)
	{
		PFX_NEED(d_x);
		return pfx_cgs_update_impl(ctx, it, d_x, d_scal);
	}
	pfx_status pfx_cgs_direction(pfx_ctx *ctx, int it, const double *d_scal)
	{
		PFX_NEED(true);
		return pfx_cgs_direction_impl(ctx, it, d_scal);
	}
	pfx_status pfx_cgs_status(pfx_ctx *ctx, int *done, int *iters, double *rr, double *bb)
	{
		PFX_NEED(done && iters && rr && bb);
		return pfx_cgs_status_impl(ctx, done, iters, rr, bb);
	}

	pfx_status pfx_get_info(pfx_ctx *ctx, pfx_info *info)
	{
		if (!ctx || !info)
			return PFX_ERR_ARG;
		info->n_el = ctx->n_el;
		info->n_nodes = ctx->n_nodes;
		info->n_dofs = ctx->n_dofs();
		info->n_blocks = ctx->n_blocks;
		info->nnz = ctx->nnz;
		info->n_loc = ctx->n_loc;
		info->n_pts = ctx->n_pts;
		info->dim = ctx->dim;
		info->bytes_device = ctx->bytes_device;
		return PFX_OK;
	}

	pfx_status pfx_timing_enable(pfx_ctx *ctx, int on)
	{
		if (!ctx)
			return PFX_ERR_ARG;
		resolve_timers(ctx);
		ctx->timing = on != 0;
		ctx->timing_stride = on > 1 ? on : 1;
		return PFX_OK;
	}

	pfx_status pfx_timing_get(pfx_ctx *ctx, const char *family, double *avg_ms, int64_t *launches)
	{
		if (!ctx || !family || !avg_ms || !launches)
			return PFX_ERR_ARG;
		resolve_timers(ctx);
		auto it = ctx->timers.find(family);
		if (it == ctx->timers.end() || it->second.launches == 0)
		{
			*avg_ms = 0;
			*launches = 0;
			return PFX_OK;
		}
		*avg_ms = it->second.total_ms / (double)it->second.launches;
		*launches = it->second.launches;
		return PFX_OK;
	}

	pfx_status pfx_timing_reset(pfx_ctx *ctx)
	{
		if (!ctx)
			return PFX_ERR_ARG;
		resolve_timers(ctx);
		ctx->timers.clear();
		ctx->timing_seen.clear();
		return PFX_OK;
	}

} // extern "C"
