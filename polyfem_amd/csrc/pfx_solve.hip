// pfx_solve.hip -- sparse matrix-vector product and the conjugate-gradient solver.
//
// Replaces the polysolve::LinearSolver "Eigen::ConjugateGradient" path called from
// src/polyfem/solver/SparseNewtonDescentSolver.tpp:128-155 and src/polyfem/state/StateSolveLinear.cpp:14-68.
// polysolve and Eigen are not part of the reference tree; the iteration below restates
// Eigen 3.4 IterativeLinearSolvers/ConjugateGradient.h (same operation order, same stopping rule
// ||r||^2 < max(tol^2 ||b||^2, DBL_MIN), same iteration count convention) with its DiagonalPreconditioner.
//
// Matrix storage is the assembler's node-block CSR: the value array is a scalar CSR's, but column indices exist
// only per node block (4 B per bs*bs values), so an SpMV streams ~8.4 B per entry instead of 12.
// Constrained (Dirichlet) dofs are handled by masking rows instead of rebuilding a reduced matrix
// (utils::full_to_reduced_matrix, src/polyfem/utils/MatrixUtils.cpp:408-469): all solver vectors are kept zero
// on constrained dofs, for which the masked operator is the identity; the iterates on the free dofs are those of
// CG on the reduced system.
//
// One CG iteration = 3 launches, no host synchronisation:
//   k_spmv_dot    tmp = A p, per-block partial sums of p.tmp
//   k_cg_update   alpha from the partials; x += alpha p; r -= alpha tmp; partials of r.r and r.M^-1 r
//   k_cg_direction reduces those partials, tests convergence (device flag), p = M^-1 r + beta p
// Partial sums are reduced in a fixed order by every block, so all blocks see the same scalars and results are
// bitwise reproducible.
#include "pfx_internal.h"

#include <cfloat>
#include <cmath>

#define PFX_SUM_STAGE 4096 // offset in d_partials (6152 doubles) of the wide partial sum's 128 chunk sums + ticket
#define PFX_NPART 1024 // number of partial sums (= grid size of the reducing kernels; 1024 measured 1 % faster per CG iteration than 2048, 512 the same)

// device-resident solver state
struct pfx_cg_state
{
	double abs_new[2]; // r.z ping-pong by iteration parity
	double rr;         // last ||r||^2
	double threshold;
	double rhs_norm2;
	int done;  // 1 once converged
	int iters; // Eigen's iteration count at exit
};

namespace
{
	__device__ __forceinline__ double wave_sum(double v)
	{
		for (int off = 32; off > 0; off >>= 1)
			v += __shfl_down(v, off, 64);
		return v;
	}

	// block-wide sum of one value per thread (256 threads), result valid in thread 0
	__device__ __forceinline__ double block_sum(double v, double *sh)
	{
		v = wave_sum(v);
		const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
		if (lane == 0)
			sh[w] = v;
		__syncthreads();
		double r = 0;
		if (threadIdx.x == 0)
			r = (sh[0] + sh[1]) + (sh[2] + sh[3]);
		__syncthreads();
		return r;
	}

	// every thread of the block returns the same fixed-order sum of part[0..count)
	__device__ __forceinline__ double reduce_partials(const double *__restrict__ part, int count, double *sh)
	{
		double acc = 0;
		for (int k = threadIdx.x; k < count; k += 256)
			acc += part[k];
		sh[threadIdx.x] = acc;
		__syncthreads();
		for (int s = 128; s > 0; s >>= 1)
		{
			if ((int)threadIdx.x < s)
				sh[threadIdx.x] += sh[threadIdx.x + s];
			__syncthreads();
		}
		const double r = sh[0];
		__syncthreads();
		return r;
	}

	// invdiag = 1/diag (1 where diag == 0 or the dof is constrained)  -- Eigen DiagonalPreconditioner::factorize
	template <int BS>
	__global__ void k_invdiag(const int64_t *__restrict__ node_ptr, const int32_t *__restrict__ node_col, const double *__restrict__ values,
							  const uint8_t *__restrict__ fixed, int64_t n_nodes, int jacobi, double shift, double *__restrict__ invdiag)
	{
		for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_nodes; g += (int64_t)gridDim.x * blockDim.x)
		{
			const int64_t b0 = node_ptr[g];
			const int deg = (int)(node_ptr[g + 1] - b0);
			int lo = 0, hi = deg - 1, s = -1;
			while (lo <= hi)
			{
				const int mid = (lo + hi) >> 1;
				const int32_t c = node_col[b0 + mid];
				if (c == g)
				{
					s = mid;
					break;
				}
				if (c < g)
					lo = mid + 1;
				else
					hi = mid - 1;
			}
			for (int m = 0; m < BS; ++m)
			{
				double d = 1.0;
				const int64_t r = g * BS + m;
				if (jacobi && s >= 0 && !(fixed && fixed[r]))
				{
					const double a = values[(b0 + s) * (BS * BS) + m * BS + m] + shift;
					if (a != 0.0)
						d = 1.0 / a;
				}
				invdiag[r] = d;
			}
		}
	}

	// Scalars of the iteration come either from the per-block partial sums of the producing kernel (single GPU:
	// every block reduces them in the same fixed order) or, when several GPUs cooperate, from `gscal`, which the
	// caller has all-reduced over the ranks after k_finish_partials wrote the local sums there.
	__device__ __forceinline__ double get_scalar(const double *__restrict__ part, int count, const double *__restrict__ gscal, int idx, double *sh)
	{
		if (gscal)
			return gscal[idx];
		return reduce_partials(part, count, sh);
	}

	// gscal[idx0 + q] = fixed-order sum of partial array q (one block)
	__global__ __launch_bounds__(256) void k_finish_partials(const double *__restrict__ part, int count, double *__restrict__ out)
	{
		__shared__ double sh[256];
		const double v = reduce_partials(part, count, sh);
		if (threadIdx.x == 0)
			*out = v;
	}

	// *out = fixed-order sum of a long partial array (one per SpMV slice): 1024 threads, eight independent loads in
	// flight per thread (a plain strided loop of dependent adds is latency bound: 21 us for 21 k partials)
	__global__ __launch_bounds__(1024) void k_sum_partials(const double *__restrict__ part, int count, double *__restrict__ out)
	{
		__shared__ double sh[16];
		double acc = 0;
		for (int base = threadIdx.x; base < count; base += 8 * 1024)
		{
			double v[8];
#pragma unroll
			for (int u = 0; u < 8; ++u)
			{
				const int idx = base + u * 1024;
				v[u] = idx < count ? part[idx] : 0.0;
			}
#pragma unroll
			for (int u = 0; u < 8; ++u)
				acc += v[u];
		}
		acc = wave_sum(acc);
		if ((threadIdx.x & 63) == 0)
			sh[threadIdx.x >> 6] = acc;
		__syncthreads();
		if (threadIdx.x == 0)
		{
			double r = 0;
			for (int w = 0; w < 16; ++w)
				r += sh[w];
			*out = r;
		}
	}

	// The same for hundreds of thousands of partials (10 M elements: one block needs 38 us, 0.7 % of a CG iteration): SUM_BLOCKS
	// blocks each sum one contiguous chunk, the block that finishes last (ticket counter) adds the chunk sums in chunk
	// order.  The result does not depend on which block that is: fixed summation order, one launch.  stage: SUM_BLOCKS
	// doubles, ticket: zero before the first launch (the last block leaves it zero again).
	constexpr int SUM_BLOCKS = 128;
	__global__ __launch_bounds__(256) void k_sum_partials_wide(const double *__restrict__ part, int count, double *__restrict__ stage, unsigned int *__restrict__ ticket,
															   double *__restrict__ out)
	{
		__shared__ double sh[4];
		__shared__ bool last;
		const int chunk = (count + SUM_BLOCKS - 1) / SUM_BLOCKS;
		const int lo = blockIdx.x * chunk, hi = min(lo + chunk, count);
		double acc = 0;
		for (int base = lo + threadIdx.x; base < hi; base += 4 * 256)
		{
			double v[4];
#pragma unroll
			for (int u = 0; u < 4; ++u)
			{
				const int idx = base + u * 256;
				v[u] = idx < hi ? part[idx] : 0.0;
			}
#pragma unroll
			for (int u = 0; u < 4; ++u)
				acc += v[u];
		}
		const double s = block_sum(acc, sh);
		if (threadIdx.x == 0)
		{
			stage[blockIdx.x] = s;
			__threadfence(); // the chunk sum is visible before the ticket is taken
			last = atomicAdd(ticket, 1u) == SUM_BLOCKS - 1;
		}
		__syncthreads();
		if (!last)
			return;
		__threadfence();
		const volatile double *vs = stage; // written by other blocks during this launch: read past the caches of this CU
		double t = threadIdx.x < SUM_BLOCKS ? vs[threadIdx.x] : 0.0;
		// 128 values: lanes 0..63 of waves 0 and 1; fixed tree inside each wave, then wave 0 + wave 1
		t = wave_sum(t);
		if ((threadIdx.x & 63) == 0)
			sh[threadIdx.x >> 6] = t;
		__syncthreads();
		if (threadIdx.x == 0)
		{
			*out = sh[0] + sh[1];
			*ticket = 0u;
		}
	}

	// r = b - tmp (tmp = A x), with b forced to zero on constrained dofs; partials: [0] b.b  [1] r.r
	__global__ __launch_bounds__(256) void k_cg_init1(const double *__restrict__ b, const double *__restrict__ tmp, const uint8_t *__restrict__ fixed,
													   int64_t n, double *__restrict__ r, double *__restrict__ part)
	{
		__shared__ double sh[4];
		double bb = 0, rr = 0;
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
		{
			const bool fx = fixed && fixed[k];
			const double bk = fx ? 0.0 : b[k];
			const double rk = fx ? 0.0 : bk - tmp[k];
			r[k] = rk;
			bb += bk * bk;
			rr += rk * rk;
		}
		const double s0 = block_sum(bb, sh);
		const double s1 = block_sum(rr, sh);
		if (threadIdx.x == 0)
		{
			part[blockIdx.x] = s0;
			part[PFX_NPART + blockIdx.x] = s1;
		}
	}

	// threshold / early exit from b.b and r.r; p = M^-1 r; partial [2] of r.p
	__global__ __launch_bounds__(256) void k_cg_init2(const double *__restrict__ r, const double *__restrict__ invdiag, int64_t n, double tol,
													   double *__restrict__ p, double *__restrict__ part, const double *__restrict__ gscal,
													   pfx_cg_state *__restrict__ state, double *__restrict__ x)
	{
		__shared__ double sh[256];
		const double bb = get_scalar(part, PFX_NPART, gscal, 0, sh);
		const double rr = get_scalar(part + PFX_NPART, PFX_NPART, gscal, 1, sh);
		double thr = tol * tol * bb;
		if (thr < DBL_MIN)
			thr = DBL_MIN;
		const bool zero_rhs = (bb == 0.0);
		const bool conv = zero_rhs || (rr < thr);
		if (blockIdx.x == 0 && threadIdx.x == 0)
		{
			state->rhs_norm2 = bb;
			state->threshold = thr;
			state->rr = zero_rhs ? 0.0 : rr;
			state->done = conv ? 1 : 0;
			state->iters = 0;
		}
		double rp = 0;
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
		{
			if (zero_rhs)
				x[k] = 0.0; // Eigen: rhsNorm2 == 0 -> x.setZero()
			const double pk = invdiag[k] * r[k];
			p[k] = pk;
			rp += r[k] * pk;
		}
		const double s = block_sum(rp, sh);
		if (threadIdx.x == 0)
			part[2 * PFX_NPART + blockIdx.x] = s;
	}

	// abs_new[0] = r.p
	__global__ __launch_bounds__(256) void k_cg_init3(const double *__restrict__ part, const double *__restrict__ gscal, pfx_cg_state *__restrict__ state)
	{
		__shared__ double sh[256];
		const double rp = get_scalar(part + 2 * PFX_NPART, PFX_NPART, gscal, 2, sh);
		if (threadIdx.x == 0)
			state->abs_new[0] = rp;
	}

	// alpha = abs_new / p.tmp ; x += alpha p ; r -= alpha tmp ; partials [1], [2] of r.r and r.(M^-1 r)
	// BS > 0: tmp holds A p in the slice order of the solver's matrix copy (see pfx_sell_spmv), pos the rows' positions in it
	template <int BS>
	__global__ __launch_bounds__(256) void k_cg_update(const double *__restrict__ p, const double *__restrict__ tmp, const int32_t *__restrict__ pos,
														const double *__restrict__ invdiag, int64_t n, int it, double *__restrict__ x, double *__restrict__ r,
														double *__restrict__ part, const double *__restrict__ pap, const pfx_cg_state *__restrict__ state)
	{
		__shared__ double sh[256];
		if (state->done)
			return;
		const double pAp = *pap; // p.Ap, summed over the SpMV blocks (and over the ranks) before this kernel
		const double alpha = state->abs_new[it & 1] / pAp;
		double rr = 0, rz = 0;
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
		{
			x[k] += alpha * p[k];
			double tk;
			if (BS > 0)
			{
				const int64_t g = k / BS;
				const int64_t q = pos[g];
				tk = tmp[((q >> 6) * BS + (k - g * BS)) * 64 + (q & 63)];
			}
			else
				tk = tmp[k];
			const double rk = r[k] - alpha * tk;
			r[k] = rk;
			rr += rk * rk;
			rz += rk * (invdiag[k] * rk);
		}
		const double s0 = block_sum(rr, sh);
		const double s1 = block_sum(rz, sh);
		if (threadIdx.x == 0)
		{
			part[PFX_NPART + blockIdx.x] = s0;
			part[2 * PFX_NPART + blockIdx.x] = s1;
		}
	}

	// convergence test, beta, p = M^-1 r + beta p
	__global__ __launch_bounds__(256) void k_cg_direction(const double *__restrict__ r, const double *__restrict__ invdiag, int64_t n, int it,
														   double *__restrict__ p, const double *__restrict__ part, const double *__restrict__ gscal,
														   pfx_cg_state *__restrict__ state)
	{
		__shared__ double sh[256];
		if (state->done)
			return;
		const double rr = get_scalar(part + PFX_NPART, PFX_NPART, gscal, 1, sh);
		const double rz = get_scalar(part + 2 * PFX_NPART, PFX_NPART, gscal, 2, sh);
		const bool conv = rr < state->threshold;
		const double beta = rz / state->abs_new[it & 1];
		if (!conv)
			for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
				p[k] = invdiag[k] * r[k] + beta * p[k];
		if (blockIdx.x == 0 && threadIdx.x == 0)
		{
			state->rr = rr;
			if (conv)
			{
				state->done = 1;
				state->iters = it; // Eigen breaks before i++
			}
			else
			{
				state->abs_new[(it + 1) & 1] = rz;
				state->iters = it + 1;
			}
		}
	}

	__global__ void k_mask_inplace(double *__restrict__ v, const uint8_t *__restrict__ fixed, int64_t n)
	{
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
			if (fixed[k])
				v[k] = 0.0;
	}

} // namespace

// Jacobi diagonal of the (masked, shifted) operator the solver works with: 1 / a_ii, 1 where a_ii == 0 or the dof is constrained
pfx_status pfx_launch_invdiag(pfx_ctx *ctx, int jacobi, double *d_invdiag)
{
	const uint8_t *fixed = ctx->n_dirichlet > 0 ? ctx->d_fixed : nullptr;
	if (ctx->bs == 1)
		hipLaunchKernelGGL((k_invdiag<1>), dim3(grid_for(ctx->n_owned)), dim3(256), 0, ctx->stream, ctx->d_node_ptr, ctx->d_node_col, ctx->d_values, fixed, ctx->n_owned, jacobi, ctx->diag_shift, d_invdiag);
	else if (ctx->bs == 2)
		hipLaunchKernelGGL((k_invdiag<2>), dim3(grid_for(ctx->n_owned)), dim3(256), 0, ctx->stream, ctx->d_node_ptr, ctx->d_node_col, ctx->d_values, fixed, ctx->n_owned, jacobi, ctx->diag_shift, d_invdiag);
	else
		hipLaunchKernelGGL((k_invdiag<3>), dim3(grid_for(ctx->n_owned)), dim3(256), 0, ctx->stream, ctx->d_node_ptr, ctx->d_node_col, ctx->d_values, fixed, ctx->n_owned, jacobi, ctx->diag_shift, d_invdiag);
	PFX_HIP(hipGetLastError());
	return PFX_OK;
}

void pfx_launch_sum_partials(pfx_ctx *ctx, const double *part, int count, double *out)
{
	if (count < 32768)
	{
		hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, ctx->stream, part, count, out);
		return;
	}
	// stage and ticket live behind the scalars of d_partials (zeroed at pfx_create)
	double *stage = ctx->d_partials + PFX_SUM_STAGE;
	hipLaunchKernelGGL(k_sum_partials_wide, dim3(SUM_BLOCKS), dim3(256), 0, ctx->stream, part, count, stage, (unsigned int *)(stage + SUM_BLOCKS), out);
}

pfx_status pfx_launch_spmv(pfx_ctx *ctx, const double *d_x, double *d_y)
{
	pfx_status rc = pfx_sell_refresh(ctx);
	if (rc)
		return rc;
	if ((rc = pfx_sell_spmv(ctx, d_x, d_y, nullptr, nullptr)))
		return rc;
	PFX_HIP(hipGetLastError());
	return PFX_OK;
}

namespace
{
	struct cg_vecs
	{
		double *r, *p, *tmp, *invdiag;
	};

	pfx_status cg_workspace(pfx_ctx *ctx, cg_vecs *w)
	{
		const int64_t n = ctx->n_dofs();
		double *base = ctx->d_work_ext;
		if (!base)
		{
			if (!ctx->d_work)
				if (pfx_alloc(ctx, &ctx->d_work, 4 * n))
					return PFX_ERR_HIP;
			base = ctx->d_work;
		}
		if (!ctx->d_cgstate)
			if (pfx_alloc(ctx, (pfx_cg_state **)&ctx->d_cgstate, 1))
				return PFX_ERR_HIP;
		w->r = base;
		w->p = base + n;
		w->tmp = base + 2 * n;
		w->invdiag = base + 3 * n;
		return PFX_OK;
	}

	// local sums -> gscal[first..first+count) for the caller's all-reduce (multi-GPU only)
	void finish(pfx_ctx *ctx, int first, int count, double *gscal)
	{
		if (!gscal)
			return;
		for (int q = first; q < first + count; ++q)
			hipLaunchKernelGGL(k_finish_partials, dim3(1), dim3(256), 0, ctx->stream, ctx->d_partials + q * PFX_NPART, PFX_NPART, gscal + q);
	}
} // namespace

// ---- step functions: one CG iteration is spmv -> update -> direction; with several ranks the caller all-reduces
// ---- gscal between the steps and refreshes the halo entries of p before every spmv --------------------------------

pfx_status pfx_cgs_prepare_impl(pfx_ctx *ctx, double *d_x, int precond)
{
	cg_vecs w;
	if (cg_workspace(ctx, &w))
		return PFX_ERR_HIP;
	hipStream_t st = ctx->stream;
	const int64_t n = ctx->n_owned * ctx->bs;
	const uint8_t *fixed = ctx->n_dirichlet > 0 ? ctx->d_fixed : nullptr;
	{
		pfx_status rc = pfx_sell_refresh(ctx); // "factorize": the solver's copy of the matrix
		if (rc)
			return rc;
		if (ctx->part0_count < pfx_sell_grid(ctx))
		{
			if (pfx_alloc(ctx, &ctx->d_part0, pfx_sell_grid(ctx)))
				return PFX_ERR_HIP;
			ctx->part0_count = pfx_sell_grid(ctx);
		}
	}
	PFX_HIP(hipMemsetAsync(ctx->d_cgstate, 0, sizeof(pfx_cg_state), st));
	if (fixed) // initial guess with constrained dofs zeroed (solveWithGuess on the reduced system)
		hipLaunchKernelGGL(k_mask_inplace, dim3(grid_for(n)), dim3(256), 0, st, d_x, fixed, n);
	const int jac = precond == PFX_PRECOND_JACOBI ? 1 : 0;
	if (ctx->bs == 1)
		hipLaunchKernelGGL((k_invdiag<1>), dim3(grid_for(ctx->n_owned)), dim3(256), 0, st, ctx->d_node_ptr, ctx->d_node_col, ctx->d_values, fixed, ctx->n_owned, jac, ctx->diag_shift, w.invdiag);
	else if (ctx->bs == 2)
		hipLaunchKernelGGL((k_invdiag<2>), dim3(grid_for(ctx->n_owned)), dim3(256), 0, st, ctx->d_node_ptr, ctx->d_node_col, ctx->d_values, fixed, ctx->n_owned, jac, ctx->diag_shift, w.invdiag);
	else
		hipLaunchKernelGGL((k_invdiag<3>), dim3(grid_for(ctx->n_owned)), dim3(256), 0, st, ctx->d_node_ptr, ctx->d_node_col, ctx->d_values, fixed, ctx->n_owned, jac, ctx->diag_shift, w.invdiag);
	PFX_HIP(hipGetLastError());
	return PFX_OK;
}

pfx_status pfx_cgs_residual_impl(pfx_ctx *ctx, const double *d_b, const double *d_x, double *gscal)
{
	cg_vecs w;
	if (cg_workspace(ctx, &w))
		return PFX_ERR_HIP;
	const int64_t n = ctx->n_owned * ctx->bs;
	const uint8_t *fixed = ctx->n_dirichlet > 0 ? ctx->d_fixed : nullptr;
	{
		pfx_status rc = pfx_sell_spmv(ctx, d_x, w.tmp, nullptr, nullptr);
		if (rc)
			return rc;
	}
	hipLaunchKernelGGL(k_cg_init1, dim3(PFX_NPART), dim3(256), 0, ctx->stream, d_b, w.tmp, fixed, n, w.r, ctx->d_partials);
	finish(ctx, 0, 2, gscal);
	PFX_HIP(hipGetLastError());
	return PFX_OK;
}

pfx_status pfx_cgs_start_impl(pfx_ctx *ctx, double *d_x, double tol, double *gscal)
{
	cg_vecs w;
	if (cg_workspace(ctx, &w))
		return PFX_ERR_HIP;
	const int64_t n = ctx->n_owned * ctx->bs;
	hipLaunchKernelGGL(k_cg_init2, dim3(PFX_NPART), dim3(256), 0, ctx->stream, w.r, w.invdiag, n, tol, w.p, ctx->d_partials, gscal, (pfx_cg_state *)ctx->d_cgstate, d_x);
	finish(ctx, 2, 1, gscal);
	PFX_HIP(hipGetLastError());
	return PFX_OK;
}

pfx_status pfx_cgs_start2_impl(pfx_ctx *ctx, const double *gscal)
{
	hipLaunchKernelGGL(k_cg_init3, dim3(1), dim3(256), 0, ctx->stream, ctx->d_partials, gscal, (pfx_cg_state *)ctx->d_cgstate);
	PFX_HIP(hipGetLastError());
	return PFX_OK;
}

pfx_status pfx_cgs_spmv_impl(pfx_ctx *ctx, double *gscal)
{
	return pfx_cgs_spmv_phase_impl(ctx, PFX_SPMV_ALL, gscal);
}

// PFX_SPMV_INTERIOR: the part of tmp = A p that needs no halo entry of p (enqueue it while the exchange is in flight);
// PFX_SPMV_BOUNDARY: the rest, then the sum of the per-slice partials of p.Ap -> gscal[0]
pfx_status pfx_cgs_spmv_phase_impl(pfx_ctx *ctx, int phase, double *gscal)
{
	cg_vecs w;
	if (cg_workspace(ctx, &w))
		return PFX_ERR_HIP;
	// a whole product may stay in the solver copy's slice order: only the update kernel reads it
	bool sliced = false;
	pfx_status rc = pfx_sell_spmv(ctx, w.p, w.tmp, ctx->d_part0, &((pfx_cg_state *)ctx->d_cgstate)->done, phase, &sliced);
	ctx->cg_tmp_sliced = sliced;
	if (rc)
		return rc;
	// one partial per slice: summed once here rather than by every block of the update kernel
	if (phase != PFX_SPMV_INTERIOR)
		pfx_launch_sum_partials(ctx, ctx->d_part0, pfx_sell_grid(ctx), gscal ? gscal : ctx->d_partials + 3 * PFX_NPART);
	return PFX_OK;
}

// tmp = A p in row order in the workspace, for a caller that reads it there
pfx_status pfx_cgs_fetch_product_impl(pfx_ctx *ctx)
{
	if (!ctx->cg_tmp_sliced)
		return PFX_OK;
	cg_vecs w;
	if (cg_workspace(ctx, &w))
		return PFX_ERR_HIP;
	pfx_status rc = pfx_sell_unpermute(ctx, w.tmp);
	if (!rc)
		ctx->cg_tmp_sliced = false;
	return rc;
}

pfx_status pfx_cgs_update_impl(pfx_ctx *ctx, int it, double *d_x, double *gscal)
{
	cg_vecs w;
	if (cg_workspace(ctx, &w))
		return PFX_ERR_HIP;
	const int64_t n = ctx->n_owned * ctx->bs;
	{
		pfx_scoped_timer tm(ctx, "cg_update");
		const double *pap = gscal ? gscal : ctx->d_partials + 3 * PFX_NPART;
		if (ctx->cg_tmp_sliced && ctx->bs == 3)
			hipLaunchKernelGGL((k_cg_update<3>), dim3(PFX_NPART), dim3(256), 0, ctx->stream, w.p, ctx->d_sell_ys, ctx->d_sell_pos, w.invdiag, n, it, d_x, w.r, ctx->d_partials, pap, (pfx_cg_state *)ctx->d_cgstate);
		else if (ctx->cg_tmp_sliced && ctx->bs == 1)
			hipLaunchKernelGGL((k_cg_update<1>), dim3(PFX_NPART), dim3(256), 0, ctx->stream, w.p, ctx->d_sell_ys, ctx->d_sell_pos, w.invdiag, n, it, d_x, w.r, ctx->d_partials, pap, (pfx_cg_state *)ctx->d_cgstate);
		else if (ctx->cg_tmp_sliced && ctx->bs == 2)
			hipLaunchKernelGGL((k_cg_update<2>), dim3(PFX_NPART), dim3(256), 0, ctx->stream, w.p, ctx->d_sell_ys, ctx->d_sell_pos, w.invdiag, n, it, d_x, w.r, ctx->d_partials, pap, (pfx_cg_state *)ctx->d_cgstate);
		else
			hipLaunchKernelGGL((k_cg_update<0>), dim3(PFX_NPART), dim3(256), 0, ctx->stream, w.p, w.tmp, (const int32_t *)nullptr, w.invdiag, n, it, d_x, w.r, ctx->d_partials, pap, (pfx_cg_state *)ctx->d_cgstate);
	}
	finish(ctx, 1, 2, gscal);
	return PFX_OK;
}

pfx_status pfx_cgs_direction_impl(pfx_ctx *ctx, int it, const double *gscal)
{
	cg_vecs w;
	if (cg_workspace(ctx, &w))
		return PFX_ERR_HIP;
	const int64_t n = ctx->n_owned * ctx->bs;
	pfx_scoped_timer tm(ctx, "cg_dir");
	hipLaunchKernelGGL(k_cg_direction, dim3(PFX_NPART), dim3(256), 0, ctx->stream, w.r, w.invdiag, n, it, w.p, ctx->d_partials, gscal, (pfx_cg_state *)ctx->d_cgstate);
	return PFX_OK;
}

pfx_status pfx_cg_direction_vector(pfx_ctx *ctx, double **p)
{
	cg_vecs w;
	if (cg_workspace(ctx, &w))
		return PFX_ERR_HIP;
	*p = w.p;
	return PFX_OK;
}

pfx_status pfx_cgs_status_impl(pfx_ctx *ctx, int *done, int *iters, double *rr, double *bb)
{
	pfx_cg_state h;
	if (!ctx->d_cgstate)
		PFX_FAIL(PFX_ERR_ARG, "no solve in progress");
	PFX_HIP(hipMemcpyAsync(&h, ctx->d_cgstate, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
	PFX_HIP(hipStreamSynchronize(ctx->stream));
	*done = h.done;
	*iters = h.iters;
	*rr = h.rr;
	*bb = h.rhs_norm2;
	return PFX_OK;
}

pfx_status pfx_cg_run(pfx_ctx *ctx, const double *d_b, double *d_x, int precond, int max_iter, double tol, int *iters, double *rel_res)
{
	pfx_status rc;
	if (precond == PFX_PRECOND_AMG)
		return pfx_pcg_amg_run(ctx, d_b, d_x, max_iter, tol, iters, rel_res);
	if ((rc = pfx_cgs_prepare_impl(ctx, d_x, precond)) || (rc = pfx_cgs_residual_impl(ctx, d_b, d_x, nullptr)) || (rc = pfx_cgs_start_impl(ctx, d_x, tol, nullptr)) || (rc = pfx_cgs_start2_impl(ctx, nullptr)))
		return rc;
	// the kernels return at once after convergence (device flag), so a late check costs a few empty launches only
	const int check_every = 64;
	int it = 0, done = 0, its = 0;
	double rr = 0, bb = 0;
	while (it < max_iter && !done)
	{
		const int stop = (it + check_every < max_iter) ? it + check_every : max_iter;
		for (; it < stop; ++it)
			if ((rc = pfx_cgs_spmv_impl(ctx, nullptr)) || (rc = pfx_cgs_update_impl(ctx, it, d_x, nullptr)) || (rc = pfx_cgs_direction_impl(ctx, it, nullptr)))
				return rc;
		if ((rc = pfx_cgs_status_impl(ctx, &done, &its, &rr, &bb)))
			return rc;
		// a long solve: worth finding a good placement for the value array (between two iterations the slice-ordered
		// product buffer is free)
		if (!done && !ctx->sell_placed && it >= PFX_PLACE_AFTER && max_iter - it >= PFX_PLACE_WORTH && (rc = pfx_sell_place(ctx)))
			return rc;
	}
	if (max_iter <= 0 && (rc = pfx_cgs_status_impl(ctx, &done, &its, &rr, &bb)))
		return rc;
	PFX_HIP(hipGetLastError());
	*iters = its;
	*rel_res = (bb == 0.0) ? 0.0 : std::sqrt(rr / bb);
	return PFX_OK;
}
