// pfx_extmat.hip -- a solver context built from a matrix the CALLER hands in (compressed column storage, the layout of
// the reference's StiffnessMatrix = Eigen::SparseMatrix<double, ColMajor, int>, src/polyfem/utils/Types.hpp:16-20).
//
// This is polysolve::LinearSolver::analyzePattern / factorize for a matrix that did not come out of this library's own
// assembly: the reduced sum of all forms' Hessians that SparseNewtonDescentSolver hands to its linear solver
// (src/polyfem/solver/SparseNewtonDescentSolver.tpp:128-155, matrix built at src/polyfem/solver/FullNLProblem.cpp:115-135 and
// src/polyfem/solver/NLProblem.cpp:92-99) or the stiffness matrix of a linear problem
// (src/polyfem/state/StateSolveLinear.cpp:27-29,51-54).
//
// analyze (once per pattern):  CSC -> CSR on the device (stable radix sort of the entries by row, so the columns of a
//   row stay ascending; no symmetry of the pattern is assumed), then either
//     block size 3: the union of the column NODES of the three rows of every node is the node-block pattern the solver's
//       block-SELL format wants; entries the caller's pattern does not hold are explicit zeros (a reduced matrix whose
//       constrained dofs are whole nodes, or a stiffness matrix with a few exact zeros dropped, fills > 95 %), chosen when
//       n is a multiple of 3 and the blocks hold at most 25 % more entries than the caller's matrix; or
//     block size 1: the scalar pattern as it is.
//   A map from every stored entry to its position in the caller's value array is kept.
// factorize (once per matrix):  the caller's values are gathered through that map into the node-block value array; a
//   non-finite entry or a non-positive diagonal entry is reported as PFX_ERR_NUMERIC (the reference's caller catches the
//   failure of factorize and regularises, SparseNewtonDescentSolver.tpp:135-150).
// Afterwards pfx_cg(_dev), pfx_spmv(_dev), pfx_set_dirichlet, pfx_dirichlet_solve, pfx_set_diag_shift and the AMG entry
// points work on this matrix exactly as on an assembled one: same SELL copy, same CG / PCG kernels.
#include "pfx_internal.h"

#include <rocprim/rocprim.hpp>

#include <cmath>

namespace
{
	constexpr uint32_t PAD = 0xFFFFFFFFu;

	__global__ void k_ext_iota(uint32_t *__restrict__ v, int64_t n)
	{
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
			v[k] = (uint32_t)k;
	}

	// colof[k] = column of entry k of the compressed-column arrays: the j with outer[j] <= k < outer[j+1]
	__global__ void k_ext_colof(const int64_t *__restrict__ outer, int64_t n, int64_t nnz, int32_t *__restrict__ colof)
	{
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x)
		{
			int64_t lo = 0, hi = n; // largest j with outer[j] <= k
			while (hi - lo > 1)
			{
				const int64_t mid = (lo + hi) >> 1;
				if (outer[mid] <= k)
					lo = mid;
				else
					hi = mid;
			}
			colof[k] = (int32_t)lo;
		}
	}

	// row_ptr[r] = first position of the row-sorted entry list whose row is >= r
	__global__ void k_ext_rowptr(const uint32_t *__restrict__ rows_sorted, int64_t nnz, int64_t n, int64_t *__restrict__ row_ptr)
	{
		for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= n; r += (int64_t)gridDim.x * blockDim.x)
		{
			int64_t lo = 0, hi = nnz;
			while (lo < hi)
			{
				const int64_t mid = (lo + hi) >> 1;
				if ((int64_t)rows_sorted[mid] < r)
					lo = mid + 1;
				else
					hi = mid;
			}
			row_ptr[r] = lo;
		}
	}

	// columns in row order; flag[0] = 1 if an entry appears twice
	__global__ void k_ext_csr_cols(const uint32_t *__restrict__ rows_sorted, const uint32_t *__restrict__ perm, const int32_t *__restrict__ colof, int64_t nnz,
								   int32_t *__restrict__ ccol, int *__restrict__ flag)
	{
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x)
		{
			const int32_t c = colof[perm[k]];
			ccol[k] = c;
			if (k > 0 && rows_sorted[k] == rows_sorted[k - 1] && colof[perm[k - 1]] == c)
				flag[0] = 1;
		}
	}

	// Node g owns rows BS g .. BS g + BS - 1: BS-way merge of their (ascending) column lists by column node.
	// FILL = false: cnt[g] = number of distinct column nodes.  FILL = true: node_col and the value map of the blocks.
	template <bool FILL, int BS>
	__global__ void k_ext_blocks(const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ ccol, const uint32_t *__restrict__ perm, int64_t n_nodes,
								 int64_t *__restrict__ cnt, const int64_t *__restrict__ node_ptr, int32_t *__restrict__ node_col, uint32_t *__restrict__ map)
	{
		for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_nodes; g += (int64_t)gridDim.x * blockDim.x)
		{
			int64_t p[BS], e[BS];
			for (int m = 0; m < BS; ++m)
				p[m] = row_ptr[BS * g + m], e[m] = row_ptr[BS * g + m + 1];
			int64_t b = FILL ? node_ptr[g] : 0;
			for (;;)
			{
				int32_t c = INT32_MAX;
				for (int m = 0; m < BS; ++m)
					if (p[m] < e[m])
					{
						const int32_t cm = ccol[p[m]] / BS;
						c = cm < c ? cm : c;
					}
				if (c == INT32_MAX)
					break;
				for (int m = 0; m < BS; ++m)
					while (p[m] < e[m] && ccol[p[m]] / BS == c)
					{
						if (FILL)
							map[b * (BS * BS) + m * BS + (ccol[p[m]] - BS * c)] = perm[p[m]];
						++p[m];
					}
				if (FILL)
					node_col[b] = c;
				++b;
			}
			if (!FILL)
				cnt[g] = b;
		}
	}

	__global__ void k_ext_copy_ptr(const int64_t *__restrict__ src, int64_t n, int64_t *__restrict__ dst)
	{
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
			dst[k] = src[k];
	}

	// values[s] = src[map[s]] (0 where the caller's pattern has no entry); flag |= 1 on a non-finite value
	__global__ void k_ext_gather(const uint32_t *__restrict__ map, const double *__restrict__ src, int64_t count, double *__restrict__ values, int *__restrict__ flag)
	{
		int bad = 0;
		for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < count; s += (int64_t)gridDim.x * blockDim.x)
		{
			const uint32_t k = map[s];
			const double v = k == PAD ? 0.0 : src[k];
			values[s] = v;
			bad |= !isfinite(v);
		}
		if (bad)
			flag[0] = 1;
	}

	// flag |= 2 if a diagonal entry is missing, not finite or not positive; row[0] = smallest such row
	template <int BS>
	__global__ void k_ext_diag_check(const int64_t *__restrict__ node_ptr, const int32_t *__restrict__ node_col, const double *__restrict__ values, int64_t n_nodes,
									 int *__restrict__ flag, int *__restrict__ row)
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
				const double d = s >= 0 ? values[(b0 + s) * (BS * BS) + m * BS + m] : 0.0;
				if (!(d > 0.0) || !isfinite(d))
				{
					atomicOr(flag, 2);
					atomicMin(row, (int)(g * BS + m));
				}
			}
		}
	}

	pfx_status analyze_impl(pfx_ctx *ctx, int64_t n, const int64_t *outer64, const int32_t *outer32, const int32_t *inner, int block_size);
} // namespace

// releases what pfx_solver_analyze added to the context (called with the rest of the pattern)
void pfx_ext_release(pfx_ctx *ctx)
{
	free_dev(ctx, ctx->d_ext_map);
	free_dev(ctx, ctx->d_ext_stage);
	free_dev(ctx, ctx->d_ext_pos);
	ctx->ext = false;
	ctx->ext_nnz = 0;
	ctx->ext_factorized = false;
}

void pfx_release_pattern(pfx_ctx *ctx); // pfx_api.hip

namespace
{
	pfx_status analyze_impl(pfx_ctx *ctx, int64_t n, const int64_t *outer64, const int32_t *outer32, const int32_t *inner, int block_size)
	{
		if (ctx->kind >= 0)
			PFX_FAIL(PFX_ERR_ARG, "pfx_solver_analyze: this context holds a mesh; a caller's matrix takes a context of its own (pfx_create)");
		if (n <= 0 || (!outer64 && !outer32) || !inner)
			PFX_FAIL(PFX_ERR_ARG, "pfx_solver_analyze: empty matrix or null pointer");
		if (block_size < 0 || block_size > 3)
			PFX_FAIL(PFX_ERR_ARG, "pfx_solver_analyze: block_size is 0 (choose between 1 and 3), 1, 2 or 3");
		if (block_size > 1 && n % block_size != 0)
			PFX_FAIL(PFX_ERR_ARG, "pfx_solver_analyze: block size 2 / 3 needs a multiple of 2 / 3 rows");
		if (n >= ((int64_t)1 << 31) - 1)
			PFX_FAIL(PFX_ERR_SIZE, "pfx_solver_analyze: the matrix dimension must fit a 32-bit index");
		// the kernels index device arrays with these values: validate them on the host
		std::vector<int64_t> outer((size_t)n + 1);
		for (int64_t j = 0; j <= n; ++j)
			outer[(size_t)j] = outer64 ? outer64[j] : (int64_t)outer32[j];
		if (outer[0] != 0)
			PFX_FAIL(PFX_ERR_ARG, "pfx_solver_analyze: outer[0] must be 0 (compressed storage)");
		for (int64_t j = 0; j < n; ++j)
			if (outer[(size_t)j + 1] < outer[(size_t)j])
				PFX_FAIL(PFX_ERR_ARG, "pfx_solver_analyze: outer must not decrease");
		const int64_t nnz = outer[(size_t)n];
		if (nnz <= 0)
			PFX_FAIL(PFX_ERR_ARG, "pfx_solver_analyze: the matrix has no entries");
		if (nnz >= (int64_t)PAD)
			PFX_FAIL(PFX_ERR_SIZE, "pfx_solver_analyze: at most 2^32 - 2 entries");
		for (int64_t k = 0; k < nnz; ++k)
			if (inner[k] < 0 || inner[k] >= n)
				PFX_FAIL(PFX_ERR_ARG, "pfx_solver_analyze: row index out of range");
		PFX_HIP(hipSetDevice(ctx->device));
		pfx_release_pattern(ctx);
		hipStream_t st = ctx->stream;

		pfx_tmp<int64_t> d_outer, d_rowptr, d_cnt;
		pfx_tmp<int32_t> d_colof, d_ccol;
		pfx_tmp<uint32_t> ka, kb, va, vb;
		pfx_tmp<char> tmp;
		pfx_tmp<int> d_flag;
		PFX_HIP(d_outer.alloc((size_t)n + 1));
		PFX_HIP(ka.alloc((size_t)nnz));
		PFX_HIP(kb.alloc((size_t)nnz));
		PFX_HIP(va.alloc((size_t)nnz));
		PFX_HIP(vb.alloc((size_t)nnz));
		PFX_HIP(d_colof.alloc((size_t)nnz));
		PFX_HIP(d_flag.alloc(2));
		PFX_HIP(hipMemsetAsync(d_flag.p, 0, 2 * sizeof(int), st));
		PFX_HIP(hipMemcpyAsync(d_outer.p, outer.data(), sizeof(int64_t) * ((size_t)n + 1), hipMemcpyHostToDevice, st));
		PFX_HIP(hipMemcpyAsync(ka.p, inner, sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice, st)); // rows as sort keys
		hipLaunchKernelGGL(k_ext_iota, dim3(grid_for(nnz)), dim3(256), 0, st, va.p, nnz);
		hipLaunchKernelGGL(k_ext_colof, dim3(grid_for(nnz)), dim3(256), 0, st, d_outer.p, n, nnz, d_colof.p);
		// entries in row order; the sort is stable and the entries come column by column: columns ascend inside a row
		int bits = 1;
		while (((int64_t)1 << bits) < n)
			++bits;
		size_t tb = 0;
		PFX_HIP(rocprim::radix_sort_pairs(nullptr, tb, ka.p, kb.p, va.p, vb.p, (size_t)nnz, 0, bits, st));
		PFX_HIP(tmp.alloc(tb));
		PFX_HIP(rocprim::radix_sort_pairs(tmp.p, tb, ka.p, kb.p, va.p, vb.p, (size_t)nnz, 0, bits, st));
		PFX_HIP(d_rowptr.alloc((size_t)n + 1));
		PFX_HIP(d_ccol.alloc((size_t)nnz));
		hipLaunchKernelGGL(k_ext_rowptr, dim3(grid_for(n + 1)), dim3(256), 0, st, kb.p, nnz, n, d_rowptr.p);
		hipLaunchKernelGGL(k_ext_csr_cols, dim3(grid_for(nnz)), dim3(256), 0, st, kb.p, vb.p, d_colof.p, nnz, d_ccol.p, d_flag.p);
		int h_flag = 0;
		PFX_HIP(hipMemcpyAsync(&h_flag, d_flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
		PFX_HIP(hipStreamSynchronize(st));
		PFX_HIP(hipGetLastError());
		if (h_flag)
			PFX_FAIL(PFX_ERR_ARG, "pfx_solver_analyze: an entry appears twice (the matrix must be in compressed form)");
		ka.reset(), va.reset(), kb.reset(), d_colof.reset(), d_outer.reset();
		// vb = perm: position in the caller's arrays of every entry of the row-ordered list

		int bs = 1;
		int64_t n_blocks = nnz;
		if (block_size == 3 || block_size == 2 || (block_size == 0 && n % 3 == 0))
		{
			const int B = block_size == 2 ? 2 : 3, B2 = B * B;
			const int64_t n_nodes = n / B;
			PFX_HIP(d_cnt.alloc((size_t)n_nodes + 1));
			PFX_HIP(hipMemsetAsync(d_cnt.p, 0, sizeof(int64_t) * ((size_t)n_nodes + 1), st));
			if (B == 2)
				hipLaunchKernelGGL((k_ext_blocks<false, 2>), dim3(grid_for(n_nodes)), dim3(256), 0, st, d_rowptr.p, d_ccol.p, vb.p, n_nodes, d_cnt.p, (const int64_t *)nullptr,
								   (int32_t *)nullptr, (uint32_t *)nullptr);
			else
				hipLaunchKernelGGL((k_ext_blocks<false, 3>), dim3(grid_for(n_nodes)), dim3(256), 0, st, d_rowptr.p, d_ccol.p, vb.p, n_nodes, d_cnt.p, (const int64_t *)nullptr,
								   (int32_t *)nullptr, (uint32_t *)nullptr);
			if (pfx_alloc(ctx, &ctx->d_node_ptr, n_nodes + 1))
				return PFX_ERR_HIP;
			tb = 0;
			PFX_HIP(rocprim::exclusive_scan(nullptr, tb, d_cnt.p, ctx->d_node_ptr, (int64_t)0, (size_t)(n_nodes + 1), rocprim::plus<int64_t>(), st));
			PFX_HIP(tmp.alloc(tb));
			PFX_HIP(rocprim::exclusive_scan(tmp.p, tb, d_cnt.p, ctx->d_node_ptr, (int64_t)0, (size_t)(n_nodes + 1), rocprim::plus<int64_t>(), st));
			int64_t nb = 0;
			PFX_HIP(hipMemcpyAsync(&nb, ctx->d_node_ptr + n_nodes, sizeof(int64_t), hipMemcpyDeviceToHost, st));
			PFX_HIP(hipStreamSynchronize(st));
			if (block_size >= 2 || (double)nb * 9.0 <= 1.25 * (double)nnz)
			{
				bs = B;
				n_blocks = nb;
				if (nb * B2 >= (int64_t)PAD * 2) // keeps the 32-bit block indices of the solver copy valid
					PFX_FAIL(PFX_ERR_SIZE, "pfx_solver_analyze: too many blocks");
				if (pfx_alloc(ctx, &ctx->d_node_col, nb) || pfx_alloc(ctx, &ctx->d_ext_map, nb * B2))
					return PFX_ERR_HIP;
				PFX_HIP(hipMemsetAsync(ctx->d_ext_map, 0xFF, sizeof(uint32_t) * (size_t)(nb * B2), st));
				if (B == 2)
					hipLaunchKernelGGL((k_ext_blocks<true, 2>), dim3(grid_for(n_nodes)), dim3(256), 0, st, d_rowptr.p, d_ccol.p, vb.p, n_nodes, (int64_t *)nullptr, ctx->d_node_ptr,
									   ctx->d_node_col, ctx->d_ext_map);
				else
					hipLaunchKernelGGL((k_ext_blocks<true, 3>), dim3(grid_for(n_nodes)), dim3(256), 0, st, d_rowptr.p, d_ccol.p, vb.p, n_nodes, (int64_t *)nullptr, ctx->d_node_ptr,
									   ctx->d_node_col, ctx->d_ext_map);
			}
		}
		if (bs == 1)
		{
			if (pfx_alloc(ctx, &ctx->d_node_ptr, n + 1) || pfx_alloc(ctx, &ctx->d_node_col, nnz) || pfx_alloc(ctx, &ctx->d_ext_map, nnz))
				return PFX_ERR_HIP;
			hipLaunchKernelGGL(k_ext_copy_ptr, dim3(grid_for(n + 1)), dim3(256), 0, st, d_rowptr.p, n + 1, ctx->d_node_ptr);
			PFX_HIP(hipMemcpyAsync(ctx->d_node_col, d_ccol.p, sizeof(int32_t) * (size_t)nnz, hipMemcpyDeviceToDevice, st));
			PFX_HIP(hipMemcpyAsync(ctx->d_ext_map, vb.p, sizeof(uint32_t) * (size_t)nnz, hipMemcpyDeviceToDevice, st));
		}
		ctx->bs = bs;
		ctx->dim = 3;
		ctx->n_nodes = ctx->n_owned = n / bs;
		ctx->n_blocks = n_blocks;
		ctx->nnz = n_blocks * bs * bs;
		ctx->ext = true;
		ctx->ext_nnz = nnz;
		ctx->ext_factorized = false;
		if (pfx_alloc(ctx, &ctx->d_values, ctx->nnz + 2) || pfx_alloc(ctx, &ctx->d_fixed, n))
			return PFX_ERR_HIP;
		PFX_HIP(hipMemsetAsync(ctx->d_values, 0, sizeof(double) * (size_t)(ctx->nnz + 2), st));
		PFX_HIP(hipMemsetAsync(ctx->d_fixed, 0, (size_t)n, st));
		PFX_HIP(hipStreamSynchronize(st));
		PFX_HIP(hipGetLastError());
		ctx->struct_gen++;
		return PFX_OK;
	}

	pfx_status factorize_impl(pfx_ctx *ctx, const double *d_src)
	{
		hipStream_t st = ctx->stream;
		pfx_tmp<int> d_flag;
		PFX_HIP(d_flag.alloc(2));
		const int init[2] = {0, INT32_MAX};
		PFX_HIP(hipMemcpyAsync(d_flag.p, init, sizeof(init), hipMemcpyHostToDevice, st));
		hipLaunchKernelGGL(k_ext_gather, dim3(grid_for(ctx->nnz)), dim3(256), 0, st, ctx->d_ext_map, d_src, ctx->nnz, ctx->d_values, d_flag.p);
		if (ctx->bs == 2)
			hipLaunchKernelGGL((k_ext_diag_check<2>), dim3(grid_for(ctx->n_nodes)), dim3(256), 0, st, ctx->d_node_ptr, ctx->d_node_col, ctx->d_values, ctx->n_nodes, d_flag.p, d_flag.p + 1);
		else if (ctx->bs == 3)
			hipLaunchKernelGGL((k_ext_diag_check<3>), dim3(grid_for(ctx->n_nodes)), dim3(256), 0, st, ctx->d_node_ptr, ctx->d_node_col, ctx->d_values, ctx->n_nodes, d_flag.p, d_flag.p + 1);
		else
			hipLaunchKernelGGL((k_ext_diag_check<1>), dim3(grid_for(ctx->n_nodes)), dim3(256), 0, st, ctx->d_node_ptr, ctx->d_node_col, ctx->d_values, ctx->n_nodes, d_flag.p, d_flag.p + 1);
		int h[2] = {0, 0};
		PFX_HIP(hipMemcpyAsync(h, d_flag.p, sizeof(h), hipMemcpyDeviceToHost, st));
		PFX_HIP(hipStreamSynchronize(st));
		PFX_HIP(hipGetLastError());
		ctx->have_mask = false;
		ctx->values_gen++;
		ctx->ext_factorized = false;
		if (h[0] & 1)
			PFX_FAIL(PFX_ERR_NUMERIC, "pfx_solver_factorize: the matrix holds a non-finite value");
		if (h[0] & 2)
			PFX_FAIL(PFX_ERR_NUMERIC, "pfx_solver_factorize: diagonal entry of row " + std::to_string(h[1]) + " is missing, not finite or not positive (the matrix is not positive definite)");
		ctx->ext_factorized = true;
		return PFX_OK;
	}
} // namespace

extern "C"
{
	pfx_status pfx_solver_analyze(pfx_ctx *ctx, int64_t n, const int32_t *outer, const int32_t *inner, int block_size)
	{
		if (!ctx)
			return PFX_ERR_ARG;
		return analyze_impl(ctx, n, nullptr, outer, inner, block_size);
	}

	pfx_status pfx_solver_analyze64(pfx_ctx *ctx, int64_t n, const int64_t *outer, const int32_t *inner, int block_size)
	{
		if (!ctx)
			return PFX_ERR_ARG;
		return analyze_impl(ctx, n, outer, nullptr, inner, block_size);
	}

	pfx_status pfx_solver_factorize(pfx_ctx *ctx, const double *values)
	{
		if (!ctx || !values)
			return PFX_ERR_ARG;
		if (!ctx->ext)
			PFX_FAIL(PFX_ERR_ARG, "pfx_solver_factorize: pfx_solver_analyze must be called first");
		PFX_HIP(hipSetDevice(ctx->device));
		if (!ctx->d_ext_stage && pfx_alloc(ctx, &ctx->d_ext_stage, ctx->ext_nnz))
			return PFX_ERR_HIP;
		PFX_HIP(hipMemcpyAsync(ctx->d_ext_stage, values, sizeof(double) * (size_t)ctx->ext_nnz, hipMemcpyHostToDevice, ctx->stream));
		return factorize_impl(ctx, ctx->d_ext_stage);
	}

	pfx_status pfx_solver_factorize_dev(pfx_ctx *ctx, const double *d_values)
	{
		if (!ctx || !d_values)
			return PFX_ERR_ARG;
		if (!ctx->ext)
			PFX_FAIL(PFX_ERR_ARG, "pfx_solver_factorize_dev: pfx_solver_analyze must be called first");
		PFX_HIP(hipSetDevice(ctx->device));
		return factorize_impl(ctx, d_values);
	}

	pfx_status pfx_solver_set_node_positions(pfx_ctx *ctx, const double *xyz, int64_t n_nodes)
	{
		if (!ctx)
			return PFX_ERR_ARG;
		if (!ctx->ext)
			PFX_FAIL(PFX_ERR_ARG, "pfx_solver_set_node_positions: pfx_solver_analyze must be called first");
		PFX_HIP(hipSetDevice(ctx->device));
		ctx->struct_gen++; // the aggregation depends on the positions
		if (!xyz)
		{
			free_dev(ctx, ctx->d_ext_pos);
			return PFX_OK;
		}
		if (n_nodes != ctx->n_nodes)
			PFX_FAIL(PFX_ERR_ARG, "pfx_solver_set_node_positions: one position per node (rows / block size) expected");
		for (int64_t k = 0; k < 3 * n_nodes; ++k)
			if (!std::isfinite(xyz[k]))
				PFX_FAIL(PFX_ERR_ARG, "pfx_solver_set_node_positions: non-finite coordinate");
		if (pfx_alloc(ctx, &ctx->d_ext_pos, 3 * n_nodes))
			return PFX_ERR_HIP;
		PFX_HIP(hipMemcpyAsync(ctx->d_ext_pos, xyz, sizeof(double) * (size_t)(3 * n_nodes), hipMemcpyHostToDevice, ctx->stream));
		PFX_HIP(hipStreamSynchronize(ctx->stream));
		return PFX_OK;
	}

	pfx_status pfx_solver_get_layout(pfx_ctx *ctx, int *block_size, int64_t *n_blocks, int64_t *stored_entries)
	{
		if (!ctx)
			return PFX_ERR_ARG;
		if (!ctx->ext)
			PFX_FAIL(PFX_ERR_ARG, "pfx_solver_get_layout: pfx_solver_analyze must be called first");
		if (block_size)
			*block_size = ctx->bs;
		if (n_blocks)
			*n_blocks = ctx->n_blocks;
		if (stored_entries)
			*stored_entries = ctx->nnz;
		return PFX_OK;
	}

	pfx_status pfx_matrix_generation(pfx_ctx *ctx, int64_t *generation)
	{
		if (!ctx || !generation)
			return PFX_ERR_ARG;
		*generation = ctx->values_gen;
		return PFX_OK;
	}
}
