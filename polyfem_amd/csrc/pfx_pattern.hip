// pfx_pattern.hip -- device-side construction of the global sparsity pattern and of the element -> entry maps.
//
// Replaces what utils::SpareMatrixCache learns during the reference's first Hessian assembly
// (src/polyfem/utils/MatrixUtils.cpp:108-116 records every add_value call, :171-241 freezes the CSC arrays and
// the per-element value-index list `second_cache_`).  Here the same information is produced without assembling
// anything:
//   1. every (element e, local row i, local col j) emits the key (node_i, node_j); a stable radix sort of the
//      keys with the payload e*n_loc^2 + i*n_loc + j groups all contributions of one node pair, in ascending
//      element order (a fixed summation order, so assembly is deterministic);
//   2. run heads of the sorted keys are the node blocks: node_ptr / node_col (ascending columns per row, as in
//      Eigen's compressed storage), and blk_ptr delimits each block's contribution list;
//   3. only the blocks on and above the diagonal (column node >= row node) keep a contribution list: the reference
//      evaluates the pairs j <= i only and mirrors them (Assembler.cpp:179-184), and so does the assembly here -- the
//      blocks below the diagonal receive the transpose of their mirror image;
//   4. the same grouping is done per node for (element, local index) incidences, used by the gradient.
// The scalar CSR of a block size bs > 1 is implied: row node*bs+m owns, per neighbour slot s, the bs columns
// node_col[s]*bs+n; it is materialised only when the caller asks for it (pfx_export_pattern).
#include <cmath>
#include "pfx_internal.h"

#include <rocprim/rocprim.hpp>
#include <map>
#include <algorithm>

namespace
{
	__global__ void k_pair_keys(const int32_t *__restrict__ conn, int64_t n_items, int n_loc, int64_t n_nodes,
								uint64_t *__restrict__ keys, uint32_t *__restrict__ vals)
	{
		const int nl2 = n_loc * n_loc;
		for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n_items; idx += (int64_t)gridDim.x * blockDim.x)
		{
			const int64_t e = idx / nl2;
			const int ij = (int)(idx - e * nl2);
			const int i = ij / n_loc, j = ij - i * n_loc;
			const uint64_t gi = (uint64_t)conn[e * n_loc + i], gj = (uint64_t)conn[e * n_loc + j];
			keys[idx] = gi * (uint64_t)n_nodes + gj;
			vals[idx] = (uint32_t)idx;
		}
	}

	__global__ void k_node_keys(const int32_t *__restrict__ conn, int64_t n_items, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals)
	{
		for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n_items; idx += (int64_t)gridDim.x * blockDim.x)
		{
			keys[idx] = (uint32_t)conn[idx];
			vals[idx] = (uint32_t)idx;
		}
	}

	struct head_flag
	{
		const uint64_t *keys;
		__host__ __device__ int64_t operator()(int64_t k) const { return (k == 0 || keys[k] != keys[k - 1]) ? 1 : 0; }
	};

	__global__ void k_fill_blocks(const uint64_t *__restrict__ keys, const int64_t *__restrict__ pos, int64_t n_items, int64_t n_nodes,
								  int32_t *__restrict__ node_col, int64_t *__restrict__ deg)
	{
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_items; k += (int64_t)gridDim.x * blockDim.x)
		{
			const uint64_t key = keys[k];
			if (k == 0 || key != keys[k - 1])
			{
				const int64_t b = pos[k];
				const uint64_t gi = key / (uint64_t)n_nodes;
				node_col[b] = (int32_t)(key - gi * (uint64_t)n_nodes);
				atomicAdd((unsigned long long *)&deg[gi], 1ull);
			}
		}
	}

	// 1 for a contribution to a block on or above the diagonal; index n_items (one past the end) counts as 0 so that an
	// exclusive scan over n_items + 1 flags ends with the total
	struct upper_flag
	{
		const uint64_t *keys;
		int64_t n_items;
		uint64_t n_nodes;
		bool heads_only;
		__host__ __device__ int64_t operator()(int64_t k) const
		{
			if (k >= n_items)
				return 0;
			const uint64_t key = keys[k];
			const uint64_t gi = key / n_nodes, gj = key - gi * n_nodes;
			if (gi > gj)
				return 0;
			return (!heads_only || k == 0 || key != keys[k - 1]) ? 1 : 0;
		}
	};

	// keeps the contributions of the upper blocks: their keys and sources in list order, the start of every upper block's
	// run, the distance between its index among all blocks and among the upper ones, and the upper blocks per row
	__global__ void k_compact_upper(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ src, const int64_t *__restrict__ upos,
									const int64_t *__restrict__ ubpos, const int64_t *__restrict__ pos, int64_t n_items, int64_t n_nodes,
									uint64_t *__restrict__ ukeys, uint32_t *__restrict__ usrc, int64_t *__restrict__ ublk_ptr,
									uint32_t *__restrict__ ublk_off, int64_t *__restrict__ udeg)
	{
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_items; k += (int64_t)gridDim.x * blockDim.x)
		{
			const uint64_t key = keys[k];
			const uint64_t gi = key / (uint64_t)n_nodes, gj = key - gi * (uint64_t)n_nodes;
			if (gi > gj)
				continue;
			const int64_t u = upos[k];
			ukeys[u] = key;
			usrc[u] = src[k];
			if (k == 0 || key != keys[k - 1])
			{
				const int64_t ub = ubpos[k];
				ublk_ptr[ub] = u;
				ublk_off[ub] = (uint32_t)(pos[k] - ub);
				atomicAdd((unsigned long long *)&udeg[gi], 1ull);
			}
		}
	}

	// for every block above the diagonal, the index of its mirror image (row = its column node, column = its row node);
	// 0xFFFFFFFF for the diagonal blocks
	__global__ void k_mirror_index(const int64_t *__restrict__ node_ptr, const int32_t *__restrict__ node_col, const int64_t *__restrict__ unode_ptr,
								   int64_t n_nodes, uint32_t *__restrict__ ulow, int *__restrict__ bad)
	{
		for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_nodes; g += (int64_t)gridDim.x * blockDim.x)
		{
			const int64_t ub0 = unode_ptr[g], n_up = unode_ptr[g + 1] - ub0;
			const int64_t b0 = node_ptr[g + 1] - n_up; // the upper blocks are the last ones of the row
			for (int64_t k = 0; k < n_up; ++k)
			{
				const int64_t c = node_col[b0 + k];
				if (c == g)
				{
					ulow[ub0 + k] = 0xFFFFFFFFu;
					continue;
				}
				int64_t lo = node_ptr[c], hi = node_ptr[c + 1]; // first position of row c with column >= g
				while (lo < hi)
				{
					const int64_t mid = (lo + hi) >> 1;
					if (node_col[mid] < g)
						lo = mid + 1;
					else
						hi = mid;
				}
				if (c < g || lo >= node_ptr[c + 1] || node_col[lo] != g)
				{
					*bad = 1; // cannot happen for a pattern built from element pairs
					ulow[ub0 + k] = 0xFFFFFFFFu;
				}
				else
					ulow[ub0 + k] = (uint32_t)lo;
			}
		}
	}

	// key that orders the contributions by (assembly window, element): see pfx_ctx::d_blk_src
	__global__ void k_window_keys(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ src, const int64_t *__restrict__ node_ptr,
								  const int64_t *__restrict__ blk_ptr, int64_t n_items, int64_t n_nodes, uint32_t nl2,
								  uint64_t *__restrict__ wkeys, uint64_t *__restrict__ wvals)
	{
		for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_items; c += (int64_t)gridDim.x * blockDim.x)
		{
			const int64_t gi = (int64_t)(keys[c] / (uint64_t)n_nodes);
			const int64_t g0 = (gi / PFX_ROWS_PER_GROUP) * PFX_ROWS_PER_GROUP;
			const int64_t wbeg = blk_ptr[node_ptr[g0]];
			const int64_t wstart = wbeg + ((c - wbeg) / 64) * 64;
			const uint32_t s = src[c];
			wkeys[c] = ((uint64_t)wstart << 28) | (uint64_t)(s / nl2);
			wvals[c] = ((uint64_t)s << 8) | (uint64_t)(c - wstart);
		}
	}

	__global__ void k_split_window_vals(const uint64_t *__restrict__ wvals, int64_t n_items, uint32_t *__restrict__ src, uint8_t *__restrict__ slot)
	{
		for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_items; c += (int64_t)gridDim.x * blockDim.x)
		{
			const uint64_t v = wvals[c];
			src[c] = (uint32_t)(v >> 8);
			slot[c] = (uint8_t)(v & 0xff);
		}
	}

	// groups of PFX_ROWS_PER_GROUP node rows: first contribution and number of 64-wide windows of each
	__global__ void k_group_windows(const int64_t *__restrict__ unode_ptr, const int64_t *__restrict__ blk_ptr, int64_t n_nodes, int64_t n_groups,
									uint32_t *__restrict__ grp_c0, uint32_t *__restrict__ n_win)
	{
		for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g <= n_groups; g += (int64_t)gridDim.x * blockDim.x)
		{
			const int64_t r0 = g * PFX_ROWS_PER_GROUP < n_nodes ? g * PFX_ROWS_PER_GROUP : n_nodes;
			const int64_t r1 = r0 + PFX_ROWS_PER_GROUP < n_nodes ? r0 + PFX_ROWS_PER_GROUP : n_nodes;
			const int64_t c0 = blk_ptr[unode_ptr[r0]], c1 = blk_ptr[unode_ptr[r1]];
			grp_c0[g] = (uint32_t)c0;
			n_win[g] = (uint32_t)((c1 - c0 + 63) / 64);
		}
	}

	// Walks the windows of every group once and writes down what the assembly kernel needs to know about each:
	// win[W] = (first upper block, number of blocks) and one descriptor per (window, block it intersects).  A window
	// boundary cuts at most one block, which then appears in both windows: block ub of window W sits at desc[ub + W].
	__global__ void k_window_descriptors(const int64_t *__restrict__ unode_ptr, const int64_t *__restrict__ blk_ptr, const uint32_t *__restrict__ ublk_off,
										 const uint32_t *__restrict__ ublk_low, const uint32_t *__restrict__ grp_win, int64_t n_nodes, int64_t n_groups,
										 uint2 *__restrict__ win, uint4 *__restrict__ desc)
	{
		for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_groups; g += (int64_t)gridDim.x * blockDim.x)
		{
			const int64_t r0 = g * PFX_ROWS_PER_GROUP;
			const int64_t r1 = r0 + PFX_ROWS_PER_GROUP < n_nodes ? r0 + PFX_ROWS_PER_GROUP : n_nodes;
			const int64_t b_end = unode_ptr[r1];
			int64_t bl = unode_ptr[r0];
			const int64_t c_end = blk_ptr[b_end];
			int64_t W = grp_win[g];
			for (int64_t w0 = blk_ptr[bl]; w0 < c_end; w0 += 64, ++W)
			{
				const int64_t nw = c_end - w0 < 64 ? c_end - w0 : 64;
				int64_t b = bl;
				bool cut = false;
				while (b < b_end && blk_ptr[b] < w0 + nw)
				{
					const int64_t s0 = blk_ptr[b] - w0, s1 = blk_ptr[b + 1] - w0;
					const uint32_t c0 = (uint32_t)(s0 < 0 ? 0 : s0), c1 = (uint32_t)(s1 > nw ? nw : s1);
					cut = s1 > nw;
					desc[b + W] = make_uint4((uint32_t)b + ublk_off[b], ublk_low[b], c0 | (c1 << 8) | (s0 < 0 ? PFX_DESC_CONT_IN : 0u) | (cut ? PFX_DESC_CONT_OUT : 0u), 0u);
					++b;
				}
				win[W] = make_uint2((uint32_t)bl, (uint32_t)(b - bl));
				bl = cut ? b - 1 : b;
			}
		}
	}

	__global__ void k_count_u32(const uint32_t *__restrict__ keys, int64_t n_items, int64_t *__restrict__ cnt)
	{
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_items; k += (int64_t)gridDim.x * blockDim.x)
			atomicAdd((unsigned long long *)&cnt[keys[k]], 1ull);
	}

	__global__ void k_set_i64(int64_t *p, int64_t v) { *p = v; }

#ifndef PFX_ROWASM_MORTON
#define PFX_ROWASM_MORTON 1
#endif
	// position of a row group: centroid of the geometry vertices of the first element incident to its middle row
	__global__ void k_group_pos(const int64_t *__restrict__ ne_ptr, const uint32_t *__restrict__ ne_src, const int32_t *__restrict__ gconn,
								const double *__restrict__ xyz, int n_loc, int n_gloc, int64_t n_nodes, int64_t n_groups, double *__restrict__ gpos)
	{
		for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_groups; g += (int64_t)gridDim.x * blockDim.x)
		{
			int64_t r = g * PFX_ROWS_PER_GROUP + PFX_ROWS_PER_GROUP / 2;
			if (r >= n_nodes)
				r = n_nodes - 1;
			// a node without elements (never the case for a mesh's own nodes) takes the nearest row that has one
			int64_t q = r;
			while (q > 0 && ne_ptr[q + 1] == ne_ptr[q])
				--q;
			double c[3] = {0, 0, 0};
			if (ne_ptr[q + 1] > ne_ptr[q])
			{
				const int64_t e = (int64_t)(ne_src[ne_ptr[q]] / (uint32_t)n_loc);
				for (int v = 0; v < n_gloc; ++v)
				{
					const double *x = xyz + 3 * (int64_t)gconn[e * n_gloc + v];
					c[0] += x[0], c[1] += x[1], c[2] += x[2];
				}
				c[0] /= n_gloc, c[1] /= n_gloc, c[2] /= n_gloc;
			}
			gpos[3 * g] = c[0], gpos[3 * g + 1] = c[1], gpos[3 * g + 2] = c[2];
		}
	}

	__device__ __forceinline__ uint64_t spread21(uint64_t v)
	{
		v &= 0x1fffffull;
		v = (v | v << 32) & 0x1f00000000ffffull;
		v = (v | v << 16) & 0x1f0000ff0000ffull;
		v = (v | v << 8) & 0x100f00f00f00f00full;
		v = (v | v << 4) & 0x10c30c30c30c30c3ull;
		v = (v | v << 2) & 0x1249249249249249ull;
		return v;
	}

	// Morton key of a group's position on a lattice of cells of size h from the corner (x0, y0, z0); `on` = 0 keeps the
	// caller's order
	__global__ void k_group_morton(const double *__restrict__ gpos, double x0, double y0, double z0, double inv_h, int64_t n_groups, int on,
								   uint64_t *__restrict__ keys, uint32_t *__restrict__ vals)
	{
		for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_groups; g += (int64_t)gridDim.x * blockDim.x)
		{
			const double fx = (gpos[3 * g] - x0) * inv_h, fy = (gpos[3 * g + 1] - y0) * inv_h, fz = (gpos[3 * g + 2] - z0) * inv_h;
			const uint64_t ix = (uint64_t)(fx < 0 ? 0 : fx > 2097151.0 ? 2097151.0 : fx);
			const uint64_t iy = (uint64_t)(fy < 0 ? 0 : fy > 2097151.0 ? 2097151.0 : fy);
			const uint64_t iz = (uint64_t)(fz < 0 ? 0 : fz > 2097151.0 ? 2097151.0 : fz);
			keys[g] = on ? (spread21(ix) | spread21(iy) << 1 | spread21(iz) << 2) : (uint64_t)g;
			vals[g] = (uint32_t)g;
		}
	}

	// scalar CSR export: one thread per scalar row
	__global__ void k_export_pattern(const int64_t *__restrict__ node_ptr, const int32_t *__restrict__ node_col, int64_t n_nodes, int bs,
									 int64_t *__restrict__ row_ptr, int32_t *__restrict__ col_idx)
	{
		const int64_t n_rows = n_nodes * bs;
		for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= n_rows; r += (int64_t)gridDim.x * blockDim.x)
		{
			if (r == n_rows)
			{
				row_ptr[r] = node_ptr[n_nodes] * bs * bs;
				continue;
			}
			const int64_t g = r / bs;
			const int m = (int)(r - g * bs);
			const int64_t b0 = node_ptr[g], deg = node_ptr[g + 1] - b0;
			const int64_t start = b0 * bs * bs + (int64_t)m * deg * bs;
			row_ptr[r] = start;
			for (int64_t s = 0; s < deg; ++s)
				for (int n = 0; n < bs; ++n)
					col_idx[start + s * bs + n] = node_col[b0 + s] * bs + n;
		}
	}

	int bits_for(uint64_t v)
	{
		int b = 1;
		while (b < 64 && (v >> b))
			++b;
		return b;
	}

} // namespace

pfx_status pfx_pattern_build(pfx_ctx *ctx)
{
	const int n_loc = ctx->n_loc;
	const int64_t n_el = ctx->n_el, n_nodes = ctx->n_nodes;
	const int64_t n_pair = n_el * n_loc * n_loc;
	const int64_t n_inc = n_el * n_loc;
	if (n_pair >= ((int64_t)1 << 32))
		PFX_FAIL(PFX_ERR_SIZE, "n_el * n_loc^2 exceeds the 32-bit contribution index; partition the mesh over more devices");
	if (n_el >= ((int64_t)1 << 28))
		PFX_FAIL(PFX_ERR_SIZE, "more than 2^28 elements on one device; partition the mesh over more devices");
	hipStream_t st = ctx->stream;

	// ---- node-pair blocks ----------------------------------------------------------------------------------
	pfx_tmp<uint64_t> keys_a, keys_b;
	pfx_tmp<uint32_t> vals_a, vals_b;
	pfx_tmp<int64_t> pos, deg;
	pfx_tmp<char> tmp;
	size_t tmp_bytes = 0;
	PFX_HIP(keys_a.alloc((size_t)n_pair));
	PFX_HIP(keys_b.alloc((size_t)n_pair));
	PFX_HIP(vals_a.alloc((size_t)n_pair));
	PFX_HIP(vals_b.alloc((size_t)n_pair));
	hipLaunchKernelGGL(k_pair_keys, dim3(grid_for(n_pair)), dim3(256), 0, st, ctx->d_conn, n_pair, n_loc, n_nodes, keys_a.p, vals_a.p);
	const int end_bit = bits_for((uint64_t)n_nodes * (uint64_t)n_nodes);
	PFX_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_a.p, keys_b.p, vals_a.p, vals_b.p, (size_t)n_pair, 0, end_bit, st));
	PFX_HIP(tmp.alloc(tmp_bytes));
	PFX_HIP(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, keys_a.p, keys_b.p, vals_a.p, vals_b.p, (size_t)n_pair, 0, end_bit, st));
	PFX_HIP(hipStreamSynchronize(st));
	tmp.reset();
	keys_a.reset();
	vals_a.reset();

	// positions of run heads
	PFX_HIP(pos.alloc((size_t)(n_pair + 1)));
	auto flags = rocprim::make_transform_iterator(rocprim::make_counting_iterator<int64_t>(0), head_flag{keys_b.p});
	tmp_bytes = 0;
	PFX_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, flags, pos.p, (int64_t)0, (size_t)n_pair, rocprim::plus<int64_t>(), st));
	PFX_HIP(tmp.alloc(tmp_bytes));
	PFX_HIP(rocprim::exclusive_scan(tmp.p, tmp_bytes, flags, pos.p, (int64_t)0, (size_t)n_pair, rocprim::plus<int64_t>(), st));
	int64_t last_pos = 0;
	PFX_HIP(hipMemcpyAsync(&last_pos, pos.p + (n_pair - 1), sizeof(int64_t), hipMemcpyDeviceToHost, st));
	PFX_HIP(hipStreamSynchronize(st));
	tmp.reset();
	// the last item is a head iff its key differs from its predecessor; count = pos[last] + flag[last]
	uint64_t k_last[2] = {0, 0};
	if (n_pair >= 2)
		PFX_HIP(hipMemcpy(k_last, keys_b.p + (n_pair - 2), 2 * sizeof(uint64_t), hipMemcpyDeviceToHost));
	const int64_t last_flag = (n_pair == 1) ? 1 : (k_last[0] != k_last[1] ? 1 : 0);
	const int64_t n_blocks = last_pos + last_flag;
	ctx->n_blocks = n_blocks;
	ctx->nnz = n_blocks * ctx->bs * ctx->bs;

	if (n_blocks >= ((int64_t)1 << 32))
		PFX_FAIL(PFX_ERR_SIZE, "more than 2^32 node blocks on one device; partition the mesh over more devices");
	if (pfx_alloc(ctx, &ctx->d_node_col, n_blocks) || pfx_alloc(ctx, &ctx->d_node_ptr, n_nodes + 1))
		return PFX_ERR_HIP;
	PFX_HIP(deg.alloc((size_t)(n_nodes + 1)));
	PFX_HIP(hipMemsetAsync(deg.p, 0, sizeof(int64_t) * (size_t)(n_nodes + 1), st));
	hipLaunchKernelGGL(k_fill_blocks, dim3(grid_for(n_pair)), dim3(256), 0, st, keys_b.p, pos.p, n_pair, n_nodes, ctx->d_node_col, deg.p);
	tmp_bytes = 0;
	PFX_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, deg.p, ctx->d_node_ptr, (int64_t)0, (size_t)(n_nodes + 1), rocprim::plus<int64_t>(), st));
	PFX_HIP(tmp.alloc(tmp_bytes));
	PFX_HIP(rocprim::exclusive_scan(tmp.p, tmp_bytes, deg.p, ctx->d_node_ptr, (int64_t)0, (size_t)(n_nodes + 1), rocprim::plus<int64_t>(), st));
	PFX_HIP(hipStreamSynchronize(st));
	tmp.reset();

	// ---- blocks on and above the diagonal: the only ones with a contribution list ------------------------------
	// (lists of the upper blocks: needed only until the window descriptors are written)
	int64_t n_upair = 0, n_ublocks = 0;
	pfx_tmp<uint64_t> ukeys;
	pfx_tmp<int64_t> unode_ptr, blk_ptr; // n_nodes+1: upper blocks before each row; n_ublocks+1: contribution range of each upper block
	pfx_tmp<uint32_t> ublk_off, ublk_low; // (index among all blocks) - (index among the upper blocks); index of the mirror image
	{
		pfx_tmp<int64_t> upos, ubpos;
		PFX_HIP(upos.alloc((size_t)(n_pair + 1)));
		PFX_HIP(ubpos.alloc((size_t)(n_pair + 1)));
		for (int pass = 0; pass < 2; ++pass)
		{
			auto uf = rocprim::make_transform_iterator(rocprim::make_counting_iterator<int64_t>(0), upper_flag{keys_b.p, n_pair, (uint64_t)n_nodes, pass == 1});
			int64_t *dst = pass == 0 ? upos.p : ubpos.p;
			tmp_bytes = 0;
			PFX_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, uf, dst, (int64_t)0, (size_t)(n_pair + 1), rocprim::plus<int64_t>(), st));
			PFX_HIP(tmp.alloc(tmp_bytes));
			PFX_HIP(rocprim::exclusive_scan(tmp.p, tmp_bytes, uf, dst, (int64_t)0, (size_t)(n_pair + 1), rocprim::plus<int64_t>(), st));
			PFX_HIP(hipMemcpyAsync(pass == 0 ? &n_upair : &n_ublocks, dst + n_pair, sizeof(int64_t), hipMemcpyDeviceToHost, st));
			PFX_HIP(hipStreamSynchronize(st));
			tmp.reset();
		}
		ctx->n_ublocks = n_ublocks;
		ctx->n_upair = n_upair;
		PFX_HIP(ukeys.alloc((size_t)n_upair));
		if (pfx_alloc(ctx, &ctx->d_blk_src, n_upair))
			return PFX_ERR_HIP;
		PFX_HIP(blk_ptr.alloc((size_t)(n_ublocks + 1)));
		PFX_HIP(ublk_off.alloc((size_t)n_ublocks));
		PFX_HIP(unode_ptr.alloc((size_t)(n_nodes + 1)));
		PFX_HIP(hipMemsetAsync(deg.p, 0, sizeof(int64_t) * (size_t)(n_nodes + 1), st));
		hipLaunchKernelGGL(k_compact_upper, dim3(grid_for(n_pair)), dim3(256), 0, st, keys_b.p, vals_b.p, upos.p, ubpos.p, pos.p, n_pair, n_nodes, ukeys.p,
						   ctx->d_blk_src, blk_ptr.p, ublk_off.p, deg.p);
		hipLaunchKernelGGL(k_set_i64, dim3(1), dim3(1), 0, st, blk_ptr.p + n_ublocks, n_upair);
		tmp_bytes = 0;
		PFX_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, deg.p, unode_ptr.p, (int64_t)0, (size_t)(n_nodes + 1), rocprim::plus<int64_t>(), st));
		PFX_HIP(tmp.alloc(tmp_bytes));
		PFX_HIP(rocprim::exclusive_scan(tmp.p, tmp_bytes, deg.p, unode_ptr.p, (int64_t)0, (size_t)(n_nodes + 1), rocprim::plus<int64_t>(), st));
		PFX_HIP(hipStreamSynchronize(st));
		tmp.reset();
	}
	pos.reset();
	keys_b.reset();
	vals_b.reset();
	{
		PFX_HIP(ublk_low.alloc((size_t)n_ublocks));
		pfx_tmp<int> bad;
		PFX_HIP(bad.alloc(1));
		PFX_HIP(hipMemsetAsync(bad.p, 0, sizeof(int), st));
		hipLaunchKernelGGL(k_mirror_index, dim3(grid_for(n_nodes)), dim3(256), 0, st, ctx->d_node_ptr, ctx->d_node_col, unode_ptr.p, n_nodes,
						   ublk_low.p, bad.p);
		int h_bad = 0;
		PFX_HIP(hipMemcpyAsync(&h_bad, bad.p, sizeof(int), hipMemcpyDeviceToHost, st));
		PFX_HIP(hipStreamSynchronize(st));
		if (h_bad)
			PFX_FAIL(PFX_ERR_SIZE, "internal: the node-block pattern is not structurally symmetric");
	}

	// ---- compute order of the contributions inside each assembly window -----------------------------------------
	{
		pfx_tmp<uint64_t> wk_a, wk_b, wv_a, wv_b;
		PFX_HIP(wk_a.alloc((size_t)n_upair));
		PFX_HIP(wk_b.alloc((size_t)n_upair));
		PFX_HIP(wv_a.alloc((size_t)n_upair));
		PFX_HIP(wv_b.alloc((size_t)n_upair));
		if (pfx_alloc(ctx, &ctx->d_blk_slot, n_upair))
			return PFX_ERR_HIP;
		hipLaunchKernelGGL(k_window_keys, dim3(grid_for(n_upair)), dim3(256), 0, st, ukeys.p, ctx->d_blk_src, unode_ptr.p, blk_ptr.p, n_upair,
						   n_nodes, (uint32_t)(n_loc * n_loc), wk_a.p, wv_a.p);
		const int wbits = 28 + bits_for((uint64_t)n_upair);
		tmp_bytes = 0;
		PFX_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, wk_a.p, wk_b.p, wv_a.p, wv_b.p, (size_t)n_upair, 0, wbits, st));
		PFX_HIP(tmp.alloc(tmp_bytes));
		PFX_HIP(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, wk_a.p, wk_b.p, wv_a.p, wv_b.p, (size_t)n_upair, 0, wbits, st));
		hipLaunchKernelGGL(k_split_window_vals, dim3(grid_for(n_upair)), dim3(256), 0, st, wv_b.p, n_upair, ctx->d_blk_src, ctx->d_blk_slot);
		PFX_HIP(hipStreamSynchronize(st));
		tmp.reset();
	}
	ukeys.reset();

	// ---- what every assembly window needs to know about its blocks ------------------------------------------------
	{
		const int64_t n_groups = (n_nodes + PFX_ROWS_PER_GROUP - 1) / PFX_ROWS_PER_GROUP;
		ctx->n_groups = n_groups;
		pfx_tmp<uint32_t> n_win;
		PFX_HIP(n_win.alloc((size_t)(n_groups + 1)));
		if (pfx_alloc(ctx, &ctx->d_grp_c0, n_groups + 1) || pfx_alloc(ctx, &ctx->d_grp_win, n_groups + 1))
			return PFX_ERR_HIP;
		hipLaunchKernelGGL(k_group_windows, dim3(grid_for(n_groups + 1)), dim3(256), 0, st, unode_ptr.p, blk_ptr.p, n_nodes, n_groups, ctx->d_grp_c0, n_win.p);
		tmp_bytes = 0;
		PFX_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, n_win.p, ctx->d_grp_win, (uint32_t)0, (size_t)(n_groups + 1), rocprim::plus<uint32_t>(), st));
		PFX_HIP(tmp.alloc(tmp_bytes));
		PFX_HIP(rocprim::exclusive_scan(tmp.p, tmp_bytes, n_win.p, ctx->d_grp_win, (uint32_t)0, (size_t)(n_groups + 1), rocprim::plus<uint32_t>(), st));
		uint32_t n_windows = 0;
		PFX_HIP(hipMemcpyAsync(&n_windows, ctx->d_grp_win + n_groups, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
		PFX_HIP(hipStreamSynchronize(st));
		tmp.reset();
		ctx->n_windows = n_windows;
		ctx->h_grp_win.resize((size_t)(n_groups + 1));
		PFX_HIP(hipMemcpyAsync(ctx->h_grp_win.data(), ctx->d_grp_win, sizeof(uint32_t) * (size_t)(n_groups + 1), hipMemcpyDeviceToHost, st));
		PFX_HIP(hipStreamSynchronize(st));
		if (pfx_alloc(ctx, &ctx->d_win, (int64_t)n_windows) || pfx_alloc(ctx, &ctx->d_desc, n_ublocks + (int64_t)n_windows))
			return PFX_ERR_HIP;
		hipLaunchKernelGGL(k_window_descriptors, dim3(grid_for(n_groups)), dim3(256), 0, st, unode_ptr.p, blk_ptr.p, ublk_off.p, ublk_low.p, ctx->d_grp_win, n_nodes,
						   n_groups, ctx->d_win, ctx->d_desc);
		PFX_HIP(hipStreamSynchronize(st));
	}
	unode_ptr.reset();
	blk_ptr.reset();
	ublk_off.reset();
	ublk_low.reset();

	// ---- node -> (element, local index) incidences ---------------------------------------------------------
	pfx_tmp<uint32_t> nk_a, nk_b, nv_a;
	PFX_HIP(nk_a.alloc((size_t)n_inc));
	PFX_HIP(nk_b.alloc((size_t)n_inc));
	PFX_HIP(nv_a.alloc((size_t)n_inc));
	if (pfx_alloc(ctx, &ctx->d_ne_src, n_inc) || pfx_alloc(ctx, &ctx->d_ne_ptr, n_nodes + 1))
		return PFX_ERR_HIP;
	hipLaunchKernelGGL(k_node_keys, dim3(grid_for(n_inc)), dim3(256), 0, st, ctx->d_conn, n_inc, nk_a.p, nv_a.p);
	const int nb = bits_for((uint64_t)n_nodes);
	tmp_bytes = 0;
	PFX_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, nk_a.p, nk_b.p, nv_a.p, ctx->d_ne_src, (size_t)n_inc, 0, nb, st));
	PFX_HIP(tmp.alloc(tmp_bytes));
	PFX_HIP(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, nk_a.p, nk_b.p, nv_a.p, ctx->d_ne_src, (size_t)n_inc, 0, nb, st));
	PFX_HIP(hipMemsetAsync(deg.p, 0, sizeof(int64_t) * (size_t)(n_nodes + 1), st));
	hipLaunchKernelGGL(k_count_u32, dim3(grid_for(n_inc)), dim3(256), 0, st, nk_b.p, n_inc, deg.p);
	PFX_HIP(hipStreamSynchronize(st));
	tmp.reset();
	tmp_bytes = 0;
	PFX_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, deg.p, ctx->d_ne_ptr, (int64_t)0, (size_t)(n_nodes + 1), rocprim::plus<int64_t>(), st));
	PFX_HIP(tmp.alloc(tmp_bytes));
	PFX_HIP(rocprim::exclusive_scan(tmp.p, tmp_bytes, deg.p, ctx->d_ne_ptr, (int64_t)0, (size_t)(n_nodes + 1), rocprim::plus<int64_t>(), st));
	PFX_HIP(hipStreamSynchronize(st));
	PFX_HIP(hipGetLastError());
	// ---- order in which the row assembly takes the row groups ---------------------------------------------------------
	// No device memory is allocated for this step beyond the order itself: its scratch is carved out of the temporaries of
	// the incidence sort above, which are still alive.  (Measured: a few tens of MB allocated and freed at this point move
	// the physical pages that the solver's arrays get afterwards, and the CG product of a 10 M-tet mesh runs 3 % slower
	// in the placement that results -- same virtual addresses, same kernel.)
	{
		const int64_t n_groups = ctx->n_groups;
		if (pfx_alloc(ctx, &ctx->d_grp_order, n_groups))
			return PFX_ERR_HIP;
		auto round256 = [](size_t b) { return (b + 255) / 256 * 256; };
		const size_t cap = sizeof(uint32_t) * (size_t)n_inc; // bytes of each of nk_a, nk_b, nv_a
		const size_t need_a = round256(sizeof(double) * 3 * (size_t)n_groups);
		const size_t need_b = 2 * round256(sizeof(uint64_t) * (size_t)n_groups) + round256(sizeof(uint32_t) * (size_t)n_groups);
		pfx_tmp<char> own_a, own_b, own_c; // only for meshes so small that the carved scratch does not fit
		char *sa = reinterpret_cast<char *>(nk_a.p), *sb = reinterpret_cast<char *>(nk_b.p);
		if (need_a > cap)
		{
			PFX_HIP(own_a.alloc(need_a));
			sa = own_a.p;
		}
		if (need_b > cap)
		{
			PFX_HIP(own_b.alloc(need_b));
			sb = own_b.p;
		}
		double *gpos = reinterpret_cast<double *>(sa);
		uint64_t *gk_a = reinterpret_cast<uint64_t *>(sb), *gk_b = reinterpret_cast<uint64_t *>(sb + round256(sizeof(uint64_t) * (size_t)n_groups));
		uint32_t *gv_a = reinterpret_cast<uint32_t *>(sb + 2 * round256(sizeof(uint64_t) * (size_t)n_groups));
		hipLaunchKernelGGL(k_group_pos, dim3(grid_for(n_groups)), dim3(256), 0, st, ctx->d_ne_ptr, ctx->d_ne_src, ctx->d_gconn, ctx->d_xyz, n_loc, ctx->n_gloc, n_nodes,
						   n_groups, gpos);
		std::vector<double> h_pos((size_t)(3 * n_groups));
		PFX_HIP(hipMemcpyAsync(h_pos.data(), gpos, sizeof(double) * h_pos.size(), hipMemcpyDeviceToHost, st));
		PFX_HIP(hipStreamSynchronize(st));
		double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
		for (int64_t g = 0; g < n_groups; ++g)
			for (int d = 0; d < 3; ++d)
			{
				lo[d] = std::min(lo[d], h_pos[(size_t)(3 * g + d)]);
				hi[d] = std::max(hi[d], h_pos[(size_t)(3 * g + d)]);
			}
		// cells of about one group each: the cube root of the bounding volume per group (flat directions do not count)
		double vol = 1.0, longest = 0.0;
		int dims = 0;
		for (int d = 0; d < 3; ++d)
			longest = std::max(longest, hi[d] - lo[d]);
		for (int d = 0; d < 3; ++d)
			if (hi[d] - lo[d] > 1e-9 * longest && longest > 0)
				vol *= hi[d] - lo[d], ++dims;
		double h = dims > 0 ? std::pow(vol / (double)n_groups, 1.0 / dims) : 1.0;
		if (!(h > 0) || !std::isfinite(h))
			h = 1.0;
		if (longest / h > 2.0e6)
			h = longest / 2.0e6;
		// Elements of more than ten nodes (Q2) keep the caller's order: their functors' tables are shared by workgroups of
		// several waves, which take consecutive groups and should finish together -- neighbours along a lattice line are of
		// similar length, neighbours along the curve are not (measured: 8 % slower with the curve, which their
		// arithmetic-bound kernels have no use for).
		const int reorder = (PFX_ROWASM_MORTON && n_loc <= 10) ? 1 : 0;
		hipLaunchKernelGGL(k_group_morton, dim3(grid_for(n_groups)), dim3(256), 0, st, gpos, lo[0], lo[1], lo[2], 1.0 / h, n_groups, reorder, gk_a, gv_a);
		tmp_bytes = 0;
		PFX_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, gk_a, gk_b, gv_a, ctx->d_grp_order, (size_t)n_groups, 0, 64, st));
		char *sc = reinterpret_cast<char *>(nv_a.p);
		if (tmp_bytes > cap)
		{
			PFX_HIP(own_c.alloc(tmp_bytes));
			sc = own_c.p;
		}
		PFX_HIP(rocprim::radix_sort_pairs(sc, tmp_bytes, gk_a, gk_b, gv_a, ctx->d_grp_order, (size_t)n_groups, 0, 64, st));
		std::vector<uint32_t> h_order((size_t)n_groups);
		PFX_HIP(hipMemcpyAsync(h_order.data(), ctx->d_grp_order, sizeof(uint32_t) * (size_t)n_groups, hipMemcpyDeviceToHost, st));
		PFX_HIP(hipStreamSynchronize(st));
		ctx->h_grp_work.assign((size_t)(n_groups + 1), 0.0);
		for (int64_t k = 0; k < n_groups; ++k)
		{
			const uint32_t g = h_order[(size_t)k];
			ctx->h_grp_work[(size_t)(k + 1)] = ctx->h_grp_work[(size_t)k] + (double)(ctx->h_grp_win[g + 1] - ctx->h_grp_win[g]) + 1.0;
		}
	}
	// XCD ranges of the vector gather (one block per 256 scalar rows), cut by the number of incidences the rows sum: the
	// few entries of ne_ptr the bisection looks at are read from the device one by one
	{
		const int64_t bs = ctx->bs, n_rows = n_nodes * bs, n_blk = (n_rows + 255) / 256;
		std::map<int64_t, int64_t> seen;
		hipError_t err = hipSuccess;
		auto ne_at = [&](int64_t node) {
			auto it = seen.find(node);
			if (it != seen.end())
				return it->second;
			int64_t v = 0;
			const hipError_t e = hipMemcpy(&v, ctx->d_ne_ptr + node, sizeof(int64_t), hipMemcpyDeviceToHost);
			if (e != hipSuccess)
				err = e;
			seen[node] = v;
			return v;
		};
		ctx->gv_split = pfx_make_xcd_split(n_blk, [&](int64_t k) {
			const int64_t node = std::min(n_nodes, (k * 256) / bs);
			return (double)ne_at(node) * (double)bs + 8.0 * (double)k;
		});
		PFX_HIP(err);
	}
	return PFX_OK;
}

pfx_status pfx_export_pattern(pfx_ctx *ctx, int64_t *row_ptr, int32_t *col_idx)
{
	const int64_t n_rows = ctx->n_dofs();
	pfx_tmp<int64_t> d_rp;
	pfx_tmp<int32_t> d_ci;
	PFX_HIP(d_rp.alloc((size_t)(n_rows + 1)));
	PFX_HIP(d_ci.alloc((size_t)ctx->nnz));
	hipLaunchKernelGGL(k_export_pattern, dim3(grid_for(n_rows + 1)), dim3(256), 0, ctx->stream, ctx->d_node_ptr, ctx->d_node_col, ctx->n_nodes, ctx->bs, d_rp.p, d_ci.p);
	PFX_HIP(hipMemcpyAsync(row_ptr, d_rp.p, sizeof(int64_t) * (size_t)(n_rows + 1), hipMemcpyDeviceToHost, ctx->stream));
	PFX_HIP(hipMemcpyAsync(col_idx, d_ci.p, sizeof(int32_t) * (size_t)ctx->nnz, hipMemcpyDeviceToHost, ctx->stream));
	PFX_HIP(hipStreamSynchronize(ctx->stream));
	return PFX_OK;
}
