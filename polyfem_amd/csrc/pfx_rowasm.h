// pfx_rowasm.h -- row-owner assembly of the global matrix, shared by the linear and the nonlinear assemblers.
//
// The reference scatters every local block into thread-local triplet / value buffers and merges them afterwards
// (src/polyfem/assembler/Assembler.cpp:185-189, 602 and the merges at :227-301, :627-632;
// src/polyfem/utils/MatrixUtils.cpp:120-130 for the cached pattern).  Here nothing is scattered and no local matrix
// is ever stored: the pattern build lists, for every block of the node-block CSR, the (element, i, j) triples that
// contribute to it (ascending element order).  The kernel walks that list in row order; one lane computes one
// contribution (a bs x bs block) from the per-element state left by the element kernel, and the lane that OWNS a
// global entry adds up its contributions in list order and writes the entry once.  The result is independent of
// scheduling (bitwise reproducible), needs no zero-fill of the value array and no FP64 atomics.
//
// The contribution functor computes the block of the pair (hi, lo), hi >= lo, exactly as the reference's loop over
// j <= i does; the block of (lo, hi) is its transpose (Assembler.cpp:179-184 mirrors the same values).  It keeps its
// constant tables (reference gradients of the bases at the quadrature points, weights) in LDS: they are indexed by
// the lane's own (i, j), and reading them through the vector memory path would cost more than the arithmetic.
//
// Value layout = scalar CSR of the structural pattern: node g owns bs consecutive rows, row g*bs+m is the
// contiguous range [bs*bs*node_ptr[g] + m*deg*bs, +deg*bs) with entry (slot s, n) at offset s*bs+n.
#pragma once
#include "pfx_internal.h"

// One 64-lane workgroup per group of RPG consecutive node rows.  The rows' blocks are consecutive in the block list
// and their contributions are consecutive in the contribution list, so the group is one stream of contributions,
// walked in windows of 64:
//   A. lane s looks at block bl+s: where its contribution range starts relative to the window, and in which row it
//      lies (-> where its entries go in the value array);
//   B. lane c computes contribution c of the window into LDS;
//   C. every entry of the CSR blocks that intersect the window is owned by one lane, which adds its contributions in
//      ascending order.  A block cut by the window boundary hands its partial sums to the next window through
//      `carry`, so the order of the additions does not depend on where the windows fall.
// The index data of a window are loaded one window ahead.
template <int NLOC, int BS, int RPG, class Op>
__global__ __launch_bounds__(64) void k_assemble_rows(const int64_t *__restrict__ node_ptr, const int64_t *__restrict__ blk_ptr,
													   const uint32_t *__restrict__ blk_src, Op op, int64_t n_nodes,
													   double *__restrict__ values, uint8_t *__restrict__ mask, int write_mask)
{
	constexpr int B2 = BS * BS;
	constexpr uint32_t NL2 = NLOC * NLOC;
	__shared__ double stage[64 * B2];
	__shared__ uint8_t src_tr[64];
	__shared__ int64_t row_first[RPG + 1]; // first block of each row of the group
	__shared__ int64_t out_at[64];        // value-array offset of entry (0,0) of block bl+s
	__shared__ int out_rowlen[64];
	__shared__ int bp[66]; // start of block bl+s in the window, clamped to [-1, nw+1]
	__shared__ double carry[2][B2];
	__shared__ uint8_t carry_any[2][B2];
	__shared__ double optab[Op::LDS_DOUBLES]; // the functor's constant tables (reference gradients, weights)
	const int lane = threadIdx.x;
	op.load_tables(optab, lane);
	const int64_t g_first = (int64_t)blockIdx.x * RPG;
	const int n_rows = (int)((g_first + RPG <= n_nodes) ? RPG : (n_nodes - g_first));
	if (lane <= n_rows)
		row_first[lane] = node_ptr[g_first + lane];
	__syncthreads();
	const int64_t b_begin = row_first[0], b_end = row_first[n_rows];
	if (b_begin == b_end)
		return;
	const int64_t c_end = blk_ptr[b_end];
	int64_t bl = b_begin; // first block that intersects the window
	int cur = 0;
	const int64_t w_begin = blk_ptr[b_begin];
	uint32_t src = (w_begin + lane < c_end) ? __builtin_nontemporal_load(blk_src + w_begin + lane) : 0u;
	int64_t bptr = (bl + lane <= b_end) ? blk_ptr[bl + lane] : c_end + 128;
	int64_t bptr2 = (lane < 2 && bl + 64 + lane <= b_end) ? blk_ptr[bl + 64 + lane] : c_end + 128;
	for (int64_t w0 = w_begin; w0 < c_end; w0 += 64)
	{
		const int64_t w1 = (w0 + 64 < c_end) ? w0 + 64 : c_end;
		const int nw = (int)(w1 - w0);
		// A
		const int64_t d = bptr - w0;
		const int rel = d < 0 ? -1 : (d > nw ? nw + 1 : (int)d);
		bp[lane] = rel;
		if (lane < 2)
		{
			const int64_t d2 = bptr2 - w0;
			bp[64 + lane] = d2 > nw ? nw + 1 : (int)d2;
		}
		// blocks bl .. bl+nb-1 intersect the window (every block has at least one contribution)
		const int nb = __popcll(__ballot(rel < nw && bl + lane < b_end));
		{
			const int64_t b = bl + lane;
			if (b < b_end)
			{
				int r = 0;
#pragma unroll
				for (int q = 1; q < RPG; ++q)
					r += (q < n_rows && row_first[q] <= b) ? 1 : 0;
				const int64_t rb0 = row_first[r];
				out_at[lane] = rb0 * B2 + (b - rb0) * BS;
				out_rowlen[lane] = (int)(row_first[r + 1] - rb0) * BS;
			}
		}
		// the last block of the window continues into the next one iff the block after it starts beyond the window
		const int next_start = __shfl(rel, nb < 64 ? nb : 0);
		const int64_t d64 = __shfl(bptr2, 0) - w0;
		const bool cont_last = (nb < 64) ? (next_start > nw) : (d64 > nw);
		const int64_t bl_next = bl + (cont_last ? nb - 1 : nb);
		const uint32_t src_now = src;
		if (w1 < c_end)
		{
			src = (w1 + lane < c_end) ? __builtin_nontemporal_load(blk_src + w1 + lane) : 0u;
			bptr = (bl_next + lane <= b_end) ? blk_ptr[bl_next + lane] : c_end + 128;
			bptr2 = (lane < 2 && bl_next + 64 + lane <= b_end) ? blk_ptr[bl_next + 64 + lane] : c_end + 128;
		}
		// B
		if (lane < nw)
		{
			const uint32_t e = src_now / NL2;
			const uint32_t ij = src_now - e * NL2;
			const int i = (int)(ij / NLOC), j = (int)(ij - (ij / NLOC) * NLOC);
			double out[B2];
			op(optab, e, i > j ? i : j, i > j ? j : i, out);
#pragma unroll
			for (int k = 0; k < B2; ++k)
				stage[lane * B2 + k] = out[k];
			src_tr[lane] = (uint8_t)(i < j);
		}
		__syncthreads();
		// C: entries ordered (m, s, n) so that consecutive lanes write consecutive doubles of one scalar row
		const int per_m = nb * BS;
		for (int m = 0; m < BS; ++m)
			for (int r = lane; r < per_m; r += 64)
			{
				const int s = r / BS, n = r - s * BS;
				const int cb0 = bp[s], cb1 = bp[s + 1];
				const bool cont_in = cb0 < 0, cont_out = cb1 > nw;
				const int c0 = cont_in ? 0 : cb0, c1 = cont_out ? nw : cb1;
				const int mn = m * BS + n, nm = n * BS + m;
				double acc = cont_in ? carry[cur][mn] : 0.0;
				int any = cont_in ? carry_any[cur][mn] : 0;
				for (int c = c0; c < c1; ++c)
				{
					const double v = stage[c * B2 + (src_tr[c] ? nm : mn)];
					acc += v;
					any |= (v != 0.0);
				}
				if (cont_out)
				{
					carry[cur ^ 1][mn] = acc;
					carry_any[cur ^ 1][mn] = (uint8_t)any;
				}
				else
				{
					const int64_t at = out_at[s] + (int64_t)m * out_rowlen[s] + n;
					values[at] = acc;
					if (write_mask)
						mask[at] = (uint8_t)any;
				}
			}
		bl = bl_next;
		cur ^= 1;
		__syncthreads();
	}
}

template <int NLOC, int BS, class Op>
inline void pfx_launch_assemble_rows(pfx_ctx *ctx, const Op &op, bool write_mask)
{
	constexpr int RPG = 16;
	const int64_t groups = (ctx->n_nodes + RPG - 1) / RPG;
	hipLaunchKernelGGL((k_assemble_rows<NLOC, BS, RPG, Op>), dim3((unsigned)groups), dim3(64), 0, ctx->stream, ctx->d_node_ptr, ctx->d_blk_ptr,
					   ctx->d_blk_src, op, ctx->n_nodes, ctx->d_values, ctx->d_mask, write_mask ? 1 : 0);
}
