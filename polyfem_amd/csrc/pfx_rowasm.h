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
// j <= i does; the block of (lo, hi) is its transpose (Assembler.cpp:179-184 mirrors the same values).  Only the blocks
// on and above the diagonal (column node >= row node) have contribution lists and are summed here; the lane that
// finishes an entry also stores it, transposed, into the block's mirror image below the diagonal, so each pair is
// evaluated once and every entry of the value array is still written exactly once.  It keeps its
// constant tables (reference gradients of the bases at the quadrature points, weights) in LDS: they are indexed by
// the lane's own (i, j), and reading them through the vector memory path would cost more than the arithmetic.
//
// Value layout = node-block storage: block b of the node-block CSR (row node g, slot s = b - node_ptr[g]) holds its
// bs*bs entries contiguously, entry (m, n) at b*bs*bs + m*bs + n.  The upper blocks of a row are its last ones, so a window's
// output is one contiguous run per row it touches (plus the transposed copies in the rows of the mirror images).
// The scalar CSR order the reference uses is produced on export only (pfx_values_to_csr).
#pragma once
#include "pfx_internal.h"

// One 64-lane workgroup per group of RPG consecutive node rows.  The rows' blocks are consecutive in the block list
// and their contributions are consecutive in the contribution list, so the group is one stream of contributions,
// walked in windows of 64:
//   A. lane s looks at block bl+s: where its contribution range starts relative to the window;
//   B. every lane computes one contribution of the window into LDS (the list stores them reordered by element
//      inside each window, so that neighbouring lanes read the same element state, with the slot they belong to);
//   C. every entry of the CSR blocks that intersect the window is owned by one lane, which adds its contributions in
//      ascending order.  A block cut by the window boundary hands its partial sums to the next window through
//      `carry`, so the order of the additions does not depend on where the windows fall.
// The index data of a window are loaded one window ahead.
// MASK: also record, per entry, whether any of its contributions is non-zero (the value-dependent pattern of the
// linear assemblers); the Hessian keeps every structural entry and skips that bookkeeping.
// WPB waves per workgroup, each with its own row group and its own staging area; they share only the functor's tables
// (worth it when those are large: 19 KB for Q2 would otherwise limit a CU to six single-wave workgroups).
template <int NLOC, int BS, int RPG, bool MASK, int WPB, class Op>
__global__ __launch_bounds__(64 * WPB) void k_assemble_rows(const int64_t *__restrict__ node_ptr, const int64_t *__restrict__ blk_ptr, const uint32_t *__restrict__ blk_off, const uint32_t *__restrict__ blk_low,
													   const uint32_t *__restrict__ blk_src, const uint8_t *__restrict__ blk_slot, Op op, int64_t n_nodes,
													   double *__restrict__ values, uint8_t *__restrict__ mask)
{
	constexpr int B2 = BS * BS;
	constexpr uint32_t NL2 = NLOC * NLOC;
	__shared__ double stage_[WPB][64 * B2];
	__shared__ int bp_[WPB][66];        // start of block bl+s in the window, clamped to [-1, nw+1]
	__shared__ uint32_t boff_[WPB][64]; // block bl+s is block bl+s+boff[s] of the whole matrix
	__shared__ uint32_t blow_[WPB][64]; // its mirror image below the diagonal (0xFFFFFFFF: none)
	__shared__ double carry_[WPB][2][B2];
	__shared__ uint8_t carry_any_[WPB][2][B2];
	__shared__ double optab[Op::LDS_DOUBLES]; // the functor's constant tables (reference gradients, weights)
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	double *const stage = stage_[wv];
	int *const bp = bp_[wv];
	uint32_t *const boff = boff_[wv], *const blow = blow_[wv];
	double(*const carry)[B2] = carry_[wv];
	uint8_t(*const carry_any)[B2] = carry_any_[wv];
	op.load_tables(optab, lane); // every wave writes the same values
	const int64_t g_first = ((int64_t)blockIdx.x * WPB + wv) * RPG;
	const bool have_rows = g_first < n_nodes;
	const int64_t g_last = (g_first + RPG <= n_nodes) ? g_first + RPG : n_nodes;
	const int64_t b_begin = have_rows ? node_ptr[g_first] : 0, b_end = have_rows ? node_ptr[g_last] : 0;
	__syncthreads();
	if (b_begin == b_end)
		return;
	// from here on a wave only talks to itself through its own part of the LDS: a wave-level barrier is enough
	// (with one wave per workgroup the workgroup barrier is the same thing)
	auto wave_sync = [] {
		if (WPB == 1)
			__syncthreads();
		else
		{
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		}
	};
	const int64_t c_end = blk_ptr[b_end];
	int64_t bl = b_begin; // first block that intersects the window
	int cur = 0;
	const int64_t w_begin = blk_ptr[b_begin];
	uint32_t src = (w_begin + lane < c_end) ? __builtin_nontemporal_load(blk_src + w_begin + lane) : 0u;
	int slot = (w_begin + lane < c_end) ? __builtin_nontemporal_load(blk_slot + w_begin + lane) : 0;
	int64_t bptr = (bl + lane <= b_end) ? blk_ptr[bl + lane] : c_end + 128;
	int64_t bptr2 = (lane < 2 && bl + 64 + lane <= b_end) ? blk_ptr[bl + 64 + lane] : c_end + 128;
	uint32_t off = (bl + lane < b_end) ? blk_off[bl + lane] : 0u;
	uint32_t low = (bl + lane < b_end) ? blk_low[bl + lane] : 0xFFFFFFFFu;
	for (int64_t w0 = w_begin; w0 < c_end; w0 += 64)
	{
		const int64_t w1 = (w0 + 64 < c_end) ? w0 + 64 : c_end;
		const int nw = (int)(w1 - w0);
		// A
		const int64_t d = bptr - w0;
		const int rel = d < 0 ? -1 : (d > nw ? nw + 1 : (int)d);
		bp[lane] = rel;
		boff[lane] = off;
		blow[lane] = low;
		if (lane < 2)
		{
			const int64_t d2 = bptr2 - w0;
			bp[64 + lane] = d2 > nw ? nw + 1 : (int)d2;
		}
		// blocks bl .. bl+nb-1 intersect the window (every block has at least one contribution)
		const int nb = __popcll(__ballot(rel < nw && bl + lane < b_end));
		// the last block of the window continues into the next one iff the block after it starts beyond the window
		const int next_start = __shfl(rel, nb < 64 ? nb : 0);
		const int64_t d64 = __shfl(bptr2, 0) - w0;
		const bool cont_last = (nb < 64) ? (next_start > nw) : (d64 > nw);
		const int64_t bl_next = bl + (cont_last ? nb - 1 : nb);
		const uint32_t src_now = src;
		const int slot_now = slot;
		if (w1 < c_end)
		{
			src = (w1 + lane < c_end) ? __builtin_nontemporal_load(blk_src + w1 + lane) : 0u;
			slot = (w1 + lane < c_end) ? __builtin_nontemporal_load(blk_slot + w1 + lane) : 0;
			bptr = (bl_next + lane <= b_end) ? blk_ptr[bl_next + lane] : c_end + 128;
			bptr2 = (lane < 2 && bl_next + 64 + lane <= b_end) ? blk_ptr[bl_next + 64 + lane] : c_end + 128;
			off = (bl_next + lane < b_end) ? blk_off[bl_next + lane] : 0u;
			low = (bl_next + lane < b_end) ? blk_low[bl_next + lane] : 0xFFFFFFFFu;
		}
		// B
		if (lane < nw)
		{
			const uint32_t e = src_now / NL2;
			const uint32_t ij = src_now - e * NL2;
			const int i = (int)(ij / NLOC), j = (int)(ij - (ij / NLOC) * NLOC);
			double out[B2];
			op(optab, e, i > j ? i : j, i > j ? j : i, out);
			const bool tr = i < j; // the functor computed the mirror image: stage its transpose
#pragma unroll
			for (int k = 0; k < B2; ++k)
			{
				const int kt = (k % BS) * BS + k / BS;
				stage[slot_now * B2 + k] = (kt != k && tr) ? out[kt] : out[k];
			}
		}
		wave_sync();
		// C: consecutive lanes own consecutive entries of the blocks bl .. bl+nb-1 (contiguous in the value array row by row)
		const int n_items = nb * B2;
		for (int o = lane; o < n_items; o += 64)
		{
			const int s = o / B2, mn = o - s * B2;
			const int cb0 = bp[s], cb1 = bp[s + 1];
			const bool cont_in = cb0 < 0, cont_out = cb1 > nw;
			const int c0 = cont_in ? 0 : cb0, c1 = cont_out ? nw : cb1;
			double acc = cont_in ? carry[cur][mn] : 0.0;
			int any = (MASK && cont_in) ? carry_any[cur][mn] : 0;
			const double *sp = stage + c0 * B2 + mn;
			int c = c0;
			for (; c + 1 < c1; c += 2, sp += 2 * B2) // two contributions per trip, added in list order
			{
				const double v0 = sp[0], v1 = sp[B2];
				acc += v0;
				acc += v1;
				if (MASK)
					any |= (v0 != 0.0) | (v1 != 0.0);
			}
			if (c < c1)
			{
				const double v = sp[0];
				acc += v;
				if (MASK)
					any |= (v != 0.0);
			}
			if (cont_out)
			{
				carry[cur ^ 1][mn] = acc;
				if (MASK)
					carry_any[cur ^ 1][mn] = (uint8_t)any;
			}
			else
			{
				const int64_t at = (bl + (int64_t)boff[s]) * B2 + o;
				values[at] = acc;
				if (MASK)
					mask[at] = (uint8_t)any;
				const uint32_t lw = blow[s];
				if (lw != 0xFFFFFFFFu)
				{
					const int64_t at2 = (int64_t)lw * B2 + (mn % BS) * BS + mn / BS;
					values[at2] = acc;
					if (MASK)
						mask[at2] = (uint8_t)any;
				}
			}
		}
		bl = bl_next;
		cur ^= 1;
		wave_sync();
	}
}

template <int NLOC, int BS, class Op>
inline void pfx_launch_assemble_rows(pfx_ctx *ctx, const Op &op, bool write_mask)
{
	constexpr int RPG = PFX_ROWS_PER_GROUP;
	constexpr int WPB = (Op::LDS_DOUBLES * 8 > 12 * 1024) ? 4 : 1; // share large tables between four waves
	const int64_t groups = (ctx->n_nodes + RPG - 1) / RPG;
	const unsigned grid = (unsigned)((groups + WPB - 1) / WPB);
	if (write_mask)
		hipLaunchKernelGGL((k_assemble_rows<NLOC, BS, RPG, true, WPB, Op>), dim3(grid), dim3(64 * WPB), 0, ctx->stream, ctx->d_unode_ptr, ctx->d_blk_ptr,
						   ctx->d_ublk_off, ctx->d_ublk_low, ctx->d_blk_src, ctx->d_blk_slot, op, ctx->n_nodes, ctx->d_values, ctx->d_mask);
	else
		hipLaunchKernelGGL((k_assemble_rows<NLOC, BS, RPG, false, WPB, Op>), dim3(grid), dim3(64 * WPB), 0, ctx->stream, ctx->d_unode_ptr, ctx->d_blk_ptr,
						   ctx->d_ublk_off, ctx->d_ublk_low, ctx->d_blk_src, ctx->d_blk_slot, op, ctx->n_nodes, ctx->d_values, ctx->d_mask);
}
