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

// One wave per group of PFX_ROWS_PER_GROUP consecutive node rows.  The rows' upper blocks are consecutive in the block
// list and their contributions are consecutive in the contribution list, so the group is one stream of contributions,
// walked in windows of 64.  Everything about a window that depends only on the pattern is worked out when the pattern
// is built (pfx_pattern.hip: k_window_descriptors) and read back here as
//   grp_c0[g], grp_win[g]   first contribution / first window of group g (the next group's entry ends it);
//   win[W] = (bl, nb)       the window's first upper block and how many blocks it intersects;
//   desc[bl + s + W]        for block bl+s of window W: where its contributions sit in the window [c0, c1), whether it
//                           continues from the previous window / into the next one, the index of the block in the value
//                           array and of its mirror image below the diagonal (pfx_window_desc);
// so that the kernel itself is three short phases per window:
//   B. every lane computes one contribution into LDS (the list stores them reordered by element inside each window,
//      so that neighbouring lanes read the same element state, with the slot they belong to);
//   C. every entry of the blocks that intersect the window is owned by one lane, which adds its contributions in
//      list order and stores the entry and its mirror image.  A lane keeps the same entry position (m, n) in every trip
//      (for 3 x 3 blocks 63 lanes take 7 blocks per trip), so its addresses inside a block are loop invariants.
//      A block cut by the window boundary hands its partial sums to the next window through `carry`, so the order of
//      the additions does not depend on where the windows fall.
// The index data of a window are loaded one window ahead (the window headers two ahead: they say which descriptors
// to fetch).
// MASK: also record, per entry, whether any of its contributions is non-zero (the value-dependent pattern of the
// linear assemblers); the Hessian keeps every structural entry and skips that bookkeeping.
// WPB waves per workgroup, each with its own row group and its own staging area; they share only the functor's tables
// (worth it when those are large: 19 KB for Q2 would otherwise limit a CU to six single-wave workgroups).
// Three waves per SIMD is what the largest functors reach without spilling; without the bound the compiler spends
// the registers of a fourth and a third wave on keeping more of an element's state loads in flight, which pays less
// than the extra wave does.
#ifndef PFX_ROWASM_XCD
#define PFX_ROWASM_XCD 1
#endif
#ifndef PFX_ROWASM_NT_STORE
#define PFX_ROWASM_NT_STORE 1
#endif
#ifndef PFX_ROWASM_WPB_BIG
#define PFX_ROWASM_WPB_BIG 4
#endif
// (A functor that cannot live with that many registers says so: Op::MIN_WAVES.)
#ifndef PFX_ROWASM_MINWAVES
#define PFX_ROWASM_MINWAVES 3
#endif
#include <type_traits>
template <class Op, class = void>
struct pfx_min_waves
{
	static constexpr int value = PFX_ROWASM_MINWAVES;
};
template <class Op>
struct pfx_min_waves<Op, std::void_t<decltype(Op::MIN_WAVES)>>
{
	static constexpr int value = Op::MIN_WAVES;
};
template <int NLOC, int BS, bool MASK, int WPB, class Op>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(pfx_min_waves<Op>::value))) void k_assemble_rows(const uint32_t *__restrict__ grp_c0, const uint32_t *__restrict__ grp_win, const uint2 *__restrict__ win,
															 const uint4 *__restrict__ desc, const uint32_t *__restrict__ blk_src, const uint8_t *__restrict__ blk_slot,
															 const uint32_t *__restrict__ grp_order, Op op, int64_t n_groups, pfx_xcd_split split, double *__restrict__ values,
															 uint8_t *__restrict__ mask)
{
	constexpr int B2 = BS * BS;
	constexpr uint32_t NL2 = NLOC * NLOC;
	constexpr int BPT = 64 / B2; // blocks per trip of phase C
	__shared__ double stage_[WPB][64 * B2];
	__shared__ uint4 sdesc_[WPB][64];
	__shared__ double carry_[WPB][2][B2];
	__shared__ uint8_t carry_any_[WPB][2][B2];
	__shared__ double optab[Op::LDS_DOUBLES]; // the functor's constant tables (reference gradients, weights)
	const int lane = threadIdx.x & 63;
	const int wv = WPB == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	double *const stage = stage_[wv];
	uint4 *const sdesc = sdesc_[wv];
	double(*const carry)[B2] = carry_[wv];
	uint8_t(*const carry_any)[B2] = carry_any_[wv];
	op.load_tables(optab, lane); // every wave writes the same values
	// The row groups are taken in the order grp_order (a space-filling curve through their positions) and every XCD takes
	// one contiguous range of that order: groups that are close in space share elements, and the element states the
	// resident workgroups of an XCD read are then fetched into one L2 instead of into all eight, whatever the caller's
	// node numbering is.  The ranges are cut by the host so that they hold equal numbers of windows.
	const int xcd = blockIdx.x % PFX_XCDS;
	const int64_t wg_in = blockIdx.x / PFX_XCDS;
	const int64_t wg = PFX_ROWASM_XCD ? (wg_in < split.first[xcd + 1] - split.first[xcd] ? split.first[xcd] + wg_in : (int64_t)-1) : (int64_t)blockIdx.x;
	const int64_t turn = wg * WPB + wv;
	__syncthreads();
	if (wg < 0 || turn >= n_groups)
		return;
	const int64_t g = grp_order[turn];
	const uint32_t c_begin = grp_c0[g], c_end = grp_c0[g + 1];
	const uint32_t W_begin = grp_win[g], W_end = grp_win[g + 1];
	if (W_begin == W_end)
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
	// phase C: this lane's entry of a block, and which block of a trip it takes
	const int mn = lane % B2, sl = lane / B2;
	const int mn_t = (mn % BS) * BS + mn / BS;
	const bool lane_c = sl < BPT;
	double *const val_mn = values + mn, *const val_t = values + mn_t;
	uint8_t *const mask_mn = mask + mn, *const mask_t = mask + mn_t;

	uint2 hdr = win[W_begin];
	uint2 hdr_next = (W_begin + 1 < W_end) ? win[W_begin + 1] : make_uint2(0u, 0u);
	uint32_t src = (c_begin + lane < c_end) ? __builtin_nontemporal_load(blk_src + c_begin + lane) : 0u;
	int slot = (c_begin + lane < c_end) ? __builtin_nontemporal_load(blk_slot + c_begin + lane) : 0;
	uint4 d = make_uint4(0u, 0u, 0u, 0u);
	if ((uint32_t)lane < hdr.y)
		d = desc[(int64_t)hdr.x + W_begin + lane];
	int cur = 0;
	uint32_t w0 = c_begin;
	for (uint32_t W = W_begin; W < W_end; ++W, w0 += 64)
	{
		const int nw = (c_end - w0 < 64u) ? (int)(c_end - w0) : 64;
		const int nb = (int)hdr.y;
		sdesc[lane] = d;
		const uint32_t src_now = src;
		const int slot_now = slot;
		if (W + 1 < W_end)
		{
			const uint32_t w1 = w0 + 64;
			src = (w1 + lane < c_end) ? __builtin_nontemporal_load(blk_src + w1 + lane) : 0u;
			slot = (w1 + lane < c_end) ? __builtin_nontemporal_load(blk_slot + w1 + lane) : 0;
			hdr = hdr_next;
			if ((uint32_t)lane < hdr.y)
				d = desc[(int64_t)hdr.x + (W + 1) + lane];
			if (W + 2 < W_end)
				hdr_next = win[W + 2];
		}
		// B
		if (lane < nw)
		{
			const uint32_t e = src_now / NL2;
			const uint32_t ij = src_now - e * NL2;
			const int i = (int)(ij / NLOC), j = (int)(ij - (ij / NLOC) * NLOC);
			double out[B2];
			op(optab, e, i > j ? i : j, i > j ? j : i, out);
			if (BS > 1)
			{
				const bool tr = i < j; // the functor computed the mirror image: stage its transpose
#pragma unroll
				for (int m = 0; m < BS; ++m)
#pragma unroll
					for (int n = m + 1; n < BS; ++n)
					{
						const double a = out[m * BS + n], b = out[n * BS + m];
						out[m * BS + n] = tr ? b : a;
						out[n * BS + m] = tr ? a : b;
					}
			}
#pragma unroll
			for (int k = 0; k < B2; ++k)
				stage[slot_now * B2 + k] = out[k];
		}
		wave_sync();
		// C
		if (lane_c)
			for (int s = sl; s < nb; s += BPT)
			{
				const uint4 dd = sdesc[s];
				const int c0 = (int)(dd.z & 0xffu), c1 = (int)((dd.z >> 8) & 0xffu);
				const bool cont_in = (dd.z & PFX_DESC_CONT_IN) != 0, cont_out = (dd.z & PFX_DESC_CONT_OUT) != 0;
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
					const int64_t at = (int64_t)dd.x * B2;
					// the entries of the upper blocks leave in full contiguous runs and are not read again by this kernel: stored past
					// the caches, they do not push the element states out of the L2 (the transposed copies below are pieces of
					// cache lines that other rows complete: those merge in the L2)
#if PFX_ROWASM_NT_STORE
					__builtin_nontemporal_store(acc, val_mn + at);
#else
					val_mn[at] = acc;
#endif
					if (MASK)
						mask_mn[at] = (uint8_t)any;
					if (dd.y != 0xFFFFFFFFu)
					{
						const int64_t at2 = (int64_t)dd.y * B2;
						val_t[at2] = acc;
						if (MASK)
							mask_t[at2] = (uint8_t)any;
					}
				}
			}
		cur ^= 1;
		wave_sync();
	}
}

template <int NLOC, int BS, class Op>
inline void pfx_launch_assemble_rows(pfx_ctx *ctx, const Op &op, bool write_mask)
{
	constexpr int WPB = (Op::LDS_DOUBLES * 8 > 12 * 1024) ? PFX_ROWASM_WPB_BIG : 1; // share large tables between the waves of a workgroup
	// workgroup k holds the row groups k*WPB .. k*WPB+WPB-1; the XCD ranges are cut by the number of windows
	const int64_t n_groups = ctx->n_groups, n_wg = (n_groups + WPB - 1) / WPB;
	const std::vector<double> &work = ctx->h_grp_work;
	const pfx_xcd_split split = pfx_make_xcd_split(n_wg, [&](int64_t k) { return work[(size_t)(k * WPB < n_groups ? k * WPB : n_groups)]; });
	const unsigned grid = PFX_ROWASM_XCD ? (unsigned)(PFX_XCDS * split.widest) : (unsigned)n_wg;
	if (write_mask)
		hipLaunchKernelGGL((k_assemble_rows<NLOC, BS, true, WPB, Op>), dim3(grid), dim3(64 * WPB), 0, ctx->stream, ctx->d_grp_c0, ctx->d_grp_win, ctx->d_win,
						   ctx->d_desc, ctx->d_blk_src, ctx->d_blk_slot, ctx->d_grp_order, op, ctx->n_groups, split, ctx->d_values, ctx->d_mask);
	else
		hipLaunchKernelGGL((k_assemble_rows<NLOC, BS, false, WPB, Op>), dim3(grid), dim3(64 * WPB), 0, ctx->stream, ctx->d_grp_c0, ctx->d_grp_win, ctx->d_win,
						   ctx->d_desc, ctx->d_blk_src, ctx->d_blk_slot, ctx->d_grp_order, op, ctx->n_groups, split, ctx->d_values, ctx->d_mask);
}
