// pfx_sell.hip -- the solver's matrix format and its matrix-vector product.
//
// Plays the role of polysolve's analyzePattern (structure, once per pattern) + factorize (values, once per
// assembled matrix) for the CG solver (call sites src/polyfem/solver/SparseNewtonDescentSolver.tpp:128-155).
//
// Format: block-SELL-64.  Owned nodes are sorted by degree inside windows of SIGMA consecutive nodes and cut into
// slices of 64; lane l of a slice owns one node = bs matrix rows.  A slice stores, for slot j = 0..len-1 (len = the
// largest degree in the slice, shorter rows padded with zero blocks pointing at the node itself):
//     cols  [j][lane]                      int32 column node
//     values[j][entry][lane]               the bs*bs block entries, lane fastest
// (for bs = 3 entries 0..7 are stored as 4 lane-interleaved double2 so that they are fetched with 16-byte loads).
// Every load of the value stream is therefore one fully coalesced 512 B / 1 KiB wave access, a lane accumulates its
// bs rows in registers (no cross-lane reduction) and the loop over j unrolls freely, which keeps ~36 loads in flight
// per lane.  Measured on MI355X, 1 M P2 tets: 0.62 ms per product against 0.81 ms for a lanes-per-row CSR kernel on
// the same data; a pure streaming read of the value array takes 0.50 ms.  Sorting only inside windows keeps the x
// entries gathered by one slice close together (L2 hits) and the padding below 0.2 %.
#include "pfx_internal.h"
#include <cstdio>

#include <rocprim/rocprim.hpp>

namespace
{
	constexpr int SL = 64;

	__global__ void k_sell_keys(const int64_t *__restrict__ node_ptr, int64_t n_owned, int sigma, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals)
	{
		for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_owned; g += (int64_t)gridDim.x * blockDim.x)
		{
			const uint64_t deg = (uint64_t)(node_ptr[g + 1] - node_ptr[g]);
			keys[g] = ((uint64_t)(g / sigma) << 24) | (0xFFFFFFull - (deg & 0xFFFFFFull));
			vals[g] = (uint32_t)g;
		}
	}

	__global__ void k_sell_perm(const uint32_t *__restrict__ sorted, int64_t n_owned, int64_t n_pad, const int64_t *__restrict__ node_ptr,
								int32_t *__restrict__ perm, int64_t *__restrict__ slice_len)
	{
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_pad; k += (int64_t)gridDim.x * blockDim.x)
		{
			const int32_t g = k < n_owned ? (int32_t)sorted[k] : -1;
			perm[k] = g;
			if ((k & (SL - 1)) == 0)
				slice_len[k / SL] = g >= 0 ? node_ptr[g + 1] - node_ptr[g] : 0; // first lane has the largest degree
		}
	}

	__global__ void k_sell_cols(const int64_t *__restrict__ node_ptr, const int32_t *__restrict__ node_col, const int32_t *__restrict__ perm,
								const int64_t *__restrict__ slice_base, int64_t n_slices, int32_t *__restrict__ cols)
	{
		const int lane = threadIdx.x & 63;
		const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
		const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
		for (int64_t s = wave; s < n_slices; s += n_waves)
		{
			const int64_t base = slice_base[s];
			const int len = (int)(slice_base[s + 1] - base);
			const int g = perm[s * SL + lane];
			const int64_t b0 = g >= 0 ? node_ptr[g] : 0;
			const int deg = g >= 0 ? (int)(node_ptr[g + 1] - b0) : 0;
			for (int j = 0; j < len; ++j)
				cols[(base + j) * SL + lane] = j < deg ? node_col[b0 + j] : (g >= 0 ? g : 0);
		}
	}

	// entry offset inside one slot of a slice.  Block size 3: entries 0..7 are stored lane-interleaved in 16-byte groups
	// (two doubles / four floats per group), entry 8 after them, so that a lane fetches its nine entries with 16-byte loads.
	// Block size 2: the four entries as two such groups of doubles, or one of floats.
	template <int BS, typename V>
	__device__ __forceinline__ int entry_offset(int c, int lane)
	{
		if (BS == 2)
			return sizeof(V) == 8 ? (c >> 1) * 128 + lane * 2 + (c & 1) : lane * 4 + c;
		if (BS == 3)
		{
			if (sizeof(V) == 8)
				return c < 8 ? (c >> 1) * 128 + lane * 2 + (c & 1) : 512 + lane;
			return c < 8 ? (c >> 2) * 256 + lane * 4 + (c & 3) : 512 + lane;
		}
		return c * SL + lane;
	}

	// node-block values -> slice layout.  One wave per slice; JB slots at a time: the JB blocks of each of the 64 rows
	// (JB * bs*bs contiguous doubles per row) are read with the lanes running over consecutive doubles -- cached loads, so
	// that a 128-byte line shared by two neighbouring blocks of a row is fetched once --, transposed through LDS and
	// written with streaming stores as the slots' contiguous runs of the slice.
	template <int BS, int JB, int WPB, typename V>
	__global__ __launch_bounds__(64 * WPB) void k_sell_values(const int64_t *__restrict__ node_ptr, const int32_t *__restrict__ perm,
															   const int64_t *__restrict__ slice_base, int64_t n_slices,
															   const double *__restrict__ blocks, V *__restrict__ out, int xmap)
	{
		constexpr int B2 = BS * BS, RUN = JB * B2; // entries per row and pass
		__shared__ V tile[WPB][JB * SL * B2];
		const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
		// xmap (the grid is then a multiple of 8): workgroup b runs on XCD b % 8; XCD x takes the contiguous range
		// [x * per, (x + 1) * per) of the slice groups, per = gridDim.x / 8; groups past the last slice do nothing
		const int64_t per = gridDim.x / PFX_XCDS;
		const int64_t blk = xmap ? (int64_t)(blockIdx.x % PFX_XCDS) * per + blockIdx.x / PFX_XCDS : (int64_t)blockIdx.x;
		const int64_t s = blk * WPB + w;
		const bool active = s < n_slices;
		const int64_t base = active ? slice_base[s] : 0;
		const int len = active ? (int)(slice_base[s + 1] - base) : 0;
		const int g_mine = active ? perm[s * SL + lane] : -1;
		const int64_t b0_mine = g_mine >= 0 ? node_ptr[g_mine] : 0;
		const int deg_mine = g_mine >= 0 ? (int)(node_ptr[g_mine + 1] - b0_mine) : 0;
		V *o = out + base * (SL * B2);
		for (int j = 0; j < len; j += JB)
		{
#pragma unroll
			for (int it = 0; it < RUN; ++it)
			{
				const int t = it * 64 + lane;
				const int l = t / RUN, r = t - l * RUN; // row of the slice, position in its run
				const int jj = r / B2, c = r - jj * B2;
				const int64_t b0 = __shfl(b0_mine, l);
				const int deg = __shfl(deg_mine, l);
				const double v = j + jj < deg ? blocks[(b0 + j) * B2 + r] : 0.0;
				tile[w][jj * (SL * B2) + entry_offset<BS, V>(c, l)] = (V)v;
			}
			// one wave reads back what it wrote itself: LDS operations of a wave complete in order
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			const int nj = len - j < JB ? len - j : JB; // slots of this pass that exist in the slice
			for (int it = 0; it < nj * B2; ++it)
				__builtin_nontemporal_store(tile[w][it * 64 + lane], o + (int64_t)j * (SL * B2) + it * 64 + lane);
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		}
	}

	// The same conversion for 3 x 3 blocks without the LDS transpose: a lane IS a row of the slice.  It walks its row's blocks
	// -- 72 contiguous bytes each, so consecutive loads of a lane run through consecutive bytes and every 128-byte line is
	// used up by the lane that opened it within a few instructions -- and writes its nine entries of a slot exactly where
	// entry_offset puts them: four 16-byte stores and one 8-byte store (two and one for float) that are contiguous
	// across the lanes.  The loads are divergent (one line per lane) but ask for few distinct lines per byte.
	typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
	template <typename V, int U>
	__global__ __launch_bounds__(64) void k_sell_values_rows3(const int64_t *__restrict__ node_ptr, const int32_t *__restrict__ perm,
															  const int64_t *__restrict__ slice_base, int64_t n_slices,
															  const double *__restrict__ blocks, V *__restrict__ out)
	{
		const int lane = threadIdx.x;
		// workgroup b runs on XCD b % 8; XCD x takes the contiguous range [x * per, (x + 1) * per) of the slices
		const int64_t per = gridDim.x / PFX_XCDS;
		const int64_t s = (int64_t)(blockIdx.x % PFX_XCDS) * per + blockIdx.x / PFX_XCDS;
		if (s >= n_slices)
			return;
		const int64_t base = slice_base[s];
		const int len = (int)(slice_base[s + 1] - base);
		const int g = perm[s * SL + lane];
		const int64_t b0 = g >= 0 ? node_ptr[g] : 0;
		const int deg = g >= 0 ? (int)(node_ptr[g + 1] - b0) : 0;
		const double *src = blocks + b0 * 9;
		V *o = out + base * (SL * 9);
		static_assert(U % 2 == 0, "whole 16-byte pieces");
		for (int j = 0; j < len; j += U)
		{
			double v[9 * U];
			if (j + U <= deg)
			{
				const d2u *p = reinterpret_cast<const d2u *>(src + (int64_t)j * 9);
#pragma unroll
				for (int q = 0; q < 9 * U / 2; ++q)
				{
					const d2u t = p[q];
					v[2 * q] = t.x, v[2 * q + 1] = t.y;
				}
			}
			else
			{
#pragma unroll
				for (int u = 0; u < U; ++u)
				{
					const bool have = j + u < deg;
#pragma unroll
					for (int q = 0; q < 9; ++q)
						v[9 * u + q] = have ? src[(int64_t)(j + u) * 9 + q] : 0.0;
				}
			}
#pragma unroll
			for (int u = 0; u < U; ++u)
				if (j + u < len)
				{
					V *os = o + (int64_t)(j + u) * (SL * 9);
					const double *w = v + 9 * u;
					if constexpr (sizeof(V) == 8)
					{
						typedef double d2 __attribute__((ext_vector_type(2)));
						d2 *od = reinterpret_cast<d2 *>(os);
#pragma unroll
						for (int q = 0; q < 4; ++q)
						{
							d2 t;
							t.x = w[2 * q], t.y = w[2 * q + 1];
							__builtin_nontemporal_store(t, od + q * 64 + lane);
						}
					}
					else
					{
						typedef float f4 __attribute__((ext_vector_type(4)));
						f4 *of = reinterpret_cast<f4 *>(os);
#pragma unroll
						for (int q = 0; q < 2; ++q)
						{
							f4 t;
							t.x = (float)w[4 * q], t.y = (float)w[4 * q + 1], t.z = (float)w[4 * q + 2], t.w = (float)w[4 * q + 3];
							__builtin_nontemporal_store(t, of + q * 64 + lane);
						}
					}
					__builtin_nontemporal_store((V)w[8], os + 512 + lane);
				}
		}
	}

	// flag[s] = 1 if slice s reads an entry of x that this rank does not own (column node >= n_owned)
	__global__ __launch_bounds__(256) void k_sell_classify(const int64_t *__restrict__ slice_base, const int32_t *__restrict__ cols,
															int64_t n_slices, int64_t n_owned, uint8_t *__restrict__ flag)
	{
		const int lane = threadIdx.x & 63;
		const int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
		if (s >= n_slices)
			return;
		const int64_t base = slice_base[s];
		const int len = (int)(slice_base[s + 1] - base);
		int any = 0;
		for (int j = 0; j < len; ++j)
			any |= cols[(base + j) * SL + lane] >= n_owned;
		if (__ballot(any) != 0ull && lane == 0)
			flag[s] = 1;
	}

	__device__ __forceinline__ double wave_sum(double v)
	{
		for (int off = 32; off > 0; off >>= 1)
			v += __shfl_down(v, off, 64);
		return v;
	}

	// y = (A + shift I) x on owned rows (constrained rows: y = x); optional per-block partial of x.y.  One slice per single-wave block (measured
	// faster than four waves per block); blocks b and b+8 run on the same XCD, so each XCD walks one contiguous range of the slice list (x gathers stay
	// in its L2), the ranges cut so that they hold equal numbers of stored slots (pfx_xcd_split).
	// SLICEOUT: the results go to `y` in slice order, y[(s * BS + m) * 64 + lane] -- one full-width store per row component --
	// and k_sell_unpermute brings them into row order afterwards.  Stored directly, a lane's rows sit 24 bytes each in a
	// cache line of their own (the rows of a slice are sorted by degree, not neighbours): every line is fetched to be
	// merged and written back, ten times the useful bytes, 16 % of the kernel.
	// V: the type the matrix values are stored in.  double: the solver's matrix.  float: the copy the AMG preconditioner
	// multiplies with (half the bytes per product; the vectors and the accumulation stay double).
	template <int BS, bool DOT, bool SLICEOUT, typename V>
	__global__ __launch_bounds__(64) void k_sell_spmv(const int64_t *__restrict__ slice_base, const int32_t *__restrict__ perm,
														const int32_t *__restrict__ cols, const V *__restrict__ values,
														const int32_t *__restrict__ order, int first, pfx_xcd_split split,
														const uint8_t *__restrict__ fixed, double shift, const double *__restrict__ x,
														double *__restrict__ y, double *__restrict__ part, const int *__restrict__ done)
	{
		if (DOT && *done)
			return;
		constexpr int B2 = BS * BS;
		const int lane = threadIdx.x;
		// XCD x walks the positions split.first[x] .. split.first[x+1]-1 of the slice list (equal numbers of stored slots)
		const int xcd = blockIdx.x % PFX_XCDS;
		const int vb = split.first[xcd] + (int)(blockIdx.x / PFX_XCDS);
		if (vb >= split.first[xcd + 1])
			return;
		// position first+vb of the slice list (interior slices first, then those that read halo entries of x)
		const int64_t s = order ? order[first + vb] : first + vb;
		double dot = 0.0;
		{
			const int64_t base = slice_base[s];
			const int len = (int)(slice_base[s + 1] - base);
			const V *v = values + base * (SL * B2);
			const int32_t *c = cols + base * SL + lane;
			double a0 = 0, a1 = 0, a2 = 0;
			// what the end of the slice needs -- the row this lane owns, its entries of x and its constraint flags -- is
			// requested now: asked for after the last slot, the chain row -> x -> store is two memory latencies during
			// which the wave holds its place and streams nothing (16 % of the kernel)
			const int g = perm[s * SL + lane];
			double xr[BS];
			bool fx[BS];
#pragma unroll
			for (int m = 0; m < BS; ++m)
			{
				xr[m] = g >= 0 ? x[(int64_t)g * BS + m] : 0.0;
				fx[m] = (g >= 0 && fixed) ? fixed[(int64_t)g * BS + m] != 0 : false;
			}
			if constexpr (BS == 3 && sizeof(V) == 4)
			{
				// single-precision values: two 16-byte loads (entries 0-3, 4-7) and one 4-byte load per slot, ten slots per
				// batch (about the bytes in flight of the double-precision batches below)
				typedef float f4 __attribute__((ext_vector_type(4)));
#ifndef PFX_SPMV_BATCH32
#define PFX_SPMV_BATCH32 10
#endif
				constexpr int U = PFX_SPMV_BATCH32;
				int cn[U];
#pragma unroll
				for (int u = 0; u < U; ++u)
					cn[u] = u < len ? __builtin_nontemporal_load(c + u * SL) : 0;
				int j = 0;
				for (; j + U <= len; j += U)
				{
					f4 p0[U], p1[U];
					float p4[U];
					double x0[U], x1[U], x2[U];
#pragma unroll
					for (int u = 0; u < U; ++u)
					{
						const float *vj = v + (int64_t)(j + u) * (SL * 9);
						const f4 *vp = reinterpret_cast<const f4 *>(vj) + lane;
						p0[u] = __builtin_nontemporal_load(vp);
						p1[u] = __builtin_nontemporal_load(vp + 64);
						p4[u] = __builtin_nontemporal_load(vj + 512 + lane);
					}
#pragma unroll
					for (int u = 0; u < U; ++u)
					{
						const int64_t cc = (int64_t)cn[u] * 3;
						x0[u] = x[cc], x1[u] = x[cc + 1], x2[u] = x[cc + 2];
					}
#pragma unroll
					for (int u = 0; u < U; ++u)
						cn[u] = j + U + u < len ? __builtin_nontemporal_load(c + (j + U + u) * SL) : 0;
#pragma unroll
					for (int u = 0; u < U; ++u)
					{
						a0 += (double)p0[u].x * x0[u] + (double)p0[u].y * x1[u] + (double)p0[u].z * x2[u];
						a1 += (double)p0[u].w * x0[u] + (double)p1[u].x * x1[u] + (double)p1[u].y * x2[u];
						a2 += (double)p1[u].z * x0[u] + (double)p1[u].w * x1[u] + (double)p4[u] * x2[u];
					}
				}
				if (j < len)
				{
					f4 p0[U], p1[U];
					float p4[U];
					double x0[U], x1[U], x2[U];
#pragma unroll
					for (int u = 0; u < U - 1; ++u)
						if (j + u < len)
						{
							const float *vj = v + (int64_t)(j + u) * (SL * 9);
							const f4 *vp = reinterpret_cast<const f4 *>(vj) + lane;
							p0[u] = __builtin_nontemporal_load(vp);
							p1[u] = __builtin_nontemporal_load(vp + 64);
							p4[u] = __builtin_nontemporal_load(vj + 512 + lane);
							const int64_t cc = (int64_t)cn[u] * 3;
							x0[u] = x[cc], x1[u] = x[cc + 1], x2[u] = x[cc + 2];
						}
#pragma unroll
					for (int u = 0; u < U - 1; ++u)
						if (j + u < len)
						{
							a0 += (double)p0[u].x * x0[u] + (double)p0[u].y * x1[u] + (double)p0[u].z * x2[u];
							a1 += (double)p0[u].w * x0[u] + (double)p1[u].x * x1[u] + (double)p1[u].y * x2[u];
							a2 += (double)p1[u].z * x0[u] + (double)p1[u].w * x1[u] + (double)p4[u] * x2[u];
						}
				}
			}
			else if constexpr (BS == 3)
			{
				// matrix values and column ids are read once per product: non-temporal loads keep them from evicting
				// the gathered x entries out of the L2
				typedef double d2 __attribute__((ext_vector_type(2)));
#ifndef PFX_SPMV_BATCH
#define PFX_SPMV_BATCH 6
#endif
				// Batches of U slots: all loads of a batch are issued back to back -- the values, the x entries (their
				// columns were fetched one batch ahead, so their addresses are known) and the next batch's columns -- and
				// only then waited for.  (Left to the compiler, a loop over single slots was scheduled with full waits in
				// the middle of the unrolled body: 5.6 TB/s on the value stream alone where a plain read reaches 7.)
				constexpr int U = PFX_SPMV_BATCH;
				int cn[U];
#pragma unroll
				for (int u = 0; u < U; ++u)
					cn[u] = u < len ? __builtin_nontemporal_load(c + u * SL) : 0;
				int j = 0;
				for (; j + U <= len; j += U)
				{
					d2 p0[U], p1[U], p2[U], p3[U];
					double p4[U], x0[U], x1[U], x2[U];
#pragma unroll
					for (int u = 0; u < U; ++u)
					{
						const double *vj = v + (int64_t)(j + u) * (SL * 9);
						const d2 *vp = reinterpret_cast<const d2 *>(vj) + lane;
						p0[u] = __builtin_nontemporal_load(vp);
						p1[u] = __builtin_nontemporal_load(vp + 64);
						p2[u] = __builtin_nontemporal_load(vp + 128);
						p3[u] = __builtin_nontemporal_load(vp + 192);
						p4[u] = __builtin_nontemporal_load(vj + 512 + lane);
					}
#pragma unroll
					for (int u = 0; u < U; ++u)
					{
						const int64_t cc = (int64_t)cn[u] * 3;
						x0[u] = x[cc], x1[u] = x[cc + 1], x2[u] = x[cc + 2];
					}
#pragma unroll
					for (int u = 0; u < U; ++u)
						cn[u] = j + U + u < len ? __builtin_nontemporal_load(c + (j + U + u) * SL) : 0;
#pragma unroll
					for (int u = 0; u < U; ++u)
					{
						a0 += p0[u].x * x0[u] + p0[u].y * x1[u] + p1[u].x * x2[u];
						a1 += p1[u].y * x0[u] + p2[u].x * x1[u] + p2[u].y * x2[u];
						a2 += p3[u].x * x0[u] + p3[u].y * x1[u] + p4[u] * x2[u];
					}
				}
				// the last len % U slots (their columns are in cn): one more batch, loads first, every slot guarded by the
				// (wave-uniform) bound -- slot by slot they would each wait out a full memory latency
				if (j < len)
				{
					d2 p0[U], p1[U], p2[U], p3[U];
					double p4[U], x0[U], x1[U], x2[U];
#pragma unroll
					for (int u = 0; u < U - 1; ++u)
						if (j + u < len)
						{
							const double *vj = v + (int64_t)(j + u) * (SL * 9);
							const d2 *vp = reinterpret_cast<const d2 *>(vj) + lane;
							p0[u] = __builtin_nontemporal_load(vp);
							p1[u] = __builtin_nontemporal_load(vp + 64);
							p2[u] = __builtin_nontemporal_load(vp + 128);
							p3[u] = __builtin_nontemporal_load(vp + 192);
							p4[u] = __builtin_nontemporal_load(vj + 512 + lane);
							const int64_t cc = (int64_t)cn[u] * 3;
							x0[u] = x[cc], x1[u] = x[cc + 1], x2[u] = x[cc + 2];
						}
#pragma unroll
					for (int u = 0; u < U - 1; ++u)
						if (j + u < len)
						{
							a0 += p0[u].x * x0[u] + p0[u].y * x1[u] + p1[u].x * x2[u];
							a1 += p1[u].y * x0[u] + p2[u].x * x1[u] + p2[u].y * x2[u];
							a2 += p3[u].x * x0[u] + p3[u].y * x1[u] + p4[u] * x2[u];
						}
				}
			}
			else if constexpr (BS == 2)
			{
				// 2-D problems: four entries per slot in 16-byte groups (entry_offset): two loads of two doubles, or one of
				// four floats, per lane and slot; the two entries of x are one 16-byte load
				typedef double d2 __attribute__((ext_vector_type(2)));
				typedef double d2x __attribute__((ext_vector_type(2), aligned(8))); // x is the caller's: only its 8-byte alignment is known
				typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll 4
				for (int j = 0; j < len; ++j)
				{
					const V *vj = v + (int64_t)j * (SL * 4);
					const int64_t cc = (int64_t)__builtin_nontemporal_load(c + j * SL) * 2;
					const d2x xv = *reinterpret_cast<const d2x *>(x + cc);
					double m0, m1, m2, m3;
					if constexpr (sizeof(V) == 8)
					{
						const d2 p0 = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(vj) + lane);
						const d2 p1 = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(vj + 128) + lane);
						m0 = p0.x, m1 = p0.y, m2 = p1.x, m3 = p1.y;
					}
					else
					{
						const f4 p = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(vj) + lane);
						m0 = p.x, m1 = p.y, m2 = p.z, m3 = p.w;
					}
					a0 += m0 * xv.x + m1 * xv.y;
					a1 += m2 * xv.x + m3 * xv.y;
				}
			}
			else
			{
#pragma unroll 8
				for (int j = 0; j < len; ++j)
					a0 += (double)__builtin_nontemporal_load(v + (int64_t)j * SL + lane) * x[__builtin_nontemporal_load(c + j * SL)];
			}
			{
				const double acc[3] = {a0, a1, a2};
#pragma unroll
				for (int m = 0; m < BS; ++m)
				{
					const double yr = g < 0 ? 0.0 : fx[m] ? xr[m] : acc[m] + shift * xr[m];
					if (SLICEOUT)
						y[(s * BS + m) * SL + lane] = yr;
					else if (g >= 0)
						y[(int64_t)g * BS + m] = yr;
					if (DOT)
						dot += xr[m] * yr;
				}
			}
		}
		if (DOT)
		{
			dot = wave_sum(dot);
			if (lane == 0)
				part[first + vb] = dot;
		}
	}

	// pos[g] = position (slice * 64 + lane) of row g in the slice order
	__global__ void k_sell_pos(const int32_t *__restrict__ perm, int64_t n_pad, int32_t *__restrict__ pos)
	{
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_pad; k += (int64_t)gridDim.x * blockDim.x)
			if (perm[k] >= 0)
				pos[perm[k]] = (int32_t)k;
	}

	// y[g * BS + m] = ys[(slice * BS + m) * 64 + lane]: consecutive threads write consecutive rows; the reads stay inside
	// the window of rows the slices were sorted in (L2)
	template <int BS>
	__global__ __launch_bounds__(256) void k_sell_unpermute(const int32_t *__restrict__ pos, const double *__restrict__ ys, int64_t n_rows, double *__restrict__ y,
															const int *__restrict__ done)
	{
		if (done && *done) // the product was skipped (the solve has converged): nothing new to bring over
			return;
		for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * blockDim.x)
		{
			const int64_t g = r / BS;
			const int m = (int)(r - g * BS);
			const int64_t k = pos[g];
			y[r] = ys[((k >> 6) * BS + m) * SL + (k & 63)];
		}
	}

} // namespace

void pfx_sell_release(pfx_ctx *ctx)
{
	free_dev(ctx, ctx->d_sell_pos);
	free_dev(ctx, ctx->d_sell_ys);
	free_dev(ctx, ctx->d_sell_perm);
	free_dev(ctx, ctx->d_sell_base);
	free_dev(ctx, ctx->d_sell_cols);
	free_dev(ctx, ctx->d_sell_vals);
	free_dev(ctx, ctx->d_sell_vals32);
	ctx->sell32_gen = -1;
	free_dev(ctx, ctx->d_sell_order);
	ctx->sell_slices = 0;
	ctx->sell_interior = 0;
	ctx->sell_owned = -1;
	ctx->sell_gen = -1;
	ctx->sell_placed = false;
	ctx->sell32_placed = false;
}

// structure: once per (pattern, owned range)
static pfx_status sell_build(pfx_ctx *ctx)
{
	pfx_sell_release(ctx);
	hipStream_t st = ctx->stream;
	const int64_t n_owned = ctx->n_owned;
	int sigma = 32768;
	if (const char *e = getenv("PFX_SELL_SIGMA"))
		sigma = atoi(e) > 0 ? (atoi(e) + SL - 1) / SL * SL : sigma;
	const int64_t n_slices = (n_owned + SL - 1) / SL;
	const int64_t n_pad = n_slices * SL;
	pfx_tmp<uint64_t> ka, kb;
	pfx_tmp<uint32_t> va, vb;
	pfx_tmp<int64_t> len;
	pfx_tmp<char> tmp;
	size_t tb = 0;
	PFX_HIP(ka.alloc((size_t)(n_owned + 1)));
	PFX_HIP(kb.alloc((size_t)(n_owned + 1)));
	PFX_HIP(va.alloc((size_t)(n_owned + 1)));
	PFX_HIP(vb.alloc((size_t)(n_owned + 1)));
	PFX_HIP(len.alloc((size_t)(n_slices + 1)));
	PFX_HIP(hipMemsetAsync(len.p, 0, sizeof(int64_t) * (size_t)(n_slices + 1), st));
	hipLaunchKernelGGL(k_sell_keys, dim3(grid_for(n_owned)), dim3(256), 0, st, ctx->d_node_ptr, n_owned, sigma, ka.p, va.p);
	PFX_HIP(rocprim::radix_sort_pairs(nullptr, tb, ka.p, kb.p, va.p, vb.p, (size_t)n_owned, 0, 64, st));
	PFX_HIP(tmp.alloc(tb));
	PFX_HIP(rocprim::radix_sort_pairs(tmp.p, tb, ka.p, kb.p, va.p, vb.p, (size_t)n_owned, 0, 64, st));
	if (pfx_alloc(ctx, &ctx->d_sell_perm, n_pad) || pfx_alloc(ctx, &ctx->d_sell_base, n_slices + 1))
		return PFX_ERR_HIP;
	hipLaunchKernelGGL(k_sell_perm, dim3(grid_for(n_pad)), dim3(256), 0, st, vb.p, n_owned, n_pad, ctx->d_node_ptr, ctx->d_sell_perm, len.p);
	if (pfx_alloc(ctx, &ctx->d_sell_pos, n_owned) || pfx_alloc(ctx, &ctx->d_sell_ys, n_pad * ctx->bs))
		return PFX_ERR_HIP;
	hipLaunchKernelGGL(k_sell_pos, dim3(grid_for(n_pad)), dim3(256), 0, st, ctx->d_sell_perm, n_pad, ctx->d_sell_pos);
	PFX_HIP(hipStreamSynchronize(st));
	tb = 0;
	PFX_HIP(rocprim::exclusive_scan(nullptr, tb, len.p, ctx->d_sell_base, (int64_t)0, (size_t)(n_slices + 1), rocprim::plus<int64_t>(), st));
	PFX_HIP(tmp.alloc(tb));
	PFX_HIP(rocprim::exclusive_scan(tmp.p, tb, len.p, ctx->d_sell_base, (int64_t)0, (size_t)(n_slices + 1), rocprim::plus<int64_t>(), st));
	int64_t total = 0;
	PFX_HIP(hipMemcpyAsync(&total, ctx->d_sell_base + n_slices, sizeof(int64_t), hipMemcpyDeviceToHost, st));
	PFX_HIP(hipStreamSynchronize(st));
	tmp.reset(), ka.reset(), kb.reset(), va.reset(), vb.reset(), len.reset();
	ctx->sell_rows = total;
	ctx->sell_slices = n_slices;
	const int b2 = ctx->bs * ctx->bs;
	(void)b2; // the value arrays (double for the solver, float for the preconditioner) are made by whoever first needs them
	if (pfx_alloc(ctx, &ctx->d_sell_cols, total * SL))
		return PFX_ERR_HIP;
	int64_t blocks = (n_slices + 3) / 4;
	if (blocks > 256 * 32)
		blocks = 256 * 32;
	if (blocks < 1)
		blocks = 1;
	hipLaunchKernelGGL(k_sell_cols, dim3((unsigned)blocks), dim3(256), 0, st, ctx->d_node_ptr, ctx->d_node_col, ctx->d_sell_perm, ctx->d_sell_base, n_slices, ctx->d_sell_cols);
	PFX_HIP(hipGetLastError());
	// slice list: interior slices first, then the slices that need halo entries (distributed solves overlap the halo
	// exchange with the interior part).  A mesh without halo keeps the natural order (no list).
	free_dev(ctx, ctx->d_sell_order);
	ctx->sell_interior = n_slices;
	if (n_owned < ctx->n_nodes && n_slices > 0)
	{
		pfx_tmp<uint8_t> d_flag;
		PFX_HIP(d_flag.alloc((size_t)n_slices));
		PFX_HIP(hipMemsetAsync(d_flag.p, 0, (size_t)n_slices, st));
		hipLaunchKernelGGL(k_sell_classify, dim3((unsigned)((n_slices + 3) / 4)), dim3(256), 0, st, ctx->d_sell_base, ctx->d_sell_cols, n_slices, n_owned, d_flag.p);
		std::vector<uint8_t> flag((size_t)n_slices);
		PFX_HIP(hipMemcpyAsync(flag.data(), d_flag.p, (size_t)n_slices, hipMemcpyDeviceToHost, st));
		PFX_HIP(hipStreamSynchronize(st));
		d_flag.reset();
		std::vector<int32_t> order;
		order.reserve((size_t)n_slices);
		for (int64_t k = 0; k < n_slices; ++k)
			if (!flag[(size_t)k])
				order.push_back((int32_t)k);
		ctx->sell_interior = (int64_t)order.size();
		for (int64_t k = 0; k < n_slices; ++k)
			if (flag[(size_t)k])
				order.push_back((int32_t)k);
		if (pfx_alloc(ctx, &ctx->d_sell_order, n_slices))
			return PFX_ERR_HIP;
		PFX_HIP(hipMemcpyAsync(ctx->d_sell_order, order.data(), sizeof(int32_t) * (size_t)n_slices, hipMemcpyHostToDevice, st));
		PFX_HIP(hipStreamSynchronize(st));
	}
	// XCD ranges of the slice list for the three phases, cut by the slices' lengths (+1 per slice for its fixed cost)
	{
		std::vector<int64_t> base((size_t)n_slices + 1);
		PFX_HIP(hipMemcpyAsync(base.data(), ctx->d_sell_base, sizeof(int64_t) * base.size(), hipMemcpyDeviceToHost, st));
		std::vector<int32_t> order;
		if (ctx->d_sell_order)
		{
			order.resize((size_t)n_slices);
			PFX_HIP(hipMemcpyAsync(order.data(), ctx->d_sell_order, sizeof(int32_t) * order.size(), hipMemcpyDeviceToHost, st));
		}
		PFX_HIP(hipStreamSynchronize(st));
		std::vector<double> pre((size_t)n_slices + 1, 0.0); // work of the positions before each position of the list
		for (int64_t k = 0; k < n_slices; ++k)
		{
			const int64_t sl = order.empty() ? k : order[(size_t)k];
			pre[(size_t)k + 1] = pre[(size_t)k] + (double)(base[(size_t)sl + 1] - base[(size_t)sl]) + 1.0;
		}
		const int64_t ni = ctx->sell_interior;
		ctx->sell_split[PFX_SPMV_ALL] = pfx_make_xcd_split(n_slices, [&](int64_t k) { return pre[(size_t)k]; });
		ctx->sell_split[PFX_SPMV_INTERIOR] = pfx_make_xcd_split(ni, [&](int64_t k) { return pre[(size_t)k]; });
		ctx->sell_split[PFX_SPMV_BOUNDARY] = pfx_make_xcd_split(n_slices - ni, [&](int64_t k) { return pre[(size_t)(ni + k)] - pre[(size_t)ni]; });
	}
	ctx->sell_owned = n_owned;
	ctx->sell_gen = -1;
	ctx->sell32_gen = -1;
	return PFX_OK;
}

// workgroups of the conversion kernel for n slice groups; with the XCD-contiguous mapping the grid is rounded up to a
// multiple of the number of XCDs (the extra groups lie past the last slice and do nothing)
static int64_t convert_grid(int64_t n, int xmap)
{
	if (n < 1)
		n = 1;
	return xmap ? (n + PFX_XCDS - 1) / PFX_XCDS * PFX_XCDS : n;
}

// makes the solver copy of the matrix current (structure and values)
// values of the node-block array -> slices (one wave per slice)
template <typename V>
static pfx_status sell_convert(pfx_ctx *ctx, V *out)
{
#define PFX_CONVERT(BS_, JB_, WPB_)                                                                                                                         \
	hipLaunchKernelGGL((k_sell_values<BS_, JB_, WPB_, V>), dim3((unsigned)convert_grid((ctx->sell_slices + WPB_ - 1) / WPB_, xmap)), \
					   dim3(64 * WPB_), 0, ctx->stream, ctx->d_node_ptr, ctx->d_sell_perm, ctx->d_sell_base, ctx->sell_slices, ctx->d_values, out, xmap)
	const int xmap = 0; // (the LDS-transposing kernel serves block sizes 1 and 2, in round-robin order)
	if (ctx->bs == 1)
		PFX_CONVERT(1, 4, 4);
	else if (ctx->bs == 3)
		// a lane per row, four slots per pass.  10 M P2 tets, double: the LDS transpose took 14.76 ms with one slot per pass
		// and round-robin slices (74 GB read for 28 GB of blocks: a 128-byte line holds 1.8 blocks and most lines were
		// fetched twice), 12.8 ms with two slots per pass and contiguous slice ranges per XCD (52 GB read); this kernel
		// 11.5 / 11.0 / 10.96 ms with 2 / 4 / 8 slots per pass (41 / 35 / 34 GB read)
		hipLaunchKernelGGL((k_sell_values_rows3<V, 4>), dim3((unsigned)convert_grid(ctx->sell_slices, 1)), dim3(64), 0, ctx->stream, ctx->d_node_ptr, ctx->d_sell_perm,
						   ctx->d_sell_base, ctx->sell_slices, ctx->d_values, out);
	else if (ctx->bs == 2)
		PFX_CONVERT(2, 2, 4);
	else
		PFX_FAIL(PFX_ERR_UNSUPPORTED, "block size must be 1, 2 or 3");
#undef PFX_CONVERT
	PFX_HIP(hipGetLastError());
	return PFX_OK;
}

// Placement search for a freshly allocated value array.  The speed of the product depends on WHICH physical pages the
// value stream lies in: re-allocating nothing but this array moves a 30 GB product between 4.85 and 5.35 ms at one and
// the same virtual address (the single-precision copy: 2.7 to 3.3 ms), and the allocator hands out a different set of pages
// each time while the previous array is still held.  So a solve that has proven long (pfx_sell_place, called by the CG loops
// after their first 192 iterations) tries a few placements -- allocate a candidate beside the current array, fill it, time
// three products, keep the faster, free the other -- and keeps the best (or the first that streams the values at 0.74 of the
// HBM peak).  Costs one conversion and four products per try, once per pattern.  Not done at allocation: the freed candidates
// keep the driver's allocator busy for seconds (about one second per 12 GB freed), which a CG loop in flight does not
// notice and a set-up phase that allocates right afterwards does (an AMG set-up after an eager search took 3 - 6 s instead
// of 0.4 s).  Skipped for arrays below PFX_PLACEMENT_MIN_BYTES (PFX_PLACEMENT_MIN_MB in the environment overrides it),
// when the device has no room for a second copy, and with PFX_PLACEMENT_TRIES=0 in the environment (default 6).
#ifndef PFX_PLACEMENT_MIN_BYTES
#define PFX_PLACEMENT_MIN_BYTES ((size_t)1 << 30)
#endif
static float good_ms_of(size_t bytes) { return 3.0f * (float)((double)bytes / (0.74 * 8.0e9)); }
template <typename V>
static pfx_status sell_place(pfx_ctx *ctx, V *&array, int64_t count, bool fp32)
{
	const char *env = getenv("PFX_PLACEMENT_TRIES");
	const int tries = env ? atoi(env) : 6;
	const size_t bytes = sizeof(V) * (size_t)count;
	const char *env_min = getenv("PFX_PLACEMENT_MIN_MB");
	const size_t min_bytes = env_min ? (size_t)atoll(env_min) << 20 : PFX_PLACEMENT_MIN_BYTES;
	if (tries <= 0 || bytes < min_bytes || ctx->n_owned < 65536)
		return PFX_OK;
	const int64_t n = ctx->n_dofs();
	size_t free_b = 0, total_b = 0;
	if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < bytes + sizeof(double) * (size_t)n + ((size_t)2 << 30))
	{
		(void)hipGetLastError();
		return PFX_OK;
	}
	pfx_scoped_timer tm(ctx, "placement");
	// the trial products and conversions are not the caller's: kept out of the per-family timings
	struct timing_off
	{
		pfx_ctx *c;
		bool was;
		explicit timing_off(pfx_ctx *ctx_) : c(ctx_), was(ctx_->timing) { c->timing = false; }
		~timing_off() { c->timing = was; }
	} quiet(ctx);
	hipStream_t st = ctx->stream;
	pfx_tmp<double> xv;
	PFX_HIP(xv.alloc((size_t)n));
	PFX_HIP(hipMemsetAsync(xv.p, 0, sizeof(double) * (size_t)n, st));
	hipEvent_t e0, e1;
	PFX_HIP(hipEventCreate(&e0));
	PFX_HIP(hipEventCreate(&e1));
	pfx_status rc = PFX_OK;
	auto measure = [&](float *ms) -> pfx_status {
		bool sliced = false; // the products stay in the slice-ordered buffer: no second vector needed
		pfx_status r = pfx_sell_spmv(ctx, xv.p, nullptr, nullptr, nullptr, PFX_SPMV_ALL, &sliced, fp32);
		if (r)
			return r;
		PFX_HIP(hipEventRecord(e0, st));
		for (int k = 0; k < 3 && !r; ++k)
			r = pfx_sell_spmv(ctx, xv.p, nullptr, nullptr, nullptr, PFX_SPMV_ALL, &sliced, fp32);
		if (r)
			return r;
		PFX_HIP(hipEventRecord(e1, st));
		PFX_HIP(hipEventSynchronize(e1));
		PFX_HIP(hipEventElapsedTime(ms, e0, e1));
		return PFX_OK;
	};
	static const bool log_tries = getenv("PFX_PLACEMENT_LOG") != nullptr; // one line per candidate on stderr
	float best = 0;
	rc = measure(&best);
	if (log_tries)
		fprintf(stderr, "[polyfem_amd] placement search (%s, %.1f GB): in place %.3f ms per product, stop below %.3f\n", fp32 ? "float" : "double", bytes / 1e9, best / 3, good_ms_of(bytes) / 3);
	// three products with the value stream alone at 0.74 of the HBM peak (0.80 with the index and vector bytes of the 3 x 3
	// double-precision format): as good as it gets, stop looking
	const float good_ms = good_ms_of(bytes);
	for (int k = 0; k < tries && !rc && best > good_ms; ++k)
	{
		V *cand = nullptr;
		if (hipMalloc((void **)&cand, bytes) != hipSuccess)
		{
			(void)hipGetLastError(); // the earlier candidates have not come back to the allocator yet: enough tried
			break;
		}
		V *held = array;
		array = cand;
		float ms = 0;
		if ((rc = sell_convert<V>(ctx, cand)) || (rc = measure(&ms)))
		{
			array = held;
			(void)hipFree(cand);
			break;
		}
		if (log_tries)
			fprintf(stderr, "[polyfem_amd]   candidate %d: %.3f ms per product (%s)\n", k + 1, ms / 3, ms < best ? "kept" : "dropped");
		if (ms < best)
		{
			best = ms;
			// the candidate takes the place of the held array in the context's allocation record
			auto it = ctx->alloc_bytes.find((const void *)held);
			const int64_t recorded = it != ctx->alloc_bytes.end() ? it->second : (int64_t)bytes;
			if (it != ctx->alloc_bytes.end())
				ctx->alloc_bytes.erase(it);
			ctx->alloc_bytes[(const void *)cand] = recorded;
			(void)hipFree(held);
		}
		else
		{
			array = held;
			(void)hipFree(cand);
		}
	}
	(void)hipEventDestroy(e0);
	(void)hipEventDestroy(e1);
	return rc;
}

pfx_status pfx_sell_refresh(pfx_ctx *ctx)
{
	if (ctx->sell_owned != ctx->n_owned || !ctx->d_sell_perm)
	{
		pfx_status rc = sell_build(ctx);
		if (rc)
			return rc;
	}
	if (ctx->sell_gen != ctx->values_gen || !ctx->d_sell_vals)
	{
		const bool fresh = !ctx->d_sell_vals;
		const int64_t count = ctx->sell_rows * SL * ctx->bs * ctx->bs + 2;
		if (fresh && pfx_alloc(ctx, &ctx->d_sell_vals, count))
			return PFX_ERR_HIP;
		pfx_status rc;
		{
			pfx_scoped_timer tm(ctx, "sell_convert");
			if ((rc = sell_convert<double>(ctx, ctx->d_sell_vals)))
				return rc;
		}
		ctx->sell_gen = ctx->values_gen;
		if (fresh)
			ctx->sell_placed = false;
	}
	return PFX_OK;
}

// placement search for the solver's value array, once per allocation (see sell_place)
pfx_status pfx_sell_place(pfx_ctx *ctx)
{
	if (ctx->sell_placed || !ctx->d_sell_vals)
		return PFX_OK;
	ctx->sell_placed = true;
	return sell_place<double>(ctx, ctx->d_sell_vals, ctx->sell_rows * SL * ctx->bs * ctx->bs + 2, false);
}

// the same for the single-precision copy (only on request, pfx_tune_placement: the preconditioner's solves are short)
pfx_status pfx_sell_place32(pfx_ctx *ctx)
{
	if (ctx->sell32_placed || !ctx->d_sell_vals32)
		return PFX_OK;
	ctx->sell32_placed = true;
	return sell_place<float>(ctx, ctx->d_sell_vals32, ctx->sell_rows * SL * ctx->bs * ctx->bs + 4, true);
}

// the single-precision copy of the matrix the AMG preconditioner multiplies with (structure shared with the solver copy)
pfx_status pfx_sell_refresh32(pfx_ctx *ctx)
{
	if (ctx->sell_owned != ctx->n_owned || !ctx->d_sell_perm)
	{
		pfx_status rc = sell_build(ctx);
		if (rc)
			return rc;
	}
	if (ctx->sell32_gen != ctx->values_gen || !ctx->d_sell_vals32)
	{
		const int64_t count = ctx->sell_rows * SL * ctx->bs * ctx->bs + 4;
		if (!ctx->d_sell_vals32)
		{
			if (pfx_alloc(ctx, &ctx->d_sell_vals32, count))
				return PFX_ERR_HIP;
			ctx->sell32_placed = false;
		}
		pfx_status rc;
		{
			pfx_scoped_timer tm(ctx, "sell_convert32");
			if ((rc = sell_convert<float>(ctx, ctx->d_sell_vals32)))
				return rc;
		}
		ctx->sell32_gen = ctx->values_gen;
	}
	return PFX_OK;
}

// number of per-slice partial sums the SpMV writes
// the product left in slice order by pfx_sell_spmv(..., &sliced) -> row order in d_y
pfx_status pfx_sell_unpermute(pfx_ctx *ctx, double *d_y)
{
	const int64_t n = ctx->n_owned * ctx->bs;
	if (ctx->bs == 3)
		hipLaunchKernelGGL((k_sell_unpermute<3>), dim3(grid_for(n)), dim3(256), 0, ctx->stream, ctx->d_sell_pos, ctx->d_sell_ys, n, d_y, (const int *)nullptr);
	else if (ctx->bs == 1)
		hipLaunchKernelGGL((k_sell_unpermute<1>), dim3(grid_for(n)), dim3(256), 0, ctx->stream, ctx->d_sell_pos, ctx->d_sell_ys, n, d_y, (const int *)nullptr);
	else if (ctx->bs == 2)
		hipLaunchKernelGGL((k_sell_unpermute<2>), dim3(grid_for(n)), dim3(256), 0, ctx->stream, ctx->d_sell_pos, ctx->d_sell_ys, n, d_y, (const int *)nullptr);
	else
		PFX_FAIL(PFX_ERR_UNSUPPORTED, "block size must be 1, 2 or 3");
	PFX_HIP(hipGetLastError());
	return PFX_OK;
}

int pfx_sell_grid(const pfx_ctx *ctx)
{
	return (int)ctx->sell_slices;
}

// phase: PFX_SPMV_ALL, or the two halves of a distributed product: PFX_SPMV_INTERIOR (slices that read owned entries of
// x only) and PFX_SPMV_BOUNDARY (the rest, after the halo entries have arrived)
pfx_status pfx_sell_spmv(pfx_ctx *ctx, const double *d_x, double *d_y, double *part, const int *done, int phase, bool *sliced, bool fp32)
{
	if (fp32 ? !ctx->d_sell_vals32 : !ctx->d_sell_vals)
		PFX_FAIL(PFX_ERR_ARG, "the solver copy of the matrix has not been made (pfx_sell_refresh)");
	const uint8_t *fixed = ctx->n_dirichlet > 0 ? ctx->d_fixed : nullptr;
	int first = 0, count = (int)ctx->sell_slices;
	if (phase == PFX_SPMV_INTERIOR)
		count = (int)ctx->sell_interior;
	else if (phase == PFX_SPMV_BOUNDARY)
		first = (int)ctx->sell_interior, count = (int)(ctx->sell_slices - ctx->sell_interior);
	// a product goes through the slice-ordered buffer and, unless the caller reads it from there, the un-permute pass; small
	// matrices, where the second launch costs more than it saves, store their rows directly
	// (a half product can only stay in slice order: the un-permute pass covers all rows)
	// Where the product lives is a property of the matrix, not of the phase: a phase without slices (no boundary slices on
	// a rank without halo) reports the same place as the phase that wrote the rows.
	const bool sliceout = ctx->n_owned >= 65536 && (phase == PFX_SPMV_ALL || sliced);
	if (sliced)
		*sliced = sliceout;
	if (count <= 0)
		return PFX_OK;
	const pfx_xcd_split &split = ctx->sell_split[phase];
	const int grid = PFX_XCDS * split.widest; // one slice (one wave) per block
	const int32_t *order = ctx->d_sell_order;
	pfx_scoped_timer tm(ctx, phase == PFX_SPMV_BOUNDARY ? "spmv_boundary" : "spmv");
#define PFX_GO2(BS_, DOT_, V_, VALS_)                                                                                                                              \
	do                                                                                                                                                             \
	{                                                                                                                                                              \
		if (sliceout)                                                                                                                                              \
		{                                                                                                                                                          \
			hipLaunchKernelGGL((k_sell_spmv<BS_, DOT_, true, V_>), dim3(grid), dim3(64), 0, ctx->stream, ctx->d_sell_base, ctx->d_sell_perm, ctx->d_sell_cols,    \
							   VALS_, order, first, split, fixed, ctx->diag_shift, d_x, ctx->d_sell_ys, part, done);                                                   \
			if (!sliced)                                                                                                                                           \
				hipLaunchKernelGGL((k_sell_unpermute<BS_>), dim3(grid_for(ctx->n_owned * BS_)), dim3(256), 0, ctx->stream, ctx->d_sell_pos, ctx->d_sell_ys,        \
								   ctx->n_owned * BS_, d_y, done);                                                                                                       \
		}                                                                                                                                                          \
		else                                                                                                                                                       \
			hipLaunchKernelGGL((k_sell_spmv<BS_, DOT_, false, V_>), dim3(grid), dim3(64), 0, ctx->stream, ctx->d_sell_base, ctx->d_sell_perm, ctx->d_sell_cols,   \
							   VALS_, order, first, split, fixed, ctx->diag_shift, d_x, d_y, part, done);                                                              \
	} while (0)
#define PFX_GO(BS_, DOT_)                                          \
	do                                                             \
	{                                                              \
		if (fp32)                                                  \
			PFX_GO2(BS_, DOT_, float, ctx->d_sell_vals32);         \
		else                                                       \
			PFX_GO2(BS_, DOT_, double, ctx->d_sell_vals);          \
	} while (0)
	if (ctx->bs == 1)
	{
		if (part)
			PFX_GO(1, true);
		else
			PFX_GO(1, false);
	}
	else if (ctx->bs == 3)
	{
		if (part)
			PFX_GO(3, true);
		else
			PFX_GO(3, false);
	}
	else if (ctx->bs == 2)
	{
		if (part)
			PFX_GO(2, true);
		else
			PFX_GO(2, false);
	}
	else
		PFX_FAIL(PFX_ERR_UNSUPPORTED, "block size must be 1, 2 or 3");
#undef PFX_GO
#undef PFX_GO2
	return PFX_OK;
}
