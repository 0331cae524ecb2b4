// pfx_amg_graph.hip -- the graph work of the AMG set-up, on the device.
//
// The hierarchy of pfx_amg.hip is built from the vertex graph of the matrix alone: rounds of pairwise matching give the
// aggregates, the quotient graph by the aggregates is the next level's graph, and the list of fine edges behind every
// coarse edge is what the Galerkin product of that level walks.  At the sizes one MI355X holds (4e8 node blocks for
// 10 M quadratic tetrahedra) these are sorts and scans over a few GB: radix sorts of (coarse row, coarse column) keys
// with the fine edge as payload (a stable sort keeps the fine edges of a coarse edge in ascending order, which fixes
// the summation order of the Galerkin products), run heads -> coarse edges, and a handshake matching whose rounds are
// independent per vertex.  Everything is deterministic: ties are broken by a hash of the edge, integer weights are
// summed, no floating-point atomics.
#include "pfx_amg_graph.h"

#include <rocprim/rocprim.hpp>

namespace
{
	__device__ __forceinline__ uint64_t mix64(uint64_t z)
	{
		z += 0x9E3779B97F4A7C15ull;
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
		return z ^ (z >> 31);
	}

	// state: -2 never matched (excluded), -1 free, >= 0 the smaller vertex of its pair (or itself)
	__global__ void k_match_init(int64_t n, const uint8_t *__restrict__ excl, int32_t *__restrict__ state)
	{
		for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x)
			state[v] = (excl && excl[v]) ? -2 : -1;
	}

	// Proposal of every free vertex: the free neighbour that is best by (size of the cluster the pair would form, smaller
	// first; distance; weight, larger first; hash of the edge).  Every key is symmetric in the two ends, so the best edge
	// of a neighbourhood is the best for both of its ends: each round some proposals are mutual.
	__global__ __launch_bounds__(256) void k_match_propose(int64_t n, const int64_t *__restrict__ ptr, const int32_t *__restrict__ col, const uint32_t *__restrict__ w,
															 const double *__restrict__ pos, const int32_t *__restrict__ vsize, const int32_t *__restrict__ state, uint64_t seed,
															 int32_t *__restrict__ prop, unsigned long long *__restrict__ n_live)
	{
		unsigned long long live = 0;
		for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x)
		{
			if (state[v] != -1)
				continue; // prop[v] is only read for free vertices
			int32_t best = -1, bs = 0;
			uint32_t bw = 0;
			double bd = 0;
			uint64_t bh = 0;
			const int32_t sv = vsize ? vsize[v] : 1;
			double pv[3] = {0, 0, 0};
			if (pos)
				pv[0] = pos[3 * v], pv[1] = pos[3 * v + 1], pv[2] = pos[3 * v + 2];
			for (int64_t e = ptr[v]; e < ptr[v + 1]; ++e)
			{
				const int32_t j = col[e];
				if (j == v || state[j] != -1)
					continue;
				const int32_t sj = sv + (vsize ? vsize[j] : 1);
				if (best >= 0 && sj > bs)
					continue;
				const uint32_t we = w ? w[e] : 1u;
				double d2 = 0;
				if (pos)
				{
					const double dx = pv[0] - pos[3 * (int64_t)j], dy = pv[1] - pos[3 * (int64_t)j + 1], dz = pv[2] - pos[3 * (int64_t)j + 2];
					d2 = dx * dx + dy * dy + dz * dz;
				}
				if (best >= 0 && sj == bs && (d2 > bd || (d2 == bd && we < bw)))
					continue;
				const uint64_t lo = (uint64_t)(v < j ? v : j), hi = (uint64_t)(v < j ? j : v);
				const uint64_t h = mix64(mix64((lo << 32) | hi) ^ seed);
				if (best < 0 || sj < bs || d2 < bd || we > bw || h > bh)
					best = j, bs = sj, bw = we, bd = d2, bh = h;
			}
			prop[v] = best;
			live += best >= 0;
		}
		if (live)
			atomicAdd(n_live, live);
	}

	// position of every cluster: mean of its (one or two) vertices weighted by what they stand for; written by the root
	__global__ void k_match_positions(int64_t n, const int32_t *__restrict__ state, const int32_t *__restrict__ prop, const uint32_t *__restrict__ root_id,
									  const double *__restrict__ pos, const int32_t *__restrict__ vsize, double *__restrict__ npos)
	{
		for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x)
		{
			if (state[v] != (int32_t)v)
				continue;
			const int64_t c = root_id[v];
			const int32_t j = prop[v]; // the mate, if v is in a pair: then state[j] == v and j > v
			const bool pair = j > (int32_t)v && state[j] == (int32_t)v;
			const double sv = vsize ? (double)vsize[v] : 1.0, sj = pair ? (vsize ? (double)vsize[j] : 1.0) : 0.0;
			for (int q = 0; q < 3; ++q)
				npos[3 * c + q] = (sv * pos[3 * v + q] + (pair ? sj * pos[3 * (int64_t)j + q] : 0.0)) / (sv + sj);
		}
	}

	__global__ void k_match_accept(int64_t n, const int32_t *__restrict__ prop, int32_t *__restrict__ state)
	{
		for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x)
		{
			if (state[v] != -1)
				continue;
			const int32_t j = prop[v];
			// prop[j] was written this round only if j was free at its start, which v's proposal to j implies
			if (j >= 0 && prop[j] == (int32_t)v)
				state[v] = (int32_t)(v < j ? v : j);
		}
	}

	__global__ void k_match_finish(int64_t n, int32_t *__restrict__ state, uint32_t *__restrict__ is_root)
	{
		for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v <= n; v += (int64_t)gridDim.x * blockDim.x)
		{
			if (v == n)
			{
				is_root[v] = 0;
				continue;
			}
			int32_t s = state[v];
			if (s == -1)
				state[v] = s = (int32_t)v;
			is_root[v] = s == (int32_t)v ? 1u : 0u;
		}
	}

	__global__ void k_match_number(int64_t n, const int32_t *__restrict__ state, const uint32_t *__restrict__ root_id, int32_t *__restrict__ m)
	{
		for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x)
		{
			const int32_t s = state[v];
			m[v] = s >= 0 ? (int32_t)root_id[s] : -1;
		}
	}

	// eight lanes per row: keys[e] = map[row] * nc + map[col] (nc * nc when either end is dropped), vals[e] = e
	__global__ __launch_bounds__(256) void k_quot_keys(int64_t n, const int64_t *__restrict__ ptr, const int32_t *__restrict__ col, const int32_t *__restrict__ map,
														 uint64_t nc, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals, int32_t *__restrict__ erow)
	{
		const int sub = threadIdx.x & 7;
		for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; i < n; i += ((int64_t)gridDim.x * blockDim.x) >> 3)
		{
			const int32_t a = map[i];
			for (int64_t e = ptr[i] + sub; e < ptr[i + 1]; e += 8)
			{
				const int32_t b = a >= 0 ? map[col[e]] : -1;
				keys[e] = (a >= 0 && b >= 0) ? (uint64_t)a * nc + (uint64_t)b : nc * nc;
				vals[e] = (uint32_t)e;
				if (erow)
					erow[e] = (int32_t)i;
			}
		}
	}

	struct quot_head_flag
	{
		const uint64_t *keys;
		uint64_t inv;
		int64_t n;
		__host__ __device__ uint32_t operator()(int64_t k) const
		{
			if (k >= n)
				return 0u;
			const uint64_t key = keys[k];
			if (key >= inv)
				return 0u;
			return (k == 0 || key != keys[k - 1]) ? 1u : 0u;
		}
	};

	// number of valid (kept) entries of the sorted key array: they come first
	__global__ void k_quot_valid(const uint64_t *__restrict__ keys, int64_t n, uint64_t inv, int64_t *__restrict__ n_valid)
	{
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
			if (keys[k] < inv && (k + 1 == n || keys[k + 1] >= inv))
				*n_valid = k + 1;
	}

	// run heads -> coarse edges
	__global__ void k_quot_heads(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ pos, int64_t n_valid, uint64_t nc, int32_t *__restrict__ ccol,
								 int32_t *__restrict__ arow, int64_t *__restrict__ pair_ptr, int64_t n_cp)
	{
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k <= n_valid; k += (int64_t)gridDim.x * blockDim.x)
		{
			if (k == n_valid)
			{
				pair_ptr[n_cp] = n_valid;
				continue;
			}
			const uint64_t key = keys[k];
			if (k == 0 || key != keys[k - 1])
			{
				const uint32_t r = pos[k];
				const uint64_t a = key / nc;
				ccol[r] = (int32_t)(key - a * nc);
				arow[r] = (int32_t)a;
				pair_ptr[r] = k;
			}
		}
	}

	__global__ void k_quot_weights(const int64_t *__restrict__ pair_ptr, const uint32_t *__restrict__ src, const uint32_t *__restrict__ w_fine, int64_t n_cp,
								   uint32_t *__restrict__ cw)
	{
		for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_cp; r += (int64_t)gridDim.x * blockDim.x)
		{
			const int64_t k0 = pair_ptr[r], k1 = pair_ptr[r + 1];
			uint64_t s = 0;
			if (w_fine)
				for (int64_t k = k0; k < k1; ++k)
					s += w_fine[src[k]];
			else
				s = (uint64_t)(k1 - k0);
			cw[r] = s > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)s;
		}
	}

	// keys ascending (values >= n_keys count as n_keys): ptr[x] = first position whose key is >= x, x = 0 .. n_keys
	template <typename K>
	__global__ void k_ptr_from_sorted(const K *__restrict__ keys, int64_t n, int64_t n_keys, int64_t *__restrict__ ptr)
	{
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k <= n; k += (int64_t)gridDim.x * blockDim.x)
		{
			int64_t cur = n_keys, prev = -1;
			if (k < n)
			{
				const int64_t c = (int64_t)keys[k];
				cur = (c < 0 || c > n_keys) ? n_keys : c;
			}
			if (k > 0)
			{
				const int64_t c = (int64_t)keys[k - 1];
				prev = (c < 0 || c > n_keys) ? n_keys : c;
			}
			for (int64_t x = prev + 1; x <= cur; ++x)
				ptr[x] = k;
		}
	}

	__global__ void k_gather_i32(const int32_t *__restrict__ table, const uint32_t *__restrict__ idx, int64_t n, int32_t *__restrict__ out)
	{
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
			out[k] = table[idx[k]];
	}

	__global__ void k_compose(int64_t n, int32_t *__restrict__ cur, const int32_t *__restrict__ m)
	{
		for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x)
		{
			const int32_t c = cur[v];
			if (c >= 0)
				cur[v] = m[c];
		}
	}

	// ---- merging of undersized clusters -----------------------------------------------------------------------
	__global__ void k_cluster_sizes(int64_t n_vert, const int32_t *__restrict__ cur, int32_t *__restrict__ size)
	{
		for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n_vert; v += (int64_t)gridDim.x * blockDim.x)
			if (cur[v] >= 0)
				atomicAdd(&size[cur[v]], 1);
	}

	__global__ void k_iota(int64_t n, int32_t *__restrict__ p)
	{
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
			p[k] = (int32_t)k;
	}

	__device__ __forceinline__ int32_t find_root(const int32_t *parent, int32_t a)
	{
		for (int it = 0; it < 64 && parent[a] != a; ++it)
			a = parent[a];
		return a;
	}

	// small = a root with fewer than min_size vertices.  to_big: the strongest edge to a root that is not small; otherwise
	// the strongest edge to another small root (ties: the smaller pair of ids), taken when the choice is mutual.
	__global__ void k_merge_pick(int64_t nc, const int64_t *__restrict__ ptr, const int32_t *__restrict__ col, const uint32_t *__restrict__ w,
								 const int32_t *__restrict__ parent, const int32_t *__restrict__ size, int min_size, int to_big, int32_t *__restrict__ tgt)
	{
		for (int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; a < nc; a += (int64_t)gridDim.x * blockDim.x)
		{
			tgt[a] = -1;
			if (parent[a] != (int32_t)a || size[a] >= min_size)
				continue;
			int32_t best = -1;
			uint32_t bw = 0;
			for (int64_t e = ptr[a]; e < ptr[a + 1]; ++e)
			{
				const int32_t rb = find_root(parent, col[e]);
				if (rb == (int32_t)a)
					continue;
				const bool small = size[rb] < min_size;
				if (to_big ? small : !small)
					continue;
				const uint32_t we = w ? w[e] : 1u;
				if (best < 0 || we > bw || (we == bw && rb < best))
					best = rb, bw = we;
			}
			tgt[a] = best;
		}
	}

	__global__ void k_merge_apply(int64_t nc, const int32_t *__restrict__ tgt, int to_big, int32_t *__restrict__ parent, int32_t *__restrict__ size,
								  int *__restrict__ changed)
	{
		for (int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; a < nc; a += (int64_t)gridDim.x * blockDim.x)
		{
			const int32_t b = tgt[a];
			if (b < 0)
				continue;
			// to a root that is not small: it does not move in this launch.  Between small roots: only mutual choices, the
			// larger id joins the smaller one
			if (to_big || (tgt[b] == (int32_t)a && (int32_t)a > b))
			{
				parent[a] = b;
				atomicAdd(&size[b], size[a]);
				*changed = 1;
			}
		}
	}

	__global__ void k_root_flags(int64_t nc, const int32_t *__restrict__ parent, uint32_t *__restrict__ flag)
	{
		for (int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; a <= nc; a += (int64_t)gridDim.x * blockDim.x)
			flag[a] = (a < nc && parent[a] == (int32_t)a) ? 1u : 0u;
	}

	__global__ void k_relabel(int64_t n_vert, int32_t *__restrict__ cur, const int32_t *__restrict__ parent, const uint32_t *__restrict__ ren)
	{
		for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n_vert; v += (int64_t)gridDim.x * blockDim.x)
		{
			const int32_t c = cur[v];
			if (c >= 0)
				cur[v] = (int32_t)ren[find_root(parent, c)];
		}
	}

	__global__ void k_member_keys(int64_t n_vert, const int32_t *__restrict__ cur, uint32_t nc, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals)
	{
		for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n_vert; v += (int64_t)gridDim.x * blockDim.x)
		{
			keys[v] = cur[v] >= 0 ? (uint32_t)cur[v] : nc;
			vals[v] = (uint32_t)v;
		}
	}

	int bits_for(uint64_t v)
	{
		int b = 1;
		while (b < 64 && (v >> b))
			++b;
		return b;
	}

	template <class K, class V>
	pfx_status sort_pairs(pfx_ctx *ctx, K *ka, K *kb, V *va, V *vb, int64_t n, int bits)
	{
		if (n <= 0)
			return PFX_OK;
		size_t tb = 0;
		PFX_HIP(rocprim::radix_sort_pairs(nullptr, tb, ka, kb, va, vb, (size_t)n, 0, bits, ctx->stream));
		pfx_tmp<char> tmp;
		PFX_HIP(tmp.alloc(tb));
		PFX_HIP(rocprim::radix_sort_pairs(tmp.p, tb, ka, kb, va, vb, (size_t)n, 0, bits, ctx->stream));
		PFX_HIP(hipStreamSynchronize(ctx->stream));
		return PFX_OK;
	}

	template <class In>
	pfx_status scan_u32(pfx_ctx *ctx, In in, uint32_t *out, int64_t n)
	{
		size_t tb = 0;
		PFX_HIP(rocprim::exclusive_scan(nullptr, tb, in, out, (uint32_t)0, (size_t)n, rocprim::plus<uint32_t>(), ctx->stream));
		pfx_tmp<char> tmp;
		PFX_HIP(tmp.alloc(tb));
		PFX_HIP(rocprim::exclusive_scan(tmp.p, tb, in, out, (uint32_t)0, (size_t)n, rocprim::plus<uint32_t>(), ctx->stream));
		PFX_HIP(hipStreamSynchronize(ctx->stream));
		return PFX_OK;
	}
} // namespace

pfx_status pfx_graph_cluster_sizes(pfx_ctx *ctx, int64_t n_vert, const int32_t *cur, int64_t nc, int32_t *size)
{
	PFX_HIP(hipMemsetAsync(size, 0, sizeof(int32_t) * (size_t)nc, ctx->stream));
	hipLaunchKernelGGL(k_cluster_sizes, dim3(grid_for(n_vert)), dim3(256), 0, ctx->stream, n_vert, cur, size);
	PFX_HIP(hipGetLastError());
	return PFX_OK;
}

pfx_status pfx_graph_match(pfx_ctx *ctx, const pfx_dgraph &g, const uint8_t *excl, const double *pos, const int32_t *vsize, uint64_t seed, int32_t *m, int64_t *nc,
						   pfx_tmp<double> *npos)
{
	hipStream_t st = ctx->stream;
	const int64_t n = g.n;
	pfx_tmp<int32_t> state, prop;
	pfx_tmp<uint32_t> is_root, root_id;
	pfx_tmp<unsigned long long> live;
	PFX_HIP(state.alloc((size_t)n));
	PFX_HIP(prop.alloc((size_t)n));
	PFX_HIP(is_root.alloc((size_t)n + 1));
	PFX_HIP(root_id.alloc((size_t)n + 1));
	PFX_HIP(live.alloc(1));
	const int grid = grid_for(n);
	hipLaunchKernelGGL(k_match_init, dim3(grid), dim3(256), 0, st, n, excl, state.p);
	int rounds = 0;
	for (int round = 0; round < 32; ++round)
	{
		++rounds;
		PFX_HIP(hipMemsetAsync(live.p, 0, sizeof(unsigned long long), st));
		hipLaunchKernelGGL(k_match_propose, dim3(grid), dim3(256), 0, st, n, g.ptr, g.col, g.w, pos, vsize, state.p, seed * 0x100000001B3ull + (uint64_t)round, prop.p, live.p);
		hipLaunchKernelGGL(k_match_accept, dim3(grid), dim3(256), 0, st, n, prop.p, state.p);
		unsigned long long h = 0;
		PFX_HIP(hipMemcpyAsync(&h, live.p, sizeof(h), hipMemcpyDeviceToHost, st));
		PFX_HIP(hipStreamSynchronize(st));
		if (h == 0)
			break;
	}
	hipLaunchKernelGGL(k_match_finish, dim3(grid_for(n + 1)), dim3(256), 0, st, n, state.p, is_root.p);
	pfx_status rc = scan_u32(ctx, is_root.p, root_id.p, n + 1);
	if (rc)
		return rc;
	uint32_t count = 0;
	PFX_HIP(hipMemcpyAsync(&count, root_id.p + n, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
	hipLaunchKernelGGL(k_match_number, dim3(grid), dim3(256), 0, st, n, state.p, root_id.p, m);
	PFX_HIP(hipStreamSynchronize(st));
	if (npos && pos)
	{
		PFX_HIP(npos->alloc((size_t)count * 3));
		hipLaunchKernelGGL(k_match_positions, dim3(grid), dim3(256), 0, st, n, state.p, prop.p, root_id.p, pos, vsize, npos->p);
		PFX_HIP(hipStreamSynchronize(st));
	}
	PFX_HIP(hipGetLastError());
	*nc = (int64_t)count;
	if (getenv("PFX_AMG_VERBOSE"))
		fprintf(stderr, "[pfx amg]   matching: %lld vertices -> %lld clusters in %d rounds\n", (long long)n, (long long)count, rounds);
	return PFX_OK;
}

pfx_status pfx_graph_quotient(pfx_ctx *ctx, const pfx_dgraph &g, const int32_t *map, int64_t nc, pfx_dgraph_own &out, pfx_quot_lists *lists)
{
	hipStream_t st = ctx->stream;
	const int64_t ne = g.ne;
	if (ne >= ((int64_t)1 << 32))
		PFX_FAIL(PFX_ERR_SIZE, "AMG set-up: more than 2^32 edges in a level graph");
	pfx_tmp<uint64_t> keys_a, keys_b;
	pfx_tmp<uint32_t> vals_a, vals_b, pos;
	pfx_tmp<int32_t> erow;
	pfx_tmp<int64_t> n_valid_d;
	PFX_HIP(keys_a.alloc((size_t)ne));
	PFX_HIP(keys_b.alloc((size_t)ne));
	PFX_HIP(vals_a.alloc((size_t)ne));
	PFX_HIP(vals_b.alloc((size_t)ne));
	if (lists)
		PFX_HIP(erow.alloc((size_t)ne));
	const uint64_t inv = (uint64_t)nc * (uint64_t)nc;
	{
		int64_t blocks = (g.n * 8 + 255) / 256;
		if (blocks > 256 * 32)
			blocks = 256 * 32;
		if (blocks < 1)
			blocks = 1;
		hipLaunchKernelGGL(k_quot_keys, dim3((unsigned)blocks), dim3(256), 0, st, g.n, g.ptr, g.col, map, (uint64_t)nc, keys_a.p, vals_a.p, erow.p);
	}
	pfx_status rc = sort_pairs(ctx, keys_a.p, keys_b.p, vals_a.p, vals_b.p, ne, bits_for(inv));
	if (rc)
		return rc;
	keys_a.reset();
	vals_a.reset();
	// valid entries, run heads
	PFX_HIP(n_valid_d.alloc(1));
	PFX_HIP(hipMemsetAsync(n_valid_d.p, 0, sizeof(int64_t), st));
	hipLaunchKernelGGL(k_quot_valid, dim3(grid_for(ne)), dim3(256), 0, st, keys_b.p, ne, inv, n_valid_d.p);
	PFX_HIP(pos.alloc((size_t)ne + 1));
	auto flags = rocprim::make_transform_iterator(rocprim::make_counting_iterator<int64_t>(0), quot_head_flag{keys_b.p, inv, ne});
	if ((rc = scan_u32(ctx, flags, pos.p, ne + 1)))
		return rc;
	int64_t n_valid = 0;
	uint32_t n_cp32 = 0;
	PFX_HIP(hipMemcpyAsync(&n_valid, n_valid_d.p, sizeof(int64_t), hipMemcpyDeviceToHost, st));
	PFX_HIP(hipMemcpyAsync(&n_cp32, pos.p + ne, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
	PFX_HIP(hipStreamSynchronize(st));
	const int64_t n_cp = (int64_t)n_cp32;
	out.ptr.reset(), out.col.reset(), out.w.reset();
	out.n = nc, out.ne = n_cp;
	PFX_HIP(out.ptr.alloc((size_t)nc + 1));
	PFX_HIP(out.col.alloc((size_t)n_cp));
	PFX_HIP(out.w.alloc((size_t)n_cp));
	pfx_tmp<int64_t> pair_ptr;
	pfx_tmp<int32_t> arow;
	PFX_HIP(pair_ptr.alloc((size_t)n_cp + 1));
	PFX_HIP(arow.alloc((size_t)n_cp));
	hipLaunchKernelGGL(k_quot_heads, dim3(grid_for(n_valid + 1)), dim3(256), 0, st, keys_b.p, pos.p, n_valid, (uint64_t)nc, out.col.p, arow.p, pair_ptr.p, n_cp);
	hipLaunchKernelGGL(k_quot_weights, dim3(grid_for(n_cp)), dim3(256), 0, st, pair_ptr.p, vals_b.p, g.w, n_cp, out.w.p);
	hipLaunchKernelGGL((k_ptr_from_sorted<int32_t>), dim3(grid_for(n_cp + 1)), dim3(256), 0, st, arow.p, n_cp, nc, out.ptr.p);
	if (lists)
	{
		lists->n_pairs = n_valid;
		PFX_HIP(lists->pair_row.alloc((size_t)n_valid));
		hipLaunchKernelGGL(k_gather_i32, dim3(grid_for(n_valid)), dim3(256), 0, st, erow.p, vals_b.p, n_valid, lists->pair_row.p);
	}
	PFX_HIP(hipStreamSynchronize(st));
	PFX_HIP(hipGetLastError());
	if (lists)
	{
		lists->pair_ptr.reset(), lists->pair_src.reset(), lists->arow.reset();
		lists->pair_ptr.p = pair_ptr.p, pair_ptr.p = nullptr;
		lists->pair_src.p = vals_b.p, vals_b.p = nullptr; // the first n_pairs entries
		lists->arow.p = arow.p, arow.p = nullptr;
	}
	return PFX_OK;
}

pfx_status pfx_graph_compose(pfx_ctx *ctx, int64_t n, int32_t *cur, const int32_t *m)
{
	hipLaunchKernelGGL(k_compose, dim3(grid_for(n)), dim3(256), 0, ctx->stream, n, cur, m);
	PFX_HIP(hipGetLastError());
	return PFX_OK;
}

pfx_status pfx_graph_merge_small(pfx_ctx *ctx, int64_t n_vert, int32_t *cur, int64_t *nc_io, const pfx_dgraph &gq, int min_size)
{
	hipStream_t st = ctx->stream;
	const int64_t nc = *nc_io;
	pfx_tmp<int32_t> size, parent, tgt;
	pfx_tmp<int> changed;
	pfx_tmp<uint32_t> flag, ren;
	PFX_HIP(size.alloc((size_t)nc));
	PFX_HIP(parent.alloc((size_t)nc));
	PFX_HIP(tgt.alloc((size_t)nc));
	PFX_HIP(changed.alloc(1));
	pfx_status rc = pfx_graph_cluster_sizes(ctx, n_vert, cur, nc, size.p);
	if (rc)
		return rc;
	hipLaunchKernelGGL(k_iota, dim3(grid_for(nc)), dim3(256), 0, st, nc, parent.p);
	const int grid = grid_for(nc);
	bool any = false;
	for (int round = 0; round < 8; ++round)
	{
		PFX_HIP(hipMemsetAsync(changed.p, 0, sizeof(int), st));
		for (int to_big = 1; to_big >= 0; --to_big)
		{
			hipLaunchKernelGGL(k_merge_pick, dim3(grid), dim3(256), 0, st, nc, gq.ptr, gq.col, gq.w, parent.p, size.p, min_size, to_big, tgt.p);
			hipLaunchKernelGGL(k_merge_apply, dim3(grid), dim3(256), 0, st, nc, tgt.p, to_big, parent.p, size.p, changed.p);
		}
		int h = 0;
		PFX_HIP(hipMemcpyAsync(&h, changed.p, sizeof(int), hipMemcpyDeviceToHost, st));
		PFX_HIP(hipStreamSynchronize(st));
		if (!h)
			break;
		any = true;
	}
	if (!any)
		return PFX_OK;
	PFX_HIP(flag.alloc((size_t)nc + 1));
	PFX_HIP(ren.alloc((size_t)nc + 1));
	hipLaunchKernelGGL(k_root_flags, dim3(grid_for(nc + 1)), dim3(256), 0, st, nc, parent.p, flag.p);
	if ((rc = scan_u32(ctx, flag.p, ren.p, nc + 1)))
		return rc;
	uint32_t count = 0;
	PFX_HIP(hipMemcpyAsync(&count, ren.p + nc, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
	hipLaunchKernelGGL(k_relabel, dim3(grid_for(n_vert)), dim3(256), 0, st, n_vert, cur, parent.p, ren.p);
	PFX_HIP(hipStreamSynchronize(st));
	PFX_HIP(hipGetLastError());
	*nc_io = (int64_t)count;
	return PFX_OK;
}

pfx_status pfx_graph_members(pfx_ctx *ctx, int64_t n_vert, const int32_t *cur, int64_t nc, int64_t *mptr, int32_t *mlist)
{
	hipStream_t st = ctx->stream;
	pfx_tmp<uint32_t> ka, kb, va;
	PFX_HIP(ka.alloc((size_t)n_vert));
	PFX_HIP(kb.alloc((size_t)n_vert));
	PFX_HIP(va.alloc((size_t)n_vert));
	hipLaunchKernelGGL(k_member_keys, dim3(grid_for(n_vert)), dim3(256), 0, st, n_vert, cur, (uint32_t)nc, ka.p, va.p);
	pfx_status rc = sort_pairs(ctx, ka.p, kb.p, va.p, reinterpret_cast<uint32_t *>(mlist), n_vert, bits_for((uint64_t)nc));
	if (rc)
		return rc;
	hipLaunchKernelGGL((k_ptr_from_sorted<uint32_t>), dim3(grid_for(n_vert + 1)), dim3(256), 0, st, kb.p, n_vert, nc, mptr);
	PFX_HIP(hipStreamSynchronize(st));
	PFX_HIP(hipGetLastError());
	return PFX_OK;
}
