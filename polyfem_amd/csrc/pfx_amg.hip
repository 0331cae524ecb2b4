// pfx_amg.hip -- aggregation multigrid preconditioner for the CG solver (3-D elasticity, block size 3; scalar problems,
// block size 1; 2-D elasticity, block size 2, with the three rigid-body modes of the plane).
//
// Role in the reference: polysolve's algebraic-multigrid solvers ("AMGCL", "Hypre"; /root/reference/CMakeLists.txt:101-102,
// options input-spec.json:1397-1558: CG with an AMG preconditioner, Chebyshev relaxation, ncycle, max_levels) are what a
// PolyFEM user selects when Jacobi-CG does not converge in max_iter.  Neither library is part of the reference tree, and
// this is NOT a restatement of either: it is this library's own plain-aggregation AMG, built on the node-block
// structure the assembler already has.  Jacobi stays the default preconditioner (Eigen parity).
//
// Hierarchy.  Level 0 = the assembled matrix (vertices = mesh nodes, 3 dofs each).  A level is coarsened by
// `aggregate_passes` rounds of pairwise matching on the vertex graph (weights = number of fine connections; the first
// round of a level, where all weights are equal, pairs nearest neighbours; pfx_amg_graph.hip), so
// aggregates hold up to 2^passes vertices.  Every aggregate carries the six rigid-body modes of its vertices about the
// aggregate's centre c: a translation T and a rotation W, u(x) = T + W x (x - c).  They are stored as TWO coarse nodes
// of 3 dofs (node 2a = T, node 2a+1 = W), so every level is again a node-block matrix with 3x3 blocks and reuses the
// assembler's SELL matrix-vector product.  Prolongation of vertex v (offset d = x_v - c from its aggregate's centre):
//     level 0:   u_v = T + W x d                       P_v   = [ I  -[d]x ]                (3 x 6)
//     level > 0: (T_v, W_v) = (T + W x d, W)           P_v   = [ I  -[d]x ; 0  I ]         (6 x 6)
// Coarse matrices are Galerkin products P^T A P, computed on the device from lists (built with the hierarchy) of the
// fine vertex pairs behind every coarse pair, in a fixed order: no atomics, bitwise reproducible.
// Scalar problems (the Laplacian): one coarse dof per aggregate, piecewise constant interpolation, the same aggregation,
// cycle and scaling; every level is a block-size-1 matrix.
// Dirichlet dofs: the level-0 operator is the masked one (identity on constrained dofs); their rows of P are zero,
// nodes with all dofs constrained are left out of the aggregation.
//
// Cycle (one application = the preconditioner z = M^-1 r): Chebyshev relaxation of degree `relax_degree` for
// D^-1 A on [relax_lower * lmax, lmax] before and after the coarse correction (lmax by a few Lanczos steps), V or W cycle,
// dense inverse at the coarsest level.  Same polynomial before and after: M is symmetric positive definite.  Every
// coarse-grid correction is scaled by `coarse_scale` in [1, 2) (the piecewise rigid-body interpolation of plain
// aggregation underestimates the smooth error; the factor makes up for it and leaves M positive definite).
//
// Over a partition (one rank per GPU, pfx_dist_cg_dev) the SAME hierarchy spans the ranks: every rank aggregates the
// vertices it owns (an aggregate never crosses a rank boundary), learns from its neighbours which aggregate each of its halo
// vertices went into (one halo exchange of the aggregate ids and of the offsets d per level), and forms the rows of the
// Galerkin product that belong to its own aggregates -- complete, because its fine rows are.  A coarse level is therefore
// again a row-distributed matrix with halo columns; every product of the cycle is preceded by a halo exchange of that
// level's vector, the dot products of the Lanczos estimate are all-reduced, and the coarsest matrix is gathered and
// inverted on every rank.  The preconditioner is the one a single rank would build from the union of the ranks'
// aggregates (the tests compare it with exactly that), so the iteration count does not degrade with the number of ranks
// the way a per-rank (block-Jacobi) hierarchy does.
//
// The symbolic part (matching, coarse patterns, Galerkin lists) runs once per pattern and Dirichlet set, the numeric
// part (Galerkin products, diagonals, spectral radii, coarsest inverse) at every new matrix; both on the device, except
// for the dense inverse of the coarsest matrix (a few hundred dofs).
#include "pfx_internal.h"
#include "pfx_amg_graph.h"
#include "pfx_dist.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cfloat>

#define AMG_MAX_LEVELS 16
#define AMG_MAX_DENSE 4096 // largest coarsest level the dense inverse is formed for (n^2 doubles on the host and the device)
#define AMG_NPART 1024 // blocks of the reducing vector kernels (256 left the streaming kernels of the PCG at 2.4 - 3 TB/s: 110 ms of a 1.02 s solve)
#define AMG_SMALL_NODES 32768 // levels of at most this many node rows multiply through k_rowwave_spmv

namespace
{
	// ---------------------------------------------------------------------------------------------------------
	// device kernels
	// ---------------------------------------------------------------------------------------------------------
	struct node_ref
	{
		int n_loc, n_gloc;
		double gval[PFX_MAX_LOC][PFX_MAX_GLOC];
	};

	// position of every basis node: geometry map of its first incident element evaluated at the local node's reference
	// position
	__global__ void k_node_positions(const int64_t *__restrict__ ne_ptr, const uint32_t *__restrict__ ne_src, const int32_t *__restrict__ gconn,
									 const double *__restrict__ xyz, node_ref ref, int64_t n_nodes, double *__restrict__ pos)
	{
		for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_nodes; g += (int64_t)gridDim.x * blockDim.x)
		{
			double p[3] = {0, 0, 0};
			if (ne_ptr[g] < ne_ptr[g + 1])
			{
				const uint32_t s = ne_src[ne_ptr[g]];
				const int64_t e = s / (uint32_t)ref.n_loc;
				const int i = (int)(s - (uint32_t)e * (uint32_t)ref.n_loc);
				for (int j = 0; j < ref.n_gloc; ++j)
				{
					const int64_t gv = gconn[e * ref.n_gloc + j];
					const double v = ref.gval[i][j];
					p[0] += v * xyz[3 * gv], p[1] += v * xyz[3 * gv + 1], p[2] += v * xyz[3 * gv + 2];
				}
			}
			pos[3 * g] = p[0], pos[3 * g + 1] = p[1], pos[3 * g + 2] = p[2];
		}
	}

	__device__ __forceinline__ void load9(const double *__restrict__ p, double *a)
	{
#pragma unroll
		for (int q = 0; q < 9; ++q)
			a[q] = p[q];
	}

	// out = [d]x * m  (row-major 3x3)
	__device__ __forceinline__ void cross_left(const double *d, const double *m, double *out)
	{
#pragma unroll
		for (int c = 0; c < 3; ++c)
		{
			out[0 + c] = d[1] * m[6 + c] - d[2] * m[3 + c];
			out[3 + c] = d[2] * m[0 + c] - d[0] * m[6 + c];
			out[6 + c] = d[0] * m[3 + c] - d[1] * m[0 + c];
		}
	}

	// out = -m * [d]x : column c of m [d]x is  m (e_c-th column of [d]x);  (m [d]x)(r, .) = row_r(m) x ... written out:
	// [d]x = [[0,-d2,d1],[d2,0,-d0],[-d1,d0,0]]
	__device__ __forceinline__ void neg_cross_right(const double *m, const double *d, double *out)
	{
#pragma unroll
		for (int r = 0; r < 3; ++r)
		{
			const double a = m[3 * r], b = m[3 * r + 1], c = m[3 * r + 2];
			out[3 * r + 0] = -(b * d[2] - c * d[1]);
			out[3 * r + 1] = -(c * d[0] - a * d[2]);
			out[3 * r + 2] = -(a * d[1] - b * d[0]);
		}
	}

	// Galerkin product of one level: eight lanes per (coarse pair, column half beta); lane s takes the fine pairs s, s + 8,
	// ... of the coarse pair's list, a fixed-order butterfly adds the eight partial sums, and the first lane writes the
	// two coarse 3x3 blocks (T row and W row) of its column half.  (One thread per list: 8.4 ms for the 38 M fine blocks
	// of 1 M quadratic tetrahedra, a tenth of the bandwidth; lists hold 17 fine pairs on average.)
	// LEVEL0: fine pairs are single 3x3 blocks of the assembled matrix (index = block index), constrained dofs masked;
	// else fine pairs are 2x2 groups of 3x3 blocks of a coarse level's node-block matrix.
	template <bool LEVEL0>
	__global__ __launch_bounds__(256) void k_galerkin(const int64_t *__restrict__ cp_ptr, const uint32_t *__restrict__ cp_src, const int32_t *__restrict__ cp_row,
														 const int64_t *__restrict__ fine_gptr, const int32_t *__restrict__ fine_gcol, const double *__restrict__ fine_values,
														 const double *__restrict__ dvec, const uint8_t *__restrict__ fixed, const int64_t *__restrict__ coarse_gptr,
														 const int32_t *__restrict__ coarse_row_of_pair, int64_t n_cpairs, double *__restrict__ coarse_values)
	{
		const int sub = threadIdx.x & 7;
		const int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
		// the grid holds whole waves: lanes beyond the last list idle through the shuffles
		const bool live = t < 2 * n_cpairs;
		const int64_t cp = live ? t >> 1 : 0;
		const int beta = (int)(t & 1);
		double ct[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, cr[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
		for (int64_t k = live ? cp_ptr[cp] + sub : 0; k < (live ? cp_ptr[cp + 1] : 0); k += 8)
		{
			const int64_t q = cp_src[k];
			const int64_t i = cp_row[k];
			const int64_t j = fine_gcol[q];
			const double di[3] = {dvec[3 * i], dvec[3 * i + 1], dvec[3 * i + 2]};
			const double dj[3] = {dvec[3 * j], dvec[3 * j + 1], dvec[3 * j + 2]};
			double mt[9], mr[9];
			if (LEVEL0)
			{
				double a[9];
				load9(fine_values + q * 9, a);
				if (fixed)
#pragma unroll
					for (int m = 0; m < 3; ++m)
#pragma unroll
						for (int n = 0; n < 3; ++n)
							if (fixed[3 * i + m] || fixed[3 * j + n])
								a[3 * m + n] = 0.0;
				if (beta == 0)
				{
#pragma unroll
					for (int z = 0; z < 9; ++z)
						mt[z] = a[z];
				}
				else
					neg_cross_right(a, dj, mt);
#pragma unroll
				for (int z = 0; z < 9; ++z)
					mr[z] = 0.0;
			}
			else
			{
				// blocks of the fine pair (I = i, slot s = q - gptr[i]): (alpha, beta') at 4 gptr[i] + alpha * 2 deg + 2 s + beta'
				const int64_t g0 = fine_gptr[i], deg = fine_gptr[i + 1] - g0, s = q - g0;
				const double *att = fine_values + (4 * g0 + 2 * s) * 9;
				const double *atr = att + 9;
				const double *art = fine_values + (4 * g0 + 2 * deg + 2 * s) * 9;
				const double *arr = art + 9;
				if (beta == 0)
				{
					load9(att, mt);
					load9(art, mr);
				}
				else
				{
					double a[9], b[9];
					load9(att, a);
					neg_cross_right(a, dj, mt);
					load9(atr, b);
#pragma unroll
					for (int z = 0; z < 9; ++z)
						mt[z] += b[z];
					load9(art, a);
					neg_cross_right(a, dj, mr);
					load9(arr, b);
#pragma unroll
					for (int z = 0; z < 9; ++z)
						mr[z] += b[z];
				}
			}
			double x[9];
			cross_left(di, mt, x);
#pragma unroll
			for (int z = 0; z < 9; ++z)
			{
				ct[z] += mt[z];
				cr[z] += x[z] + mr[z];
			}
		}
#pragma unroll
		for (int off = 4; off > 0; off >>= 1)
#pragma unroll
			for (int z = 0; z < 9; ++z)
			{
				ct[z] += __shfl_xor(ct[z], off, 64);
				cr[z] += __shfl_xor(cr[z], off, 64);
			}
		if (!live || sub != 0)
			return;
		const int64_t a = coarse_row_of_pair[cp];
		const int64_t g0 = coarse_gptr[a], deg = coarse_gptr[a + 1] - g0, s = cp - g0;
		double *ot = coarse_values + (4 * g0 + 2 * s + beta) * 9;
		double *orr = coarse_values + (4 * g0 + 2 * deg + 2 * s + beta) * 9;
#pragma unroll
		for (int z = 0; z < 9; ++z)
		{
			ot[z] = ct[z];
			orr[z] = cr[z];
		}
	}

	// coarse right-hand side: per aggregate, sum over its member vertices of P_v^T (b - t).  Sixteen lanes per aggregate
	// (an aggregate has 8 - 20 members): lane m takes the members m, m + 16, ..., then a fixed-order butterfly.
	template <bool LEVEL0>
	__global__ __launch_bounds__(256) void k_restrict(const int64_t *__restrict__ mptr, const int32_t *__restrict__ mlist, const double *__restrict__ dvec,
														 const double *__restrict__ b, const double *__restrict__ t, int64_t n_agg, double *__restrict__ bc)
	{
		const int sub = threadIdx.x & 15;
		for (int64_t a = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4; a < ((n_agg + 3) & ~(int64_t)3); a += ((int64_t)gridDim.x * blockDim.x) >> 4)
		{
			// (the loop bound is rounded up to whole waves of four aggregates so that every lane reaches the shuffles)
			double T[3] = {0, 0, 0}, W[3] = {0, 0, 0};
			if (a < n_agg)
				for (int64_t k = mptr[a] + sub; k < mptr[a + 1]; k += 16)
				{
					const int64_t v = mlist[k];
					const double d[3] = {dvec[3 * v], dvec[3 * v + 1], dvec[3 * v + 2]};
					const int64_t o = LEVEL0 ? 3 * v : 6 * v;
					const double r[3] = {b[o] - t[o], b[o + 1] - t[o + 1], b[o + 2] - t[o + 2]};
					T[0] += r[0], T[1] += r[1], T[2] += r[2];
					W[0] += d[1] * r[2] - d[2] * r[1];
					W[1] += d[2] * r[0] - d[0] * r[2];
					W[2] += d[0] * r[1] - d[1] * r[0];
					if (!LEVEL0)
					{
						W[0] += b[o + 3] - t[o + 3];
						W[1] += b[o + 4] - t[o + 4];
						W[2] += b[o + 5] - t[o + 5];
					}
				}
#pragma unroll
			for (int off = 8; off > 0; off >>= 1)
#pragma unroll
				for (int q = 0; q < 3; ++q)
				{
					T[q] += __shfl_xor(T[q], off, 64);
					W[q] += __shfl_xor(W[q], off, 64);
				}
			if (a < n_agg && sub == 0)
			{
				bc[6 * a] = T[0], bc[6 * a + 1] = T[1], bc[6 * a + 2] = T[2];
				bc[6 * a + 3] = W[0], bc[6 * a + 4] = W[1], bc[6 * a + 5] = W[2];
			}
		}
	}

	// x += P xc
	template <bool LEVEL0>
	__global__ __launch_bounds__(256) void k_prolong(const int32_t *__restrict__ agg, const double *__restrict__ dvec, const double *__restrict__ xc,
														const uint8_t *__restrict__ fixed, int64_t n_vert, double omega, double *__restrict__ x)
	{
		for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n_vert; v += (int64_t)gridDim.x * blockDim.x)
		{
			const int64_t a = agg[v];
			if (a < 0)
				continue;
			const double d[3] = {dvec[3 * v], dvec[3 * v + 1], dvec[3 * v + 2]};
			const double T[3] = {omega * xc[6 * a], omega * xc[6 * a + 1], omega * xc[6 * a + 2]}, W[3] = {omega * xc[6 * a + 3], omega * xc[6 * a + 4], omega * xc[6 * a + 5]};
			const double u[3] = {T[0] + W[1] * d[2] - W[2] * d[1], T[1] + W[2] * d[0] - W[0] * d[2], T[2] + W[0] * d[1] - W[1] * d[0]};
			const int64_t o = LEVEL0 ? 3 * v : 6 * v;
#pragma unroll
			for (int m = 0; m < 3; ++m)
				if (!(LEVEL0 && fixed && fixed[o + m]))
					x[o + m] += u[m];
			if (!LEVEL0)
			{
				x[o + 3] += W[0];
				x[o + 4] += W[1];
				x[o + 5] += W[2];
			}
		}
	}

	// ---- piecewise constant interpolation, component by component: BS coarse dofs per aggregate on every level (scalar
	// problems: block size 1, the Laplacian, where the constant is the near-null space).  A node whose dofs are all
	// constrained is in no aggregate; a partly constrained one is, and `fixed` (level 0 only) keeps its constrained dofs out
	// of the Galerkin sums and of both transfers.  One lane per entry of a coarse block; the sums run in list order.
	template <int BS>
	__global__ __launch_bounds__(256) void k_galerkin_const(const int64_t *__restrict__ cp_ptr, const uint32_t *__restrict__ cp_src, const int32_t *__restrict__ cp_row,
															 const int32_t *__restrict__ node_col, const double *__restrict__ fine_values, const uint8_t *__restrict__ fixed,
															 int64_t n_cpairs, double *__restrict__ coarse_values)
	{
		constexpr int B2 = BS * BS;
		const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
		if (t >= n_cpairs * B2)
			return;
		const int64_t cp = t / B2;
		const int q = (int)(t - cp * B2), m = q / BS, n = q - m * BS;
		double acc = 0;
		for (int64_t k = cp_ptr[cp]; k < cp_ptr[cp + 1]; ++k)
		{
			const int64_t blk = cp_src[k];
			if (fixed && (fixed[(int64_t)cp_row[k] * BS + m] || fixed[(int64_t)node_col[blk] * BS + n]))
				continue;
			acc += fine_values[blk * B2 + q];
		}
		coarse_values[t] = acc;
	}

	template <int BS>
	__global__ __launch_bounds__(256) void k_restrict_const(const int64_t *__restrict__ mptr, const int32_t *__restrict__ mlist, const double *__restrict__ b,
															 const double *__restrict__ t, const uint8_t *__restrict__ fixed, int64_t n_agg, double *__restrict__ bc)
	{
		for (int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; a < n_agg; a += (int64_t)gridDim.x * blockDim.x)
		{
			double acc[BS];
			for (int m = 0; m < BS; ++m)
				acc[m] = 0;
			for (int64_t k = mptr[a]; k < mptr[a + 1]; ++k)
			{
				const int64_t v = mlist[k];
				for (int m = 0; m < BS; ++m)
					if (!fixed || !fixed[v * BS + m])
						acc[m] += b[v * BS + m] - t[v * BS + m];
			}
			for (int m = 0; m < BS; ++m)
				bc[a * BS + m] = acc[m];
		}
	}

	template <int BS>
	__global__ __launch_bounds__(256) void k_prolong_const(const int32_t *__restrict__ agg, const double *__restrict__ xc, const uint8_t *__restrict__ fixed, int64_t n_vert,
															double omega, double *__restrict__ x)
	{
		for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n_vert; v += (int64_t)gridDim.x * blockDim.x)
		{
			const int64_t a = agg[v];
			if (a < 0)
				continue;
			for (int m = 0; m < BS; ++m)
				if (!fixed || !fixed[v * BS + m])
					x[v * BS + m] += omega * xc[a * BS + m];
		}
	}

	// ---- 2-D elasticity (block size 2): three rigid-body modes per aggregate -- the two translations and the in-plane
	// rotation about the aggregate's centre -- held as ONE node of block size 3 per aggregate on every coarse level.  With
	// d the offset of a vertex from its aggregate's centre (z = 0), P_v = [[1, 0, -d_y], [0, 1, d_x]] on level 0 and the same
	// with the row (0, 0, 1) below it between coarse levels: the plane case of u = T + W x d.
	// Galerkin product: one lane per coarse pair, C = sum over its fine pairs of P_i^T B_ij P_j in list order.
	template <bool LEVEL0>
	__global__ __launch_bounds__(256) void k_galerkin_plane(const int64_t *__restrict__ cp_ptr, const uint32_t *__restrict__ cp_src, const int32_t *__restrict__ cp_row,
															 const int32_t *__restrict__ node_col, const double *__restrict__ fine_values, const double *__restrict__ dvec,
															 const uint8_t *__restrict__ fixed, int64_t n_cpairs, double *__restrict__ coarse_values)
	{
		constexpr int FR = LEVEL0 ? 2 : 3; // rows / columns of a fine block
		const int64_t cp = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
		if (cp >= n_cpairs)
			return;
		double c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
		for (int64_t k = cp_ptr[cp]; k < cp_ptr[cp + 1]; ++k)
		{
			const int64_t q = cp_src[k];
			const int64_t i = cp_row[k];
			const int64_t j = node_col[q];
			const double dix = dvec[3 * i], diy = dvec[3 * i + 1], djx = dvec[3 * j], djy = dvec[3 * j + 1];
			double a[FR * FR];
#pragma unroll
			for (int z = 0; z < FR * FR; ++z)
				a[z] = fine_values[q * (FR * FR) + z];
			if (LEVEL0 && fixed)
#pragma unroll
				for (int m = 0; m < 2; ++m)
#pragma unroll
					for (int n = 0; n < 2; ++n)
						if (fixed[2 * i + m] || fixed[2 * j + n])
							a[2 * m + n] = 0.0;
			// t = B P_j (FR x 3), then c += P_i^T t
			double t[FR][3];
#pragma unroll
			for (int m = 0; m < FR; ++m)
			{
				t[m][0] = a[FR * m];
				t[m][1] = a[FR * m + 1];
				t[m][2] = a[FR * m + 1] * djx - a[FR * m] * djy;
				if (!LEVEL0)
					t[m][2] += a[FR * m + 2];
			}
#pragma unroll
			for (int n = 0; n < 3; ++n)
			{
				c[n] += t[0][n];
				c[3 + n] += t[1][n];
				double w = t[1][n] * dix - t[0][n] * diy;
				if (!LEVEL0)
					w += t[FR - 1][n];
				c[6 + n] += w;
			}
		}
#pragma unroll
		for (int z = 0; z < 9; ++z)
			coarse_values[cp * 9 + z] = c[z];
	}

	template <bool LEVEL0>
	__global__ __launch_bounds__(256) void k_restrict_plane(const int64_t *__restrict__ mptr, const int32_t *__restrict__ mlist, const double *__restrict__ dvec,
															 const double *__restrict__ b, const double *__restrict__ t, const uint8_t *__restrict__ fixed, int64_t n_agg,
															 double *__restrict__ bc)
	{
		constexpr int FR = LEVEL0 ? 2 : 3;
		for (int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; a < n_agg; a += (int64_t)gridDim.x * blockDim.x)
		{
			double T0 = 0, T1 = 0, W = 0;
			for (int64_t k = mptr[a]; k < mptr[a + 1]; ++k)
			{
				const int64_t v = mlist[k];
				const int64_t o = FR * v;
				double r0 = b[o] - t[o], r1 = b[o + 1] - t[o + 1];
				if (LEVEL0 && fixed)
				{
					r0 = fixed[o] ? 0.0 : r0;
					r1 = fixed[o + 1] ? 0.0 : r1;
				}
				T0 += r0, T1 += r1;
				W += dvec[3 * v] * r1 - dvec[3 * v + 1] * r0;
				if (!LEVEL0)
					W += b[o + 2] - t[o + 2];
			}
			bc[3 * a] = T0, bc[3 * a + 1] = T1, bc[3 * a + 2] = W;
		}
	}

	template <bool LEVEL0>
	__global__ __launch_bounds__(256) void k_prolong_plane(const int32_t *__restrict__ agg, const double *__restrict__ dvec, const double *__restrict__ xc,
															const uint8_t *__restrict__ fixed, int64_t n_vert, double omega, double *__restrict__ x)
	{
		constexpr int FR = LEVEL0 ? 2 : 3;
		for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n_vert; v += (int64_t)gridDim.x * blockDim.x)
		{
			const int64_t a = agg[v];
			if (a < 0)
				continue;
			const double T0 = omega * xc[3 * a], T1 = omega * xc[3 * a + 1], W = omega * xc[3 * a + 2];
			const double u0 = T0 - W * dvec[3 * v + 1], u1 = T1 + W * dvec[3 * v];
			const int64_t o = FR * v;
			if (!(LEVEL0 && fixed && fixed[o]))
				x[o] += u0;
			if (!(LEVEL0 && fixed && fixed[o + 1]))
				x[o + 1] += u1;
			if (!LEVEL0)
				x[o + 2] += W;
		}
	}

	// y = A x for a SMALL level straight from its node-block arrays: one wave per node row, the lanes over the row's
	// blocks, a fixed-order butterfly over the wave.  The slice format gives every slice to one wave, which walks its
	// longest row slot by slot: right for the large levels, but a level of a few thousand rows of 50 - 130 blocks is then a
	// few dozen waves in a latency chain (32 - 39 us per product on levels of 1 - 13 k rows; 5 - 10 us with this kernel).
	template <int BS>
	__global__ __launch_bounds__(256) void k_rowwave_spmv(const int64_t *__restrict__ node_ptr, const int32_t *__restrict__ node_col, const double *__restrict__ values,
														  const double *__restrict__ x, int64_t n_nodes, double *__restrict__ y)
	{
		const int lane = threadIdx.x & 63;
		const int64_t g = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
		if (g >= n_nodes)
			return;
		double acc[BS];
#pragma unroll
		for (int m = 0; m < BS; ++m)
			acc[m] = 0.0;
		for (int64_t b = node_ptr[g] + lane; b < node_ptr[g + 1]; b += 64)
		{
			const int64_t c = node_col[b];
			double xv[BS];
#pragma unroll
			for (int n = 0; n < BS; ++n)
				xv[n] = x[c * BS + n];
			const double *a = values + b * (BS * BS);
#pragma unroll
			for (int m = 0; m < BS; ++m)
#pragma unroll
				for (int n = 0; n < BS; ++n)
					acc[m] += a[m * BS + n] * xv[n];
		}
#pragma unroll
		for (int off = 32; off > 0; off >>= 1)
#pragma unroll
			for (int m = 0; m < BS; ++m)
				acc[m] += __shfl_xor(acc[m], off, 64);
		if (lane < BS)
		{
			double v = acc[0];
#pragma unroll
			for (int m = 1; m < BS; ++m)
				v = lane == m ? acc[m] : v;
			y[g * BS + lane] = v;
		}
	}

	// relaxation pieces (Chebyshev for D^-1 A, see amg_relax)
	__global__ void k_cheb_first(const double *__restrict__ dinv, const double *__restrict__ b, const double *__restrict__ t, double s, int zero_guess,
								 int64_t n, double *__restrict__ r, double *__restrict__ dv, double *__restrict__ x)
	{
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
		{
			const double rk = zero_guess ? b[k] : b[k] - t[k];
			const double d = s * dinv[k] * rk;
			if (r)
			{
				r[k] = rk;
				dv[k] = d;
			}
			x[k] = zero_guess ? d : x[k] + d;
		}
	}

	__global__ void k_cheb_next(const double *__restrict__ dinv, const double *__restrict__ t, double c1, double c2, int64_t n, double *__restrict__ r,
								double *__restrict__ dv, double *__restrict__ x)
	{
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
		{
			const double rk = r[k] - t[k];
			const double d = c1 * dv[k] + c2 * dinv[k] * rk;
			r[k] = rk;
			dv[k] = d;
			x[k] += d;
		}
	}

	__global__ void k_vec_add(const double *__restrict__ a, int64_t n, double *__restrict__ x)
	{
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
			x[k] += a[k];
	}

	__global__ void k_vec_sub(const double *__restrict__ a, const double *__restrict__ b, int64_t n, double *__restrict__ out)
	{
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
			out[k] = a[k] - b[k];
	}

	// x = Ainv b (dense, row-major, n x n): one wave per row
	__global__ __launch_bounds__(64) void k_dense_matvec(const double *__restrict__ ainv, const double *__restrict__ b, int n, double *__restrict__ x)
	{
		const int row = blockIdx.x;
		double acc = 0;
		for (int k = threadIdx.x; k < n; k += 64)
			acc += ainv[(int64_t)row * n + k] * b[k];
		for (int off = 32; off > 0; off >>= 1)
			acc += __shfl_down(acc, off, 64);
		if (threadIdx.x == 0)
			x[row] = acc;
	}

	__device__ __forceinline__ double block_sum256(double v, double *sh)
	{
		sh[threadIdx.x] = v;
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

	// partial sums of a.b (AMG_NPART blocks of 256 threads), fixed order
	__global__ __launch_bounds__(256) void k_dot_part(const double *__restrict__ a, const double *__restrict__ b, int64_t n, double *__restrict__ part)
	{
		__shared__ double sh[256];
		double acc = 0;
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
			acc += a[k] * b[k];
		const double s = block_sum256(acc, sh);
		if (threadIdx.x == 0)
			part[blockIdx.x] = s;
	}

	__global__ __launch_bounds__(256) void k_dot_finish(const double *__restrict__ part, int count, double *__restrict__ out)
	{
		__shared__ double sh[256];
		double acc = 0;
		for (int k = threadIdx.x; k < count; k += 256)
			acc += part[k];
		const double s = block_sum256(acc, sh);
		if (threadIdx.x == 0)
			*out = s;
	}

	// Lanczos steps for S = D^-1/2 A D^-1/2 (the eigenvalues of D^-1 A): q the current vector, qp the previous one,
	// u = D^-1/2 q the vector A is applied to, t = A u.  scal: alpha_j, beta_{j+1}^2 as they are produced.
	// w = D^-1/2 t - beta qp; partial sums of w.q
	__global__ __launch_bounds__(256) void k_lz_a(const double *__restrict__ dinv, const double *__restrict__ t, const double *__restrict__ qp,
													const double *__restrict__ q, const double *__restrict__ beta2, int64_t n, double *__restrict__ w,
													double *__restrict__ part)
	{
		__shared__ double sh[256];
		const double beta = beta2 ? sqrt(*beta2) : 0.0;
		double acc = 0;
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
		{
			const double v = sqrt(dinv[k]) * t[k] - beta * qp[k];
			w[k] = v;
			acc += v * q[k];
		}
		const double s = block_sum256(acc, sh);
		if (threadIdx.x == 0)
			part[blockIdx.x] = s;
	}

	// w -= alpha q; partial sums of w.w
	__global__ __launch_bounds__(256) void k_lz_b(const double *__restrict__ q, const double *__restrict__ alpha, int64_t n, double *__restrict__ w,
													double *__restrict__ part)
	{
		__shared__ double sh[256];
		const double a = *alpha;
		double acc = 0;
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
		{
			const double v = w[k] - a * q[k];
			w[k] = v;
			acc += v * v;
		}
		const double s = block_sum256(acc, sh);
		if (threadIdx.x == 0)
			part[blockIdx.x] = s;
	}

	// qp = q; q = w / sqrt(nrm2) (0 once the Krylov space is exhausted); u = D^-1/2 q
	__global__ void k_lz_c(const double *__restrict__ dinv, const double *__restrict__ w, const double *__restrict__ nrm2, int64_t n, double *__restrict__ qp,
						   double *__restrict__ q, double *__restrict__ u)
	{
		const double s = *nrm2 > 1e-300 ? 1.0 / sqrt(*nrm2) : 0.0;
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
		{
			const double v = w[k] * s;
			qp[k] = q[k];
			q[k] = v;
			u[k] = sqrt(dinv[k]) * v;
		}
	}

	__global__ void k_seed_vector(int64_t n, const uint8_t *__restrict__ fixed, double *__restrict__ v)
	{
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
		{
			uint64_t z = (uint64_t)k * 0x9E3779B97F4A7C15ull + 0x1234567ull;
			z ^= z >> 31;
			z *= 0xBF58476D1CE4E5B9ull;
			z ^= z >> 29;
			const double u = (double)(z >> 11) / 9007199254740992.0 - 0.5;
			v[k] = (fixed && fixed[k]) ? 0.0 : u;
		}
	}

	// ---- preconditioned CG with a general preconditioner (Eigen's operation order, z = M^-1 r as its own vector) ------
	struct pcg_state
	{
		double abs_new, pap, rr, threshold, rhs_norm2, tmp0, tmp1;
		int done, iters;
	};

	__global__ __launch_bounds__(256) void k_pcg_init(const double *__restrict__ b, const double *__restrict__ t, const uint8_t *__restrict__ fixed, int64_t n,
													 double *__restrict__ r, double *__restrict__ part)
	{
		__shared__ double sh[256];
		double bb = 0, rr = 0;
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
		{
			const bool fx = fixed && fixed[k];
			const double bk = fx ? 0.0 : b[k];
			const double rk = fx ? 0.0 : bk - t[k];
			r[k] = rk;
			bb += bk * bk;
			rr += rk * rk;
		}
		const double s0 = block_sum256(bb, sh);
		const double s1 = block_sum256(rr, sh);
		if (threadIdx.x == 0)
		{
			part[blockIdx.x] = s0;
			part[AMG_NPART + blockIdx.x] = s1;
		}
	}

	__global__ __launch_bounds__(256) void k_pcg_init2(const double *__restrict__ part, double tol, int64_t n, double *__restrict__ x, pcg_state *__restrict__ st)
	{
		__shared__ double sh[256];
		double a0 = 0, a1 = 0;
		for (int k = threadIdx.x; k < AMG_NPART; k += 256)
			a0 += part[k], a1 += part[AMG_NPART + k];
		const double bb = block_sum256(a0, sh);
		const double rr = block_sum256(a1, sh);
		double thr = tol * tol * bb;
		if (thr < DBL_MIN)
			thr = DBL_MIN;
		const bool zero_rhs = bb == 0.0;
		if (zero_rhs)
			for (int64_t k = threadIdx.x; k < n; k += 256)
				x[k] = 0.0;
		if (threadIdx.x == 0)
		{
			st->rhs_norm2 = bb;
			st->threshold = thr;
			st->rr = zero_rhs ? 0.0 : rr;
			st->done = (zero_rhs || rr < thr) ? 1 : 0;
			st->iters = 0;
		}
	}

	// first: p = z, abs_new = r.z ; later: beta = r.z / abs_new, p = z + beta p      (r.z partials in part[2*NPART..])
	__global__ __launch_bounds__(256) void k_pcg_direction(const double *__restrict__ z, const double *__restrict__ part, int first, int64_t n,
														  double *__restrict__ p, pcg_state *__restrict__ st)
	{
		__shared__ double sh[256];
		if (st->done)
			return;
		double a = 0;
		for (int k = threadIdx.x; k < AMG_NPART; k += 256)
			a += part[2 * AMG_NPART + k];
		const double rz = block_sum256(a, sh);
		const double beta = first ? 0.0 : rz / st->abs_new;
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
			p[k] = first ? z[k] : z[k] + beta * p[k];
		// every block has read abs_new before the last block can have finished?  No such guarantee: the new value goes
		// to tmp0 and is moved by k_pcg_commit, a separate launch.
		if (blockIdx.x == 0 && threadIdx.x == 0)
			st->tmp0 = rz;
	}

	__global__ void k_pcg_commit(pcg_state *__restrict__ st)
	{
		if (!st->done)
			st->abs_new = st->tmp0;
	}

	// alpha = abs_new / p.Ap ; x += alpha p ; r -= alpha t ; partials of r.r
	__global__ __launch_bounds__(256) void k_pcg_update(const double *__restrict__ p, const double *__restrict__ t, const double *__restrict__ pap, int64_t n,
													   double *__restrict__ x, double *__restrict__ r, double *__restrict__ part, const pcg_state *__restrict__ st)
	{
		__shared__ double sh[256];
		if (st->done)
			return;
		const double alpha = st->abs_new / *pap;
		double rr = 0;
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
		{
			x[k] += alpha * p[k];
			const double rk = r[k] - alpha * t[k];
			r[k] = rk;
			rr += rk * rk;
		}
		const double s = block_sum256(rr, sh);
		if (threadIdx.x == 0)
			part[AMG_NPART + blockIdx.x] = s;
	}

	// convergence test after the update of iteration `it` (Eigen breaks before counting the iteration)
	__global__ __launch_bounds__(256) void k_pcg_test(const double *__restrict__ part, int it, pcg_state *__restrict__ st)
	{
		__shared__ double sh[256];
		if (st->done)
			return;
		double a = 0;
		for (int k = threadIdx.x; k < AMG_NPART; k += 256)
			a += part[AMG_NPART + k];
		const double rr = block_sum256(a, sh);
		if (threadIdx.x == 0)
		{
			st->rr = rr;
			if (rr < st->threshold)
			{
				st->done = 1;
				st->iters = it;
			}
			else
				st->iters = it + 1;
		}
	}

	__global__ void k_mask(double *__restrict__ v, const uint8_t *__restrict__ fixed, int64_t n)
	{
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
			if (fixed[k])
				v[k] = 0.0;
	}

	double now_s()
	{
		return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
	}
} // namespace

// One level of the hierarchy.  `m` holds the level's matrix (level 0: the context itself; coarser levels: a context of
// their own that shares the stream and carries only pattern, values and the solver copy).
struct pfx_amg_level
{
	pfx_ctx *m = nullptr;
	bool own_m = false;
	int64_t n_vert = 0, n_dofs = 0; // vertices of this level; dofs of its vectors
	int64_t n_agg = 0, n_cpairs = 0;
	// transfer to the next level (empty on the coarsest)
	int32_t *d_agg = nullptr;  // vertex -> aggregate or -1
	double *d_dvec = nullptr;  // vertex -> offset from its aggregate's centre
	int64_t *d_mptr = nullptr; // aggregate -> members
	int32_t *d_mlist = nullptr;
	int64_t *d_cp_ptr = nullptr; // coarse pair -> fine pairs
	uint32_t *d_cp_src = nullptr;
	int32_t *d_cp_row = nullptr;
	int32_t *d_cp_arow = nullptr; // coarse pair -> its aggregate row
	int64_t *d_gptr = nullptr;    // this level's vertex graph (levels > 0: pair structure of the matrix; level 0: unused)
	int32_t *d_gcol = nullptr;
	// vectors
	double *d_b = nullptr, *d_x = nullptr, *d_t = nullptr, *d_r = nullptr, *d_dv = nullptr, *d_b2 = nullptr, *d_x2 = nullptr, *d_dinv = nullptr;
	double lmax = 0;
	bool small = false; // products through k_rowwave_spmv on the node-block arrays; no solver copy of the matrix
	// partitioned hierarchies: vertices are numbered [owned | halo]; n_vert / n_dofs count both, these the owned ones
	int64_t n_overt = 0, n_odofs = 0;
	pfx_halo halo;      // halo lists of this level's vertices (levels > 0; level 0 uses the partition's own)
	int64_t n_agg_all = 0; // owned aggregates + aggregates of neighbour ranks that the halo vertices went into
};

struct pfx_amg
{
	std::vector<pfx_amg_level> lv;
	int64_t struct_gen = -1, values_gen = -1;
	double shift = -1;
	double *d_ainv = nullptr; // dense inverse of the coarsest matrix
	int n_coarsest = 0;
	double *d_part = nullptr; // 3 * AMG_NPART partial sums + scalars
	double *d_scal = nullptr;
	double *d_z = nullptr, *d_r0 = nullptr, *d_dv0 = nullptr, *d_pos = nullptr;
	pcg_state *d_state = nullptr;
	double symbolic_s = 0, numeric_s = 0;
	int64_t work_blocks = 0; // sum of the levels' blocks
	// hierarchy that spans the ranks of a partition
	pfx_dist *dist = nullptr;
	int64_t coarse_off = 0;     // first global dof of this rank on the coarsest level
	double *d_bglob = nullptr;  // coarsest right-hand side of all ranks (n_coarsest doubles)
};

namespace
{
	void free_level(pfx_ctx *ctx, pfx_amg_level &L)
	{
		free_dev(ctx, L.d_agg);
		free_dev(ctx, L.d_dvec);
		free_dev(ctx, L.d_mptr);
		free_dev(ctx, L.d_mlist);
		free_dev(ctx, L.d_cp_ptr);
		free_dev(ctx, L.d_cp_src);
		free_dev(ctx, L.d_cp_row);
		free_dev(ctx, L.d_cp_arow);
		free_dev(ctx, L.d_gptr);
		free_dev(ctx, L.d_gcol);
		free_dev(ctx, L.d_b);
		free_dev(ctx, L.d_x);
		free_dev(ctx, L.d_t);
		free_dev(ctx, L.d_r);
		free_dev(ctx, L.d_dv);
		free_dev(ctx, L.d_b2);
		free_dev(ctx, L.d_x2);
		pfx_halo_release(&L.halo);
		if (L.own_m)
		{
			free_dev(ctx, L.d_dinv);
			pfx_ctx *s = L.m;
			pfx_sell_release(s);
			free_dev(s, s->d_node_ptr);
			free_dev(s, s->d_node_col);
			free_dev(s, s->d_values);
			free_dev(s, s->d_part0);
			delete s;
		}
		L.m = nullptr;
	}

	template <typename T>
	pfx_status upload(pfx_ctx *ctx, T **dst, const std::vector<T> &src)
	{
		if (pfx_alloc(ctx, dst, (int64_t)src.size()))
			return PFX_ERR_HIP;
		if (!src.empty())
			PFX_HIP(hipMemcpyAsync(*dst, src.data(), sizeof(T) * src.size(), hipMemcpyHostToDevice, ctx->stream));
		return PFX_OK;
	}

	// ---- kernels of the symbolic phase -----------------------------------------------------------------------
	// vertices left out of the aggregation at level 0: halo nodes and nodes whose three dofs are all constrained
	__global__ void k_excluded(const uint8_t *__restrict__ fixed, int64_t n_nodes, int64_t n_owned, int bs, uint8_t *__restrict__ excl)
	{
		for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_nodes; g += (int64_t)gridDim.x * blockDim.x)
		{
			bool all = fixed != nullptr;
			for (int m = 0; all && m < bs; ++m)
				all = fixed[bs * g + m] != 0;
			excl[g] = (g >= n_owned) || all;
		}
	}

	// one node per aggregate (scalar and plane problems): the coarse matrix has one block per coarse edge, its pattern is the coarse graph itself
	__global__ void k_copy_i64(const int64_t *__restrict__ src, int64_t n, int64_t *__restrict__ dst)
	{
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
			dst[k] = src[k];
	}

	// centre of every aggregate: mean of its members' positions, summed in member order
	__global__ void k_centres(const int64_t *__restrict__ mptr, const int32_t *__restrict__ mlist, const double *__restrict__ vpos, int64_t n_agg,
							  double *__restrict__ centre)
	{
		for (int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; a < n_agg; a += (int64_t)gridDim.x * blockDim.x)
		{
			double c[3] = {0, 0, 0};
			const int64_t k0 = mptr[a], k1 = mptr[a + 1];
			for (int64_t k = k0; k < k1; ++k)
			{
				const int64_t v = mlist[k];
				c[0] += vpos[3 * v], c[1] += vpos[3 * v + 1], c[2] += vpos[3 * v + 2];
			}
			const double cnt = (double)(k1 - k0);
			centre[3 * a] = c[0] / cnt, centre[3 * a + 1] = c[1] / cnt, centre[3 * a + 2] = c[2] / cnt;
		}
	}

	__global__ void k_offsets(const int32_t *__restrict__ agg, const double *__restrict__ vpos, const double *__restrict__ centre, int64_t n_vert,
							  double *__restrict__ dvec)
	{
		for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n_vert; v += (int64_t)gridDim.x * blockDim.x)
		{
			const int64_t a = agg[v];
			for (int q = 0; q < 3; ++q)
				dvec[3 * v + q] = a >= 0 ? vpos[3 * v + q] - centre[3 * a + q] : 0.0;
		}
	}

	// node-block pattern of a coarse level: two nodes (translation, rotation) per aggregate; the blocks of coarse edge k
	// of aggregate a (slot s = k - gptr[a], deg = its number of edges) sit at 4 gptr[a] + alpha * 2 deg + 2 s + beta
	__global__ void k_coarse_pattern(const int64_t *__restrict__ gptr, const int32_t *__restrict__ gcol, int64_t n_agg, int64_t *__restrict__ nptr,
									 int32_t *__restrict__ ncol)
	{
		for (int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; a <= n_agg; a += (int64_t)gridDim.x * blockDim.x)
		{
			if (a == n_agg)
			{
				nptr[2 * n_agg] = 4 * gptr[n_agg];
				continue;
			}
			const int64_t g0 = gptr[a], deg = gptr[a + 1] - g0;
			nptr[2 * a] = 4 * g0;
			nptr[2 * a + 1] = 4 * g0 + 2 * deg;
			for (int64_t k = 0; k < deg; ++k)
			{
				const int32_t c = gcol[g0 + k];
				for (int al = 0; al < 2; ++al)
					for (int be = 0; be < 2; ++be)
						ncol[4 * g0 + al * 2 * deg + 2 * k + be] = 2 * c + be;
			}
		}
	}

	// moves a temporary buffer into an allocation the context accounts for (device-to-device copy of `count` entries)
	template <typename T>
	pfx_status keep(pfx_ctx *ctx, T **dst, const T *src, int64_t count)
	{
		if (pfx_alloc(ctx, dst, count))
			return PFX_ERR_HIP;
		if (count > 0)
			PFX_HIP(hipMemcpyAsync(*dst, src, sizeof(T) * (size_t)count, hipMemcpyDeviceToDevice, ctx->stream));
		return PFX_OK;
	}

	// ---- pieces of a hierarchy that spans the ranks of a partition ------------------------------------------------
	// v[0 .. n) := sum over the ranks (host numbers, n <= 32; the same bits on every rank)
	pfx_status dist_sum(pfx_ctx *ctx, pfx_dist *dist, double *v, int n)
	{
		if (!dist)
			return PFX_OK;
		hipStream_t st = ctx->stream;
		PFX_HIP(hipMemcpyAsync(dist->d_scal + 32, v, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st));
		if (pfx_comm_all_reduce(dist->comm, st, dist->d_scal + 32, n))
			PFX_FAIL(PFX_ERR_HIP, pfx_comm_error(dist->comm));
		PFX_HIP(hipMemcpyAsync(v, dist->d_scal + 32, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, st));
		PFX_HIP(hipStreamSynchronize(st));
		return PFX_OK;
	}

	// the halo entries of a vector of level L (`width` doubles per vertex) from their owners
	pfx_status level_exchange(pfx_ctx *ctx, pfx_amg *A, pfx_amg_level &L, double *vec, int width)
	{
		pfx_dist *dist = A->dist;
		if (!dist)
			return PFX_OK;
		pfx_halo *h = L.m == ctx ? &dist->halo : &L.halo;
		if (pfx_halo_start(dist->comm, h, ctx->stream, vec, width) || pfx_halo_finish(dist->comm, h, ctx->stream))
			PFX_FAIL(PFX_ERR_HIP, pfx_comm_error(dist->comm));
		return PFX_OK;
	}

	__global__ void k_excl_halo(int64_t n_vert, int64_t n_overt, uint8_t *__restrict__ excl)
	{
		for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n_vert; v += (int64_t)gridDim.x * blockDim.x)
			excl[v] = v >= n_overt;
	}

	__global__ void k_i32_to_f64(const int32_t *__restrict__ a, int64_t n, double *__restrict__ out)
	{
		for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x)
			out[v] = (double)a[v];
	}

	// After the local aggregation of a level (cur[v] in [0, nc) for owned vertices, -1 for halo and left-out ones): learns
	// from the neighbours which of THEIR aggregates every halo vertex went into, gives those aggregates local ids nc, nc + 1,
	// ... (neighbour by neighbour, ascending remote id: both sides derive the same order) and writes them into cur; builds
	// the halo lists of the coarse level (coarse vertex = aggregate).  *nct = nc + number of such halo aggregates.
	pfx_status dist_halo_aggregates(pfx_ctx *ctx, pfx_amg *A, pfx_amg_level &L, int32_t *cur, int64_t n_vert, int64_t n_overt, int64_t nc, pfx_halo *coarse,
									int64_t *nct, int coarse_width)
	{
		pfx_dist *dist = A->dist;
		hipStream_t st = ctx->stream;
		pfx_halo *h = L.m == ctx ? &dist->halo : &L.halo;
		const int nn = (int)h->peer.size();
		pfx_tmp<double> ids;
		PFX_HIP(ids.alloc((size_t)n_vert));
		hipLaunchKernelGGL(k_i32_to_f64, dim3(grid_for(n_vert)), dim3(256), 0, st, cur, n_vert, ids.p);
		pfx_status rc = level_exchange(ctx, A, L, ids.p, 1);
		if (rc)
			return rc;
		const int64_t n_halo = n_vert - n_overt;
		std::vector<double> hid((size_t)n_halo);
		std::vector<int32_t> hcur((size_t)n_vert);
		if (n_halo > 0)
			PFX_HIP(hipMemcpyAsync(hid.data(), ids.p + n_overt, sizeof(double) * (size_t)n_halo, hipMemcpyDeviceToHost, st));
		PFX_HIP(hipMemcpyAsync(hcur.data(), cur, sizeof(int32_t) * (size_t)n_vert, hipMemcpyDeviceToHost, st));
		PFX_HIP(hipStreamSynchronize(st));
		std::vector<int64_t> csend_ptr((size_t)nn + 1, 0), crecv_ptr((size_t)nn + 1, 0);
		std::vector<int32_t> csend;
		int64_t next = nc;
		for (int k = 0; k < nn; ++k)
		{
			// what arrives from neighbour k: the distinct aggregates of the halo vertices it owns
			std::vector<int64_t> u;
			for (int64_t j = h->recv_ptr[(size_t)k]; j < h->recv_ptr[(size_t)k + 1]; ++j)
				if (hid[(size_t)j] >= 0.0)
					u.push_back((int64_t)hid[(size_t)j]);
			std::sort(u.begin(), u.end());
			u.erase(std::unique(u.begin(), u.end()), u.end());
			for (int64_t j = h->recv_ptr[(size_t)k]; j < h->recv_ptr[(size_t)k + 1]; ++j)
			{
				const double r = hid[(size_t)j];
				hcur[(size_t)(n_overt + j)] = r >= 0.0 ? (int32_t)(next + (std::lower_bound(u.begin(), u.end(), (int64_t)r) - u.begin())) : -1;
			}
			next += (int64_t)u.size();
			crecv_ptr[(size_t)k + 1] = next - nc;
			// what goes to neighbour k: the distinct aggregates of the vertices sent to it, in the same (ascending) order
			std::vector<int32_t> w;
			for (int64_t j = h->send_ptr[(size_t)k]; j < h->send_ptr[(size_t)k + 1]; ++j)
			{
				const int32_t a = hcur[(size_t)h->h_send_idx[(size_t)j]];
				if (a >= 0)
					w.push_back(a);
			}
			std::sort(w.begin(), w.end());
			w.erase(std::unique(w.begin(), w.end()), w.end());
			csend.insert(csend.end(), w.begin(), w.end());
			csend_ptr[(size_t)k + 1] = (int64_t)csend.size();
		}
		if (n_halo > 0)
			PFX_HIP(hipMemcpyAsync(cur + n_overt, hcur.data() + n_overt, sizeof(int32_t) * (size_t)n_halo, hipMemcpyHostToDevice, st));
		PFX_HIP(hipStreamSynchronize(st));
		*nct = next;
		if (pfx_halo_setup(dist->comm, coarse, nc, nn, h->peer.data(), csend_ptr.data(), csend.data(), crecv_ptr.data(), coarse_width))
			PFX_FAIL(PFX_ERR_HIP, pfx_comm_error(dist->comm));
		return PFX_OK;
	}

	pfx_status amg_symbolic(pfx_ctx *ctx)
	{
		const double t_begin = now_s();
		pfx_amg_release(ctx);
		pfx_amg *A = new pfx_amg();
		ctx->amg = A;
		pfx_dist *dist = ctx->amg_dist; // the hierarchy spans the ranks of this partition (null: it is this context's own)
		A->dist = dist;
		const pfx_amg_params &par = ctx->amg_params;
		hipStream_t st = ctx->stream;
		const int64_t n_nodes = ctx->n_nodes, n_owned = ctx->n_owned;
		const int bs = ctx->bs;
		// dofs of a coarse vertex: the constant (scalar problems), the three rigid-body modes of the plane (block size 2) or
		// the six of space (block size 3).  Up to three they are one node of that block size; the six are two nodes of block size 3.
		const int per = bs == 1 ? 1 : bs == 2 ? 3 : 6;
		const bool scalar = bs == 1, one_node = bs < 3;
		pfx_status rc;

		// level-0 node positions and the vertices left out of the aggregation
		if (pfx_alloc(ctx, &A->d_pos, n_nodes * 3))
			return PFX_ERR_HIP;
		if (ctx->ext)
		{
			// a caller's matrix: positions come from pfx_solver_set_node_positions.  The rigid-body modes of the vector
			// problem cannot be formed without them; the scalar problem only uses them to prefer near neighbours.
			if (ctx->d_ext_pos)
				PFX_HIP(hipMemcpyAsync(A->d_pos, ctx->d_ext_pos, sizeof(double) * (size_t)(n_nodes * 3), hipMemcpyDeviceToDevice, st));
			else if (scalar)
				PFX_HIP(hipMemsetAsync(A->d_pos, 0, sizeof(double) * (size_t)(n_nodes * 3), st));
			else
				PFX_FAIL(PFX_ERR_ARG, "the AMG preconditioner of a caller's block-2 or block-3 matrix needs node positions (pfx_solver_set_node_positions), or analyze with block size 1");
		}
		else
		{
			node_ref ref;
			double lp[PFX_MAX_LOC][3];
			pfx_local_node_positions(ctx->kind, &ref.n_loc, &ref.n_gloc, lp, ref.gval);
			hipLaunchKernelGGL(k_node_positions, dim3(grid_for(n_nodes)), dim3(256), 0, st, ctx->d_ne_ptr, ctx->d_ne_src, ctx->d_gconn, ctx->d_xyz, ref, n_nodes, A->d_pos);
		}
		pfx_tmp<uint8_t> excluded;
		PFX_HIP(excluded.alloc((size_t)n_nodes));
		hipLaunchKernelGGL(k_excluded, dim3(grid_for(n_nodes)), dim3(256), 0, st, ctx->n_dirichlet > 0 ? (const uint8_t *)ctx->d_fixed : (const uint8_t *)nullptr, n_nodes, n_owned,
						   ctx->bs, excluded.p);

		// the level being coarsened: its vertex graph (level 0: the node-block pattern itself) and vertex positions
		pfx_dgraph g;
		g.n = n_nodes, g.ne = ctx->n_blocks, g.ptr = ctx->d_node_ptr, g.col = ctx->d_node_col, g.w = nullptr;
		pfx_dgraph_own glevel; // owns the graph of levels > 0 while they are coarsened
		pfx_tmp<double> centre_prev;
		const double *vpos = A->d_pos;
		int64_t n_vert = n_nodes;
		A->work_blocks = ctx->n_blocks;

		pfx_amg_level L;
		L.m = ctx;
		L.own_m = false;
		L.n_vert = n_vert;
		L.n_dofs = n_nodes * bs;
		L.n_overt = n_owned;
		L.n_odofs = n_owned * bs;
		int64_t n_overt = n_owned; // owned vertices of the level being coarsened (all of them unless the hierarchy spans ranks)
		for (int level = 0;; ++level)
		{
			const bool level0 = level == 0;
			// every rank takes the same decisions: they are made on the sums over the ranks
			double gl[2] = {(double)(level0 ? bs * n_owned : per * n_overt), (double)n_overt};
			if ((rc = dist_sum(ctx, dist, gl, 2)))
				return rc;
			if ((!level0 && gl[0] <= (double)par.coarsest_dofs) || gl[1] < 2.0 || level + 1 >= par.max_levels || level + 1 >= AMG_MAX_LEVELS)
			{
				A->lv.push_back(L);
				break;
			}
			// halo vertices take no part in the aggregation of a level that spans the ranks (level 0: `excluded` says so already)
			pfx_tmp<uint8_t> excl_halo;
			if (dist && !level0)
			{
				PFX_HIP(excl_halo.alloc((size_t)n_vert));
				hipLaunchKernelGGL(k_excl_halo, dim3(grid_for(n_vert)), dim3(256), 0, st, n_vert, n_overt, excl_halo.p);
			}
			// ---- aggregation by repeated pairwise matching: every round matches the clusters of the round before
			pfx_tmp<int32_t> cur, m, csize;
			pfx_tmp<double> ppos, ppos_next; // positions of the clusters of the last round
			PFX_HIP(cur.alloc((size_t)n_vert));
			pfx_dgraph_own gq;
			int64_t nc = 0;
			for (int pass = 0; pass < par.aggregate_passes; ++pass)
			{
				const uint64_t seed = (uint64_t)level * 16 + (uint64_t)pass + 1;
				if (pass == 0)
				{
					if ((rc = pfx_graph_match(ctx, g, level0 ? excluded.p : excl_halo.p, vpos, nullptr, seed, cur.p, &nc, &ppos)))
						return rc;
				}
				else
				{
					PFX_HIP(m.alloc((size_t)gq.n));
					PFX_HIP(csize.alloc((size_t)gq.n));
					if ((rc = pfx_graph_cluster_sizes(ctx, n_vert, cur.p, gq.n, csize.p)) || (rc = pfx_graph_match(ctx, gq.view(), nullptr, ppos.p, csize.p, seed, m.p, &nc, &ppos_next))
						|| (rc = pfx_graph_compose(ctx, n_vert, cur.p, m.p)))
						return rc;
					ppos.reset();
					ppos.p = ppos_next.p, ppos_next.p = nullptr;
				}
				// the clusters' graph: quotient of the previous round's (same graph as the quotient of the level's graph by `cur`)
				pfx_dgraph_own next;
				if (pass == 0)
					rc = pfx_graph_quotient(ctx, g, cur.p, nc, next, nullptr);
				else
					rc = pfx_graph_quotient(ctx, gq.view(), m.p, nc, next, nullptr);
				if (rc)
					return rc;
				gq.take(next);
				if (nc <= 1)
					break;
			}
			// level 0: aggregates of fewer than three nodes (two in the plane) cannot carry the rotations; they join their strongest neighbour
			if (level0 && nc > 1 && !scalar)
				if ((rc = pfx_graph_merge_small(ctx, n_vert, cur.p, &nc, gq.view(), bs == 2 ? 2 : 3)))
					return rc;
			gq.ptr.reset(), gq.col.reset(), gq.w.reset();
			m.reset();
			csize.reset();
			ppos.reset();
			{
				double ga[2] = {(double)nc, (double)n_overt};
				if ((rc = dist_sum(ctx, dist, ga, 2)))
					return rc;
				if (ga[0] < 1.0 || (!level0 && ga[0] * 10.0 > ga[1] * 8.0)) // no free vertex, or the coarsening stalls
				{
					A->lv.push_back(L);
					break;
				}
			}
			// ---- members, centres, offsets (cur holds -1 for halo vertices here: members are owned vertices)
			if (pfx_alloc(ctx, &L.d_mptr, nc + 1) || pfx_alloc(ctx, &L.d_mlist, n_vert) || pfx_alloc(ctx, &L.d_dvec, scalar ? 1 : n_vert * 3))
				return PFX_ERR_HIP;
			if ((rc = pfx_graph_members(ctx, n_vert, cur.p, nc, L.d_mptr, L.d_mlist)))
				return rc;
			// ---- a hierarchy that spans the ranks: the aggregates of the halo vertices, and the coarse level's halo lists
			int64_t nct = nc; // coarse vertices this rank sees: its own aggregates, then those its halo vertices went into
			pfx_halo chalo;
			if (dist)
				if ((rc = dist_halo_aggregates(ctx, A, L, cur.p, n_vert, n_overt, nc, &chalo, &nct, scalar ? 3 : 6)))
					return rc;
			pfx_tmp<double> centre;
			PFX_HIP(centre.alloc((size_t)nct * 3));
			PFX_HIP(hipMemsetAsync(centre.p, 0, sizeof(double) * (size_t)nct * 3, st));
			hipLaunchKernelGGL(k_centres, dim3(grid_for(nc)), dim3(256), 0, st, L.d_mptr, L.d_mlist, vpos, nc, centre.p);
			if (!scalar)
			{
				// offsets of the owned vertices from their aggregates' centres; the owners send those of the halo vertices
				hipLaunchKernelGGL(k_offsets, dim3(grid_for(n_overt)), dim3(256), 0, st, cur.p, vpos, centre.p, n_overt, L.d_dvec);
				if (n_vert > n_overt)
					PFX_HIP(hipMemsetAsync(L.d_dvec + 3 * n_overt, 0, sizeof(double) * (size_t)(3 * (n_vert - n_overt)), st));
				if ((rc = level_exchange(ctx, A, L, L.d_dvec, 3)))
					return rc;
			}
			// ---- coarse graph with the Galerkin lists: the rows of the owned vertices, all their columns
			pfx_dgraph g_rows = g;
			if (dist && n_overt < n_vert)
			{
				int64_t ne_owned = 0;
				PFX_HIP(hipMemcpyAsync(&ne_owned, g.ptr + n_overt, sizeof(int64_t), hipMemcpyDeviceToHost, st));
				PFX_HIP(hipStreamSynchronize(st));
				g_rows.n = n_overt, g_rows.ne = ne_owned;
			}
			pfx_dgraph_own gc;
			pfx_quot_lists lists;
			if ((rc = pfx_graph_quotient(ctx, g_rows, cur.p, nct, gc, &lists)))
				return rc;
			const int64_t n_cp = gc.ne;
			if (getenv("PFX_AMG_VERBOSE"))
				fprintf(stderr, "[pfx amg] level %d: %lld vertices, %lld edges -> %lld aggregates, %lld coarse edges, %lld pairs (%.3f s)\n", level, (long long)n_vert,
						(long long)g.ne, (long long)nc, (long long)n_cp, (long long)lists.n_pairs, now_s() - t_begin);
			// ---- the transfer data of this level
			L.n_agg = nc;
			L.n_agg_all = nct;
			L.n_cpairs = n_cp;
			if (keep(ctx, &L.d_agg, cur.p, n_vert) || keep(ctx, &L.d_cp_ptr, lists.pair_ptr.p, n_cp + 1) || keep(ctx, &L.d_cp_src, lists.pair_src.p, lists.n_pairs)
				|| keep(ctx, &L.d_cp_row, lists.pair_row.p, lists.n_pairs) || keep(ctx, &L.d_cp_arow, lists.arow.p, n_cp))
				return PFX_ERR_HIP;
			PFX_HIP(hipStreamSynchronize(st));
			lists.pair_ptr.reset(), lists.pair_src.reset(), lists.pair_row.reset(), lists.arow.reset();
			cur.reset();
			A->lv.push_back(L);

			// ---- the next level: a matrix holder with the pattern of the coarse pairs (two nodes per aggregate)
			pfx_amg_level C;
			pfx_ctx *s = new pfx_ctx();
			s->device = ctx->device;
			s->stream = ctx->stream;
			s->own_stream = false;
			s->bs = one_node ? per : 3;
			s->dim = 3;
			s->n_owned = one_node ? nc : 2 * nc;
			s->n_nodes = one_node ? nct : 2 * nct;
			s->n_blocks = one_node ? n_cp : 4 * n_cp;
			s->nnz = s->n_blocks * s->bs * s->bs;
			C.m = s;
			C.own_m = true;
			C.n_vert = nct;
			C.n_overt = nc;
			C.n_dofs = per * nct;
			C.n_odofs = per * nc;
			C.halo = chalo; // (the lists' device arrays now belong to the level)
			C.small = s->n_owned <= AMG_SMALL_NODES;
			if (pfx_alloc(s, &s->d_node_ptr, s->n_nodes + 1) || pfx_alloc(s, &s->d_node_col, s->n_blocks) || pfx_alloc(s, &s->d_values, s->nnz + 2) || keep(ctx, &C.d_gptr, gc.ptr.p, nct + 1)
				|| keep(ctx, &C.d_gcol, gc.col.p, n_cp))
			{
				if (ctx->err.empty())
					ctx->err = s->err;
				A->lv.push_back(C); // released with the hierarchy
				return PFX_ERR_HIP;
			}
			if (one_node)
			{
				hipLaunchKernelGGL(k_copy_i64, dim3(grid_for(nct + 1)), dim3(256), 0, st, C.d_gptr, nct + 1, s->d_node_ptr);
				PFX_HIP(hipMemcpyAsync(s->d_node_col, C.d_gcol, sizeof(int32_t) * (size_t)n_cp, hipMemcpyDeviceToDevice, st));
			}
			else
				hipLaunchKernelGGL(k_coarse_pattern, dim3(grid_for(nct + 1)), dim3(256), 0, st, C.d_gptr, C.d_gcol, nct, s->d_node_ptr, s->d_node_col);
			PFX_HIP(hipMemsetAsync(s->d_values, 0, sizeof(double) * (size_t)(s->nnz + 2), st));
			PFX_HIP(hipStreamSynchronize(st));
			PFX_HIP(hipGetLastError());
			L = C;
			A->work_blocks += s->n_blocks;
			// descend: the coarse graph (with its weights) is the next level's vertex graph, the centres its positions
			glevel.take(gc);
			g = glevel.view();
			n_vert = nct;
			n_overt = nc;
			// the centres of the neighbours' aggregates from their owners: the positions of the next level's halo vertices
			if ((rc = level_exchange(ctx, A, L, centre.p, 3)))
				return rc;
			centre_prev.reset();
			centre_prev.p = centre.p, centre.p = nullptr;
			vpos = centre_prev.p;
		}
		// the coarsest level is solved densely on every rank: this rank's place in the global numbering of its dofs
		A->coarse_off = 0;
		if (dist && A->lv.size() >= 2)
		{
			const int world = pfx_comm_world(dist->comm), rank = pfx_comm_rank(dist->comm);
			if (world > 32)
				PFX_FAIL(PFX_ERR_UNSUPPORTED, "a hierarchy over more than 32 ranks is not implemented");
			double cnt[32] = {0};
			cnt[rank] = (double)A->lv.back().n_odofs;
			if ((rc = dist_sum(ctx, dist, cnt, world)))
				return rc;
			for (int r = 0; r < rank; ++r)
				A->coarse_off += (int64_t)cnt[r];
		}
		// vectors
		const int64_t n0 = ctx->n_dofs();
		if (pfx_alloc(ctx, &A->d_part, 3 * AMG_NPART + 8) || pfx_alloc(ctx, &A->d_scal, 8) || pfx_alloc(ctx, &A->d_z, n0) || pfx_alloc(ctx, &A->d_state, 1))
			return PFX_ERR_HIP;
		PFX_HIP(hipMemsetAsync(A->d_z, 0, sizeof(double) * (size_t)n0, st));
		for (size_t l = 0; l < A->lv.size(); ++l)
		{
			pfx_amg_level &L = A->lv[l];
			const int64_t n = L.n_dofs;
			if (l == 0)
			{
				// level 0 works on the caller's vectors (b = r, x = z) and the solver's scratch; its own are the Chebyshev pair
				if (par.relax_degree > 1)
				{
					if (pfx_alloc(ctx, &L.d_r, n) || pfx_alloc(ctx, &L.d_dv, n))
						return PFX_ERR_HIP;
					// entries of halo nodes are read by the level-0 product and never written: they stay zero
					PFX_HIP(hipMemsetAsync(L.d_dv, 0, sizeof(double) * (size_t)n, st));
				}
				continue;
			}
			if (pfx_alloc(ctx, &L.d_b, n) || pfx_alloc(ctx, &L.d_x, n) || pfx_alloc(ctx, &L.d_t, n) || pfx_alloc(ctx, &L.d_dinv, n) || pfx_alloc(ctx, &L.d_b2, n) || pfx_alloc(ctx, &L.d_x2, n))
				return PFX_ERR_HIP;
			if (par.relax_degree > 1 && (pfx_alloc(ctx, &L.d_r, n) || pfx_alloc(ctx, &L.d_dv, n)))
				return PFX_ERR_HIP;
		}
		PFX_HIP(hipStreamSynchronize(st));
		A->struct_gen = ctx->struct_gen;
		A->values_gen = -1;
		A->symbolic_s = now_s() - t_begin;
		return PFX_OK;
	}

	// largest eigenvalue of the symmetric tridiagonal matrix (a_1..a_m; b_2..b_m): bisection on the Sturm count
	double tridiagonal_lambda_max(const std::vector<double> &a, const std::vector<double> &b2)
	{
		const int m = (int)a.size();
		double lo = a[0], hi = a[0];
		for (int i = 0; i < m; ++i)
		{
			const double r = (i > 0 ? std::sqrt(b2[i - 1]) : 0.0) + (i + 1 < m ? std::sqrt(b2[i]) : 0.0);
			lo = std::min(lo, a[i] - r), hi = std::max(hi, a[i] + r);
		}
		auto below = [&](double x) { // number of eigenvalues < x
			int c = 0;
			double d = a[0] - x;
			c += d < 0;
			for (int i = 1; i < m; ++i)
			{
				if (d == 0)
					d = 1e-300;
				d = a[i] - x - b2[i - 1] / d;
				c += d < 0;
			}
			return c;
		};
		for (int it = 0; it < 100 && hi - lo > 1e-12 * std::max(std::fabs(hi), 1.0); ++it)
		{
			const double mid = 0.5 * (lo + hi);
			if (below(mid) >= m)
				hi = mid;
			else
				lo = mid;
		}
		return hi;
	}

	// y = A_l x on level l (no Dirichlet mask or shift below level 0; level 0 always takes the slice format).  When the
	// hierarchy spans the ranks the halo entries of x are fetched from their owners first (x: a full local vector).
	pfx_status amg_matvec(pfx_ctx *ctx, pfx_amg_level &L, double *x, double *y)
	{
		pfx_amg *A = ctx->amg;
		if (A->dist)
		{
			pfx_status rc = level_exchange(ctx, A, L, x, L.m == ctx ? ctx->bs : (ctx->bs == 1 ? 1 : ctx->bs == 2 ? 3 : 6));
			if (rc)
				return rc;
		}
		if (!L.small)
		{
			pfx_status rc = pfx_sell_spmv(L.m, x, y, nullptr, nullptr, PFX_SPMV_ALL, nullptr, ctx->amg_params.mixed_precision != 0);
			if (rc)
				ctx->err = L.m->err;
			return rc;
		}
		pfx_ctx *s = L.m;
		if (s->n_owned <= 0)
			return PFX_OK;
		const unsigned grid = (unsigned)((s->n_owned * 64 + 255) / 256);
		if (s->bs == 1)
			hipLaunchKernelGGL((k_rowwave_spmv<1>), dim3(grid), dim3(256), 0, ctx->stream, s->d_node_ptr, s->d_node_col, s->d_values, x, s->n_owned, y);
		else if (s->bs == 2)
			hipLaunchKernelGGL((k_rowwave_spmv<2>), dim3(grid), dim3(256), 0, ctx->stream, s->d_node_ptr, s->d_node_col, s->d_values, x, s->n_owned, y);
		else
			hipLaunchKernelGGL((k_rowwave_spmv<3>), dim3(grid), dim3(256), 0, ctx->stream, s->d_node_ptr, s->d_node_col, s->d_values, x, s->n_owned, y);
		return PFX_OK;
	}

	// a scalar a kernel has just written, summed over the ranks (in place, on the stream)
	pfx_status dist_scalar(pfx_ctx *ctx, pfx_amg *A, double *d_s, int count = 1)
	{
		if (!A->dist)
			return PFX_OK;
		if (pfx_comm_all_reduce(A->dist->comm, ctx->stream, d_s, count))
			PFX_FAIL(PFX_ERR_HIP, pfx_comm_error(A->dist->comm));
		return PFX_OK;
	}

	// lmax of D^-1 A: `eig_iters` Lanczos steps on D^-1/2 A D^-1/2 from a fixed pseudo-random vector, largest Ritz value
	// plus 5 % (Ritz values approach the spectrum's end from inside; a Chebyshev interval that ends below lmax would
	// amplify the top modes).  Vectors: v (the one A is applied to) and t (A v) from the caller, three more allocated here.
	pfx_status amg_lambda_max(pfx_ctx *ctx, pfx_amg *A, pfx_amg_level &L, double *v, double *t, const double *dinv, int64_t n_rows)
	{
		hipStream_t st = ctx->stream;
		const uint8_t *fixed = (L.m == ctx && ctx->n_dirichlet > 0) ? ctx->d_fixed : nullptr;
		int steps = ctx->amg_params.power_iters;
		{
			// no more steps than the operator has rows (over a partition: than all ranks together have)
			double rows = (double)n_rows;
			pfx_status rc = dist_sum(ctx, A->dist, &rows, 1);
			if (rc)
				return rc;
			if ((double)steps > rows)
				steps = (int)rows;
		}
		if (steps < 1)
		{
			L.lmax = 1.05;
			return PFX_OK;
		}
		pfx_tmp<double> q, qp, w, scal;
		PFX_HIP(q.alloc((size_t)n_rows));
		PFX_HIP(qp.alloc((size_t)n_rows));
		PFX_HIP(w.alloc((size_t)n_rows));
		PFX_HIP(scal.alloc((size_t)(2 * steps + 1)));
		const int grid = grid_for(n_rows);
		PFX_HIP(hipMemsetAsync(q.p, 0, sizeof(double) * (size_t)n_rows, st));
		hipLaunchKernelGGL(k_seed_vector, dim3(grid), dim3(256), 0, st, n_rows, fixed, w.p);
		hipLaunchKernelGGL(k_dot_part, dim3(AMG_NPART), dim3(256), 0, st, w.p, w.p, n_rows, A->d_part);
		hipLaunchKernelGGL(k_dot_finish, dim3(1), dim3(256), 0, st, A->d_part, AMG_NPART, scal.p);
		pfx_status rcs = dist_scalar(ctx, A, scal.p);
		if (rcs)
			return rcs;
		hipLaunchKernelGGL(k_lz_c, dim3(grid), dim3(256), 0, st, dinv, w.p, scal.p, n_rows, qp.p, q.p, v);
		for (int j = 0; j < steps; ++j)
		{
			pfx_status rc = amg_matvec(ctx, L, v, t);
			if (rc)
				return rc;
			double *alpha = scal.p + 1 + 2 * j, *beta2 = alpha + 1;
			hipLaunchKernelGGL(k_lz_a, dim3(AMG_NPART), dim3(256), 0, st, dinv, t, qp.p, q.p, j > 0 ? (const double *)(alpha - 1) : (const double *)nullptr, n_rows, w.p,
							   A->d_part);
			hipLaunchKernelGGL(k_dot_finish, dim3(1), dim3(256), 0, st, A->d_part, AMG_NPART, alpha);
			if ((rc = dist_scalar(ctx, A, alpha)))
				return rc;
			hipLaunchKernelGGL(k_lz_b, dim3(AMG_NPART), dim3(256), 0, st, q.p, alpha, n_rows, w.p, A->d_part);
			hipLaunchKernelGGL(k_dot_finish, dim3(1), dim3(256), 0, st, A->d_part, AMG_NPART, beta2);
			if ((rc = dist_scalar(ctx, A, beta2)))
				return rc;
			hipLaunchKernelGGL(k_lz_c, dim3(grid), dim3(256), 0, st, dinv, w.p, beta2, n_rows, qp.p, q.p, v);
		}
		std::vector<double> h((size_t)(2 * steps + 1));
		PFX_HIP(hipMemcpyAsync(h.data(), scal.p, sizeof(double) * h.size(), hipMemcpyDeviceToHost, st));
		PFX_HIP(hipStreamSynchronize(st));
		std::vector<double> a, b2;
		for (int j = 0; j < steps; ++j)
		{
			a.push_back(h[1 + 2 * j]);
			if (!(h[2 + 2 * j] > 1e-300) || j + 1 == steps)
				break;
			b2.push_back(h[2 + 2 * j]);
		}
		double ritz = 1.0;
		if (!a.empty() && std::isfinite(a[0]))
			ritz = tridiagonal_lambda_max(a, b2);
		// constrained level-0 dofs carry the eigenvalue 1 of the identity rows: never report less
		L.lmax = 1.05 * std::max(ritz, 1.0);
		return PFX_OK;
	}

	// values_changed = false: only the diagonal shift of the level-0 operator changed (a regularisation retry of the
	// Newton solver).  The coarse matrices are Galerkin products of the UNSHIFTED matrix and stay as they are; what the
	// shift changes is the level-0 relaxation: its Jacobi diagonal (the caller's dinv0) and its spectral radius.
	pfx_status amg_numeric(pfx_ctx *ctx, double *dinv0, double *scratch_v, double *scratch_t, bool values_changed)
	{
		const double t_begin = now_s();
		pfx_amg *A = ctx->amg;
		hipStream_t st = ctx->stream;
		const uint8_t *fixed = ctx->n_dirichlet > 0 ? ctx->d_fixed : nullptr;
		for (size_t l = 0; values_changed && l + 1 < A->lv.size(); ++l)
		{
			pfx_amg_level &F = A->lv[l], &C = A->lv[l + 1];
			const unsigned grid = (unsigned)((2 * F.n_cpairs * 8 + 255) / 256);
			// (level 0 reads the caller's unmasked matrix: its constrained dofs are left out here; the coarse matrices have none)
			if (ctx->bs == 1)
				hipLaunchKernelGGL((k_galerkin_const<1>), dim3((unsigned)((F.n_cpairs + 255) / 256)), dim3(256), 0, st, F.d_cp_ptr, F.d_cp_src, F.d_cp_row, F.m->d_node_col,
								   F.m->d_values, l == 0 ? fixed : nullptr, F.n_cpairs, C.m->d_values);
			else if (ctx->bs == 2 && l == 0)
				hipLaunchKernelGGL((k_galerkin_plane<true>), dim3((unsigned)((F.n_cpairs + 255) / 256)), dim3(256), 0, st, F.d_cp_ptr, F.d_cp_src, F.d_cp_row, ctx->d_node_col,
								   ctx->d_values, F.d_dvec, fixed, F.n_cpairs, C.m->d_values);
			else if (ctx->bs == 2)
				hipLaunchKernelGGL((k_galerkin_plane<false>), dim3((unsigned)((F.n_cpairs + 255) / 256)), dim3(256), 0, st, F.d_cp_ptr, F.d_cp_src, F.d_cp_row, F.m->d_node_col,
								   F.m->d_values, F.d_dvec, (const uint8_t *)nullptr, F.n_cpairs, C.m->d_values);
			else if (l == 0)
				hipLaunchKernelGGL((k_galerkin<true>), dim3(grid), dim3(256), 0, st, F.d_cp_ptr, F.d_cp_src, F.d_cp_row, (const int64_t *)nullptr, ctx->d_node_col, ctx->d_values, F.d_dvec,
								   fixed, C.d_gptr, F.d_cp_arow, F.n_cpairs, C.m->d_values);
			else
				hipLaunchKernelGGL((k_galerkin<false>), dim3(grid), dim3(256), 0, st, F.d_cp_ptr, F.d_cp_src, F.d_cp_row, F.d_gptr, F.d_gcol, F.m->d_values, F.d_dvec,
								   (const uint8_t *)nullptr, C.d_gptr, F.d_cp_arow, F.n_cpairs, C.m->d_values);
			C.m->values_gen++;
			// the solver copy of a coarse matrix serves the preconditioner's products only: single precision unless asked otherwise
			pfx_status rc = C.small ? PFX_OK : (ctx->amg_params.mixed_precision ? pfx_sell_refresh32(C.m) : pfx_sell_refresh(C.m));
			if (!rc)
				rc = pfx_launch_invdiag(C.m, 1, C.d_dinv);
			if (rc)
			{
				ctx->err = C.m->err;
				return rc;
			}
		}
		PFX_HIP(hipGetLastError());
		// spectral radii of D^-1 A on every level that is relaxed
		for (size_t l = 0; l + 1 < A->lv.size() && (values_changed || l == 0); ++l)
		{
			pfx_amg_level &L = A->lv[l];
			if (l == 0 && ctx->n_owned < ctx->n_nodes) // the level-0 operator is the owned-owned block: halo entries of the vector it is applied to are zero
				PFX_HIP(hipMemsetAsync(scratch_v + ctx->n_owned * ctx->bs, 0, sizeof(double) * (size_t)((ctx->n_nodes - ctx->n_owned) * ctx->bs), st));
			pfx_status rc = (l == 0) ? amg_lambda_max(ctx, A, L, scratch_v, scratch_t, dinv0, ctx->n_owned * ctx->bs) : amg_lambda_max(ctx, A, L, L.d_x, L.d_t, L.d_dinv, L.n_odofs);
			if (rc)
				return rc;
		}
		// dense inverse of the coarsest matrix (host: Gauss-Jordan with partial pivoting)
		if (values_changed)
		{
			pfx_amg_level &L = A->lv.back();
			if (A->lv.size() < 2)
			{
				A->n_coarsest = 0;
			}
			else
			{
				pfx_ctx *s = L.m;
				// The hierarchy can end early (max_levels reached, coarsening stalled) with a last level far above
				// coarsest_dofs; the dense inverse below is O(n^2) memory and O(n^3) host work: refuse instead.
				// Over a partition the coarsest matrix is gathered: every rank places the rows of its own aggregates at its
				// offset in the global numbering (columns: the global ids of its coarse vertices, those of the halo ones
				// learned from their owners), the sum over the ranks is the whole matrix, and every rank inverts it.
				pfx_dist *dist = A->dist;
				const int per = ctx->bs == 1 ? 1 : ctx->bs == 2 ? 3 : 6; // dofs per coarse vertex
				double total = (double)L.n_odofs;
				{
					pfx_status rc = dist_sum(ctx, dist, &total, 1);
					if (rc)
						return rc;
				}
				if (total > (double)AMG_MAX_DENSE)
					PFX_FAIL(PFX_ERR_SIZE, "AMG: the coarsest level has " + std::to_string((long long)total) + " dofs, more than the dense coarse solve takes ("
											   + std::to_string(AMG_MAX_DENSE) + "): raise max_levels, or the coarsening stalled on this matrix");
				const int n = (int)total;
				std::vector<int64_t> gvert((size_t)L.n_vert); // global id of every local coarse vertex
				for (int64_t v = 0; v < L.n_overt; ++v)
					gvert[(size_t)v] = A->coarse_off / per + v;
				if (dist && L.n_vert > L.n_overt)
				{
					pfx_tmp<double> gid;
					PFX_HIP(gid.alloc((size_t)L.n_vert));
					std::vector<double> hg((size_t)L.n_vert, -1.0);
					for (int64_t v = 0; v < L.n_overt; ++v)
						hg[(size_t)v] = (double)gvert[(size_t)v];
					PFX_HIP(hipMemcpyAsync(gid.p, hg.data(), sizeof(double) * hg.size(), hipMemcpyHostToDevice, st));
					pfx_status rc = level_exchange(ctx, A, L, gid.p, 1);
					if (rc)
						return rc;
					PFX_HIP(hipMemcpyAsync(hg.data(), gid.p, sizeof(double) * hg.size(), hipMemcpyDeviceToHost, st));
					PFX_HIP(hipStreamSynchronize(st));
					for (int64_t v = L.n_overt; v < L.n_vert; ++v)
						gvert[(size_t)v] = (int64_t)hg[(size_t)v];
				}
				else if (dist)
				{
					// (every rank takes part in the exchange, also one without halo vertices on this level)
					pfx_tmp<double> gid;
					PFX_HIP(gid.alloc((size_t)(L.n_vert > 0 ? L.n_vert : 1)));
					pfx_status rc = level_exchange(ctx, A, L, gid.p, 1);
					if (rc)
						return rc;
				}
				std::vector<double> vals((size_t)s->nnz);
				PFX_HIP(hipMemcpyAsync(vals.data(), s->d_values, sizeof(double) * vals.size(), hipMemcpyDeviceToHost, st));
				PFX_HIP(hipStreamSynchronize(st));
				std::vector<double> M((size_t)n * n, 0.0), I((size_t)n * n, 0.0);
				const int64_t nv = L.n_overt; // rows: the aggregates this rank owns
				// the coarsest level's vertex graph, read back for the dense matrix
				std::vector<int64_t> h_gptr((size_t)nv + 1);
				PFX_HIP(hipMemcpyAsync(h_gptr.data(), L.d_gptr, sizeof(int64_t) * h_gptr.size(), hipMemcpyDeviceToHost, st));
				PFX_HIP(hipStreamSynchronize(st));
				std::vector<int32_t> h_gcol((size_t)h_gptr[(size_t)nv]);
				if (!h_gcol.empty())
					PFX_HIP(hipMemcpyAsync(h_gcol.data(), L.d_gcol, sizeof(int32_t) * h_gcol.size(), hipMemcpyDeviceToHost, st));
				PFX_HIP(hipStreamSynchronize(st));
				double dmax = 0;
				for (int64_t a = 0; a < nv; ++a)
				{
					const int64_t g0 = h_gptr[(size_t)a], deg = h_gptr[(size_t)a + 1] - g0;
					const int64_t ga = gvert[(size_t)a];
					if (ctx->bs < 3)
					{
						const int cb = per; // one node of block size `per` per coarse vertex
						for (int64_t k = 0; k < deg; ++k)
							for (int m = 0; m < cb; ++m)
								for (int q = 0; q < cb; ++q)
									M[(size_t)((ga * cb + m) * n + gvert[(size_t)h_gcol[(size_t)(g0 + k)]] * cb + q)] = vals[(size_t)((g0 + k) * cb * cb + m * cb + q)];
						continue;
					}
					for (int64_t k = 0; k < deg; ++k)
						for (int al = 0; al < 2; ++al)
							for (int be = 0; be < 2; ++be)
							{
								const double *blk = vals.data() + (4 * g0 + al * 2 * deg + 2 * k + be) * 9;
								const int64_t r0 = (2 * ga + al) * 3, c0 = (2 * gvert[(size_t)h_gcol[(size_t)(g0 + k)]] + be) * 3;
								for (int m = 0; m < 3; ++m)
									for (int q = 0; q < 3; ++q)
										M[(size_t)((r0 + m) * n + c0 + q)] = blk[3 * m + q];
							}
				}
				if (dist)
				{
					// the rows of the other ranks: sum of everybody's contribution (each row is written by exactly one rank)
					pfx_tmp<double> dM;
					PFX_HIP(dM.alloc((size_t)n * n));
					PFX_HIP(hipMemcpyAsync(dM.p, M.data(), sizeof(double) * M.size(), hipMemcpyHostToDevice, st));
					if (pfx_comm_all_reduce(dist->comm, st, dM.p, n * n))
						PFX_FAIL(PFX_ERR_HIP, pfx_comm_error(dist->comm));
					PFX_HIP(hipMemcpyAsync(M.data(), dM.p, sizeof(double) * M.size(), hipMemcpyDeviceToHost, st));
					PFX_HIP(hipStreamSynchronize(st));
				}
				for (int k = 0; k < n; ++k)
					dmax = std::max(dmax, std::fabs(M[(size_t)k * n + k]));
				for (int k = 0; k < n; ++k)
				{
					// a coarse dof without any stiffness (rotation of a degenerate aggregate) is decoupled; a tiny shift
					// keeps nearly dependent rigid-body modes from blowing up the inverse
					if (M[(size_t)k * n + k] == 0.0)
						M[(size_t)k * n + k] = dmax > 0 ? dmax : 1.0;
					else
						M[(size_t)k * n + k] += 1e-12 * dmax;
					I[(size_t)k * n + k] = 1.0;
				}
				for (int c = 0; c < n; ++c)
				{
					int piv = c;
					for (int r = c + 1; r < n; ++r)
						if (std::fabs(M[(size_t)r * n + c]) > std::fabs(M[(size_t)piv * n + c]))
							piv = r;
					if (M[(size_t)piv * n + c] == 0.0)
						PFX_FAIL(PFX_ERR_NUMERIC, "AMG: the coarsest matrix is singular");
					if (piv != c)
						for (int k = 0; k < n; ++k)
						{
							std::swap(M[(size_t)piv * n + k], M[(size_t)c * n + k]);
							std::swap(I[(size_t)piv * n + k], I[(size_t)c * n + k]);
						}
					const double inv = 1.0 / M[(size_t)c * n + c];
					for (int k = 0; k < n; ++k)
					{
						M[(size_t)c * n + k] *= inv;
						I[(size_t)c * n + k] *= inv;
					}
					for (int r = 0; r < n; ++r)
					{
						if (r == c)
							continue;
						const double f = M[(size_t)r * n + c];
						if (f == 0.0)
							continue;
						double *mr = &M[(size_t)r * n], *mc = &M[(size_t)c * n], *ir = &I[(size_t)r * n], *ic = &I[(size_t)c * n];
						for (int k = 0; k < n; ++k)
						{
							mr[k] -= f * mc[k];
							ir[k] -= f * ic[k];
						}
					}
				}
				if (A->n_coarsest != n)
				{
					if (pfx_alloc(ctx, &A->d_ainv, (int64_t)n * n) || pfx_alloc(ctx, &A->d_bglob, (int64_t)n))
						return PFX_ERR_HIP;
					A->n_coarsest = n;
				}
				PFX_HIP(hipMemcpyAsync(A->d_ainv, I.data(), sizeof(double) * I.size(), hipMemcpyHostToDevice, st));
				PFX_HIP(hipStreamSynchronize(st));
			}
		}
		A->values_gen = ctx->values_gen;
		A->shift = ctx->diag_shift;
		A->numeric_s = now_s() - t_begin;
		return PFX_OK;
	}

	struct relax_vecs
	{
		const double *b;
		double *x, *t, *r, *dv;
		const double *dinv;
		int64_t n;
	};

	// Chebyshev relaxation of degree `deg` for D^-1 A on [lower * lmax, lmax]; zero_guess: x is overwritten
	pfx_status amg_relax(pfx_ctx *ctx, pfx_amg_level &L, const relax_vecs &v, bool zero_guess)
	{
		hipStream_t st = ctx->stream;
		const pfx_amg_params &par = ctx->amg_params;
		const int deg = par.relax_degree < 1 ? 1 : par.relax_degree;
		const double lmax = L.lmax, lmin = L.lmax * par.relax_lower;
		const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin);
		const double sigma = theta / delta;
		double rho = 1.0 / sigma;
		const int grid = grid_for(v.n);
		if (!zero_guess)
		{
			pfx_status rc = amg_matvec(ctx, L, v.x, v.t);
			if (rc)
				return rc;
		}
		hipLaunchKernelGGL(k_cheb_first, dim3(grid), dim3(256), 0, st, v.dinv, v.b, v.t, 1.0 / theta, zero_guess ? 1 : 0, v.n, deg > 1 ? v.r : nullptr, v.dv, v.x);
		for (int k = 1; k < deg; ++k)
		{
			pfx_status rc = amg_matvec(ctx, L, v.dv, v.t);
			if (rc)
				return rc;
			const double rho_new = 1.0 / (2.0 * sigma - rho);
			hipLaunchKernelGGL(k_cheb_next, dim3(grid), dim3(256), 0, st, v.dinv, v.t, rho_new * rho, 2.0 * rho_new / delta, v.n, v.r, v.dv, v.x);
			rho = rho_new;
		}
		return PFX_OK;
	}

	// x = M_l^-1 b on level l (b, x: vectors of that level; level 0: the caller's)
	pfx_status amg_cycle(pfx_ctx *ctx, pfx_amg *A, size_t l, const double *b, double *x, double *t0, const double *dinv0)
	{
		hipStream_t st = ctx->stream;
		pfx_amg_level &L = A->lv[l];
		if (l + 1 == A->lv.size())
		{
			if (!A->dist)
			{
				hipLaunchKernelGGL(k_dense_matvec, dim3(A->n_coarsest), dim3(64), 0, st, A->d_ainv, b, A->n_coarsest, x);
				return PFX_OK;
			}
			// gathered right-hand side (every rank adds its part at its offset), then the rows of the inverse that belong to
			// this rank's dofs
			PFX_HIP(hipMemsetAsync(A->d_bglob, 0, sizeof(double) * (size_t)A->n_coarsest, st));
			if (L.n_odofs > 0)
				PFX_HIP(hipMemcpyAsync(A->d_bglob + A->coarse_off, b, sizeof(double) * (size_t)L.n_odofs, hipMemcpyDeviceToDevice, st));
			pfx_status rc = dist_scalar(ctx, A, A->d_bglob, A->n_coarsest);
			if (rc)
				return rc;
			if (L.n_odofs > 0)
				hipLaunchKernelGGL(k_dense_matvec, dim3((unsigned)L.n_odofs), dim3(64), 0, st, A->d_ainv + A->coarse_off * A->n_coarsest, A->d_bglob, A->n_coarsest, x);
			return PFX_OK;
		}
		const bool level0 = l == 0;
		const int64_t n = level0 ? ctx->n_owned * ctx->bs : L.n_odofs;
		const uint8_t *fixed = (level0 && ctx->n_dirichlet > 0) ? ctx->d_fixed : nullptr;
		relax_vecs v{b, x, level0 ? t0 : L.d_t, L.d_r, L.d_dv, level0 ? dinv0 : L.d_dinv, n};
		pfx_status rc = amg_relax(ctx, L, v, true);
		if (rc)
			return rc;
		// coarse correction
		if ((rc = amg_matvec(ctx, L, x, v.t)))
			return rc;
		pfx_amg_level &C = A->lv[l + 1];
		if (ctx->bs == 1)
			hipLaunchKernelGGL((k_restrict_const<1>), dim3(grid_for(L.n_agg)), dim3(256), 0, st, L.d_mptr, L.d_mlist, b, v.t, fixed, L.n_agg, C.d_b);
		else if (ctx->bs == 2 && level0)
			hipLaunchKernelGGL((k_restrict_plane<true>), dim3(grid_for(L.n_agg)), dim3(256), 0, st, L.d_mptr, L.d_mlist, L.d_dvec, b, v.t, fixed, L.n_agg, C.d_b);
		else if (ctx->bs == 2)
			hipLaunchKernelGGL((k_restrict_plane<false>), dim3(grid_for(L.n_agg)), dim3(256), 0, st, L.d_mptr, L.d_mlist, L.d_dvec, b, v.t, fixed, L.n_agg, C.d_b);
		else if (level0)
			hipLaunchKernelGGL((k_restrict<true>), dim3(grid_for(L.n_agg * 16)), dim3(256), 0, st, L.d_mptr, L.d_mlist, L.d_dvec, b, v.t, L.n_agg, C.d_b);
		else
			hipLaunchKernelGGL((k_restrict<false>), dim3(grid_for(L.n_agg * 16)), dim3(256), 0, st, L.d_mptr, L.d_mlist, L.d_dvec, b, v.t, L.n_agg, C.d_b);
		if ((rc = amg_cycle(ctx, A, l + 1, C.d_b, C.d_x, nullptr, nullptr)))
			return rc;
		if (ctx->amg_params.ncycle > 1 && l + 2 < A->lv.size())
		{
			// W cycle: a second coarse cycle on the coarse residual
			for (int rep = 1; rep < ctx->amg_params.ncycle; ++rep)
			{
				if ((rc = amg_matvec(ctx, C, C.d_x, C.d_t)))
					return rc;
				hipLaunchKernelGGL(k_vec_sub, dim3(grid_for(C.n_odofs)), dim3(256), 0, st, C.d_b, C.d_t, C.n_odofs, C.d_b2);
				if ((rc = amg_cycle(ctx, A, l + 1, C.d_b2, C.d_x2, nullptr, nullptr)))
					return rc;
				hipLaunchKernelGGL(k_vec_add, dim3(grid_for(C.n_odofs)), dim3(256), 0, st, C.d_x2, C.n_odofs, C.d_x);
			}
		}
		const double omega = ctx->amg_params.coarse_scale;
		if (ctx->bs == 1)
			hipLaunchKernelGGL((k_prolong_const<1>), dim3(grid_for(level0 ? ctx->n_owned : L.n_overt)), dim3(256), 0, st, L.d_agg, C.d_x, fixed, level0 ? ctx->n_owned : L.n_overt, omega, x);
		else if (ctx->bs == 2 && level0)
			hipLaunchKernelGGL((k_prolong_plane<true>), dim3(grid_for(ctx->n_owned)), dim3(256), 0, st, L.d_agg, L.d_dvec, C.d_x, fixed, ctx->n_owned, omega, x);
		else if (ctx->bs == 2)
			hipLaunchKernelGGL((k_prolong_plane<false>), dim3(grid_for(L.n_overt)), dim3(256), 0, st, L.d_agg, L.d_dvec, C.d_x, fixed, L.n_overt, omega, x);
		else if (level0)
			hipLaunchKernelGGL((k_prolong<true>), dim3(grid_for(ctx->n_owned)), dim3(256), 0, st, L.d_agg, L.d_dvec, C.d_x, fixed, ctx->n_owned, omega, x);
		else
			hipLaunchKernelGGL((k_prolong<false>), dim3(grid_for(L.n_overt)), dim3(256), 0, st, L.d_agg, L.d_dvec, C.d_x, fixed, L.n_overt, omega, x);
		return amg_relax(ctx, L, v, false);
	}
} // namespace

void pfx_amg_release(pfx_ctx *ctx)
{
	pfx_amg *A = ctx->amg;
	if (!A)
		return;
	for (auto &L : A->lv)
		free_level(ctx, L);
	free_dev(ctx, A->d_ainv);
	free_dev(ctx, A->d_part);
	free_dev(ctx, A->d_scal);
	free_dev(ctx, A->d_z);
	free_dev(ctx, A->d_r0);
	free_dev(ctx, A->d_dv0);
	free_dev(ctx, A->d_pos);
	free_dev(ctx, A->d_state);
	free_dev(ctx, A->d_bglob);
	delete A;
	ctx->amg = nullptr;
}

// hierarchy for the current pattern / Dirichlet set (symbolic) and matrix values (numeric); dinv0 = Jacobi diagonal of
// the level-0 operator (n_dofs doubles), scratch_v / scratch_t: two level-0 vectors that may be overwritten
pfx_status pfx_amg_prepare(pfx_ctx *ctx, double *dinv0, double *scratch_v, double *scratch_t)
{
	if (ctx->bs < 1 || ctx->bs > 3)
		PFX_FAIL(PFX_ERR_UNSUPPORTED, "the AMG preconditioner is implemented for block sizes 1, 2 and 3");
	pfx_status rc = pfx_sell_refresh(ctx);
	if (rc)
		return rc;
	if (ctx->amg_params.mixed_precision && (rc = pfx_sell_refresh32(ctx)))
		return rc;
	if (!ctx->amg || ctx->amg->struct_gen != ctx->struct_gen || ctx->amg->dist != ctx->amg_dist)
		if ((rc = amg_symbolic(ctx)))
			return rc;
	// the Jacobi diagonal lives in the solver's workspace, which other solves overwrite: always rebuilt
	if ((rc = pfx_launch_invdiag(ctx, 1, dinv0)))
		return rc;
	const bool values_changed = ctx->amg->values_gen != ctx->values_gen;
	if (values_changed || ctx->amg->shift != ctx->diag_shift)
		if ((rc = amg_numeric(ctx, dinv0, scratch_v, scratch_t, values_changed)))
			return rc;
	return PFX_OK;
}

// z = M^-1 r (level-0 vectors; t0: scratch, dinv0: Jacobi diagonal)
pfx_status pfx_amg_apply(pfx_ctx *ctx, const double *d_r, double *d_z, double *t0, const double *dinv0)
{
	pfx_scoped_timer tm(ctx, "precond");
	pfx_amg *A = ctx->amg;
	if (A->lv.size() < 2)
	{
		// nothing to coarsen (every node constrained): plain Jacobi
		hipLaunchKernelGGL(k_cheb_first, dim3(grid_for(ctx->n_owned * ctx->bs)), dim3(256), 0, ctx->stream, dinv0, d_r, (const double *)nullptr, 1.0, 1, ctx->n_owned * ctx->bs,
						   (double *)nullptr, (double *)nullptr, d_z);
		return PFX_OK;
	}
	pfx_status rc = amg_cycle(ctx, A, 0, d_r, d_z, t0, dinv0);
	if (rc)
		return rc;
	PFX_HIP(hipGetLastError());
	return PFX_OK;
}

namespace
{
	// part[0] = *s, part[1 ..] = 0: an all-reduced scalar put back where the kernels expect per-block partial sums
	__global__ void k_spread_scalar(const double *__restrict__ s, double *__restrict__ part)
	{
		for (int k = threadIdx.x; k < AMG_NPART; k += blockDim.x)
			part[k] = k == 0 ? *s : 0.0;
	}

	// partial-sum arrays q0 .. q0 + nq - 1 of the PCG kernels -> their sums over the ranks (no-op on one rank)
	pfx_status reduce_partials(pfx_ctx *ctx, pfx_amg *A, int q0, int nq)
	{
		if (!A->dist)
			return PFX_OK;
		hipStream_t st = ctx->stream;
		for (int q = q0; q < q0 + nq; ++q)
			hipLaunchKernelGGL(k_dot_finish, dim3(1), dim3(256), 0, st, A->d_part + q * AMG_NPART, AMG_NPART, A->d_scal + 4 + q);
		pfx_status rc = dist_scalar(ctx, A, A->d_scal + 4 + q0, nq);
		if (rc)
			return rc;
		for (int q = q0; q < q0 + nq; ++q)
			hipLaunchKernelGGL(k_spread_scalar, dim3(1), dim3(256), 0, st, A->d_scal + 4 + q, A->d_part + q * AMG_NPART);
		return PFX_OK;
	}

	// Preconditioned CG in Eigen's operation order with z = M^-1 r from the hierarchy.  dist = null: one context, the whole
	// mesh.  Otherwise the iteration over a partition: the halo entries of the search direction travel while the interior
	// rows are multiplied, the scalars are all-reduced (two all-reduces per iteration), and the hierarchy spans the ranks.
	pfx_status pcg_amg_impl(pfx_ctx *ctx, pfx_dist *dist, const double *d_b, double *d_x, int max_iter, double tol, int check_every, int *iters, double *rel_res)
	{
		hipStream_t st = ctx->stream;
		const int64_t n = ctx->n_owned * ctx->bs;
		const int64_t nfull = ctx->n_dofs();
		const uint8_t *fixed = ctx->n_dirichlet > 0 ? ctx->d_fixed : nullptr;
		if (!dist && ctx->n_owned != ctx->n_nodes)
			PFX_FAIL(PFX_ERR_UNSUPPORTED, "this loop has no halo exchange: on a partitioned mesh use pfx_dist_cg_dev, or drive the iteration and apply the preconditioner with pfx_amg_apply_dev");
		if (ctx->d_work_ext)
			PFX_FAIL(PFX_ERR_UNSUPPORTED, "the AMG-preconditioned solve uses the context's own workspace (clear pfx_cgs_set_workspace first)");
		if (!ctx->d_work && pfx_alloc(ctx, &ctx->d_work, 4 * nfull))
			return PFX_ERR_HIP;
		double *r = ctx->d_work, *p = ctx->d_work + nfull, *t = ctx->d_work + 2 * nfull, *dinv = ctx->d_work + 3 * nfull;
		pfx_status rc = pfx_amg_prepare(ctx, dinv, p, t);
		if (rc)
			return rc;
		pfx_amg *A = ctx->amg;
		pfx_comm *comm = dist ? dist->comm : nullptr;
		auto comm_fail = [&]() {
			ctx->err = pfx_comm_error(comm);
			return PFX_ERR_HIP;
		};
		if (ctx->part0_count < pfx_sell_grid(ctx))
		{
			if (pfx_alloc(ctx, &ctx->d_part0, pfx_sell_grid(ctx)))
				return PFX_ERR_HIP;
			ctx->part0_count = pfx_sell_grid(ctx);
		}
		pcg_state *state = A->d_state;
		PFX_HIP(hipMemsetAsync(state, 0, sizeof(pcg_state), st));
		if (fixed)
			hipLaunchKernelGGL(k_mask, dim3(grid_for(n)), dim3(256), 0, st, d_x, fixed, n);
		if (dist && (pfx_halo_start(comm, &dist->halo, st, d_x, ctx->bs) || pfx_halo_finish(comm, &dist->halo, st)))
			return comm_fail();
		if ((rc = pfx_sell_spmv(ctx, d_x, t, nullptr, nullptr)))
			return rc;
		hipLaunchKernelGGL(k_pcg_init, dim3(AMG_NPART), dim3(256), 0, st, d_b, t, fixed, n, r, A->d_part);
		if ((rc = reduce_partials(ctx, A, 0, 2)))
			return rc;
		hipLaunchKernelGGL(k_pcg_init2, dim3(1), dim3(256), 0, st, A->d_part, tol, n, d_x, state);
		// z = M^-1 r ; p = z ; abs_new = r.z
		if ((rc = pfx_amg_apply(ctx, r, A->d_z, t, dinv)))
			return rc;
		hipLaunchKernelGGL(k_dot_part, dim3(AMG_NPART), dim3(256), 0, st, r, A->d_z, n, A->d_part + 2 * AMG_NPART);
		if ((rc = reduce_partials(ctx, A, 2, 1)))
			return rc;
		hipLaunchKernelGGL(k_pcg_direction, dim3(AMG_NPART), dim3(256), 0, st, A->d_z, A->d_part, 1, n, p, state);
		hipLaunchKernelGGL(k_pcg_commit, dim3(1), dim3(1), 0, st, state);
		if (check_every <= 0)
			check_every = 8;
		int it = 0;
		pcg_state h;
		memset(&h, 0, sizeof(h));
		PFX_HIP(hipMemcpyAsync(&h, state, sizeof(h), hipMemcpyDeviceToHost, st));
		PFX_HIP(hipStreamSynchronize(st));
		while (it < max_iter && !h.done)
		{
			const int stop = (it + check_every < max_iter) ? it + check_every : max_iter;
			for (; it < stop; ++it)
			{
				if (dist)
				{
					// the rows that need no halo entry of p are multiplied while the exchange is in flight
					bool sliced = false;
					if (pfx_halo_start(comm, &dist->halo, st, p, ctx->bs))
						return comm_fail();
					if ((rc = pfx_sell_spmv(ctx, p, t, ctx->d_part0, &state->done, PFX_SPMV_INTERIOR, &sliced)))
						return rc;
					if (pfx_halo_finish(comm, &dist->halo, st))
						return comm_fail();
					if ((rc = pfx_sell_spmv(ctx, p, t, ctx->d_part0, &state->done, PFX_SPMV_BOUNDARY, &sliced)))
						return rc;
					if (sliced && (rc = pfx_sell_unpermute(ctx, t)))
						return rc;
				}
				else if ((rc = pfx_sell_spmv(ctx, p, t, ctx->d_part0, &state->done)))
					return rc;
				pfx_launch_sum_partials(ctx, ctx->d_part0, pfx_sell_grid(ctx), A->d_scal);
				if ((rc = dist_scalar(ctx, A, A->d_scal)))
					return rc;
				hipLaunchKernelGGL(k_pcg_update, dim3(AMG_NPART), dim3(256), 0, st, p, t, A->d_scal, n, d_x, r, A->d_part, state);
				if ((rc = pfx_amg_apply(ctx, r, A->d_z, t, dinv)))
					return rc;
				hipLaunchKernelGGL(k_dot_part, dim3(AMG_NPART), dim3(256), 0, st, r, A->d_z, n, A->d_part + 2 * AMG_NPART);
				if ((rc = reduce_partials(ctx, A, 1, 2))) // r.r and r.z in one all-reduce
					return rc;
				hipLaunchKernelGGL(k_pcg_test, dim3(1), dim3(256), 0, st, A->d_part, it, state);
				hipLaunchKernelGGL(k_pcg_direction, dim3(AMG_NPART), dim3(256), 0, st, A->d_z, A->d_part, 0, n, p, state);
				hipLaunchKernelGGL(k_pcg_commit, dim3(1), dim3(1), 0, st, state);
			}
			PFX_HIP(hipMemcpyAsync(&h, state, sizeof(h), hipMemcpyDeviceToHost, st));
			PFX_HIP(hipStreamSynchronize(st));
		}
		if (dist && (pfx_halo_start(comm, &dist->halo, st, d_x, ctx->bs) || pfx_halo_finish(comm, &dist->halo, st)))
			return comm_fail();
		PFX_HIP(hipGetLastError());
		*iters = h.iters;
		*rel_res = (h.rhs_norm2 == 0.0) ? 0.0 : std::sqrt(h.rr / h.rhs_norm2);
		return PFX_OK;
	}
} // namespace

pfx_status pfx_pcg_amg_run(pfx_ctx *ctx, const double *d_b, double *d_x, int max_iter, double tol, int *iters, double *rel_res)
{
	ctx->amg_dist = nullptr;
	return pcg_amg_impl(ctx, nullptr, d_b, d_x, max_iter, tol, 8, iters, rel_res);
}

// the same over a partition (pfx_dist_cg_dev with PFX_PRECOND_AMG): the hierarchy spans the ranks while this call runs
pfx_status pfx_dist_pcg_amg_run(pfx_dist *dist, const double *d_b, double *d_x, int max_iter, double tol, int check_every, int *iters, double *rel_res)
{
	pfx_ctx *ctx = dist->ctx;
	ctx->amg_dist = pfx_comm_world(dist->comm) > 1 ? dist : nullptr;
	const pfx_status rc = pcg_amg_impl(ctx, ctx->amg_dist, d_b, d_x, max_iter, tol, check_every > 0 ? (check_every < 8 ? check_every : 8) : 8, iters, rel_res);
	ctx->amg_dist = nullptr; // calls on the context alone (pfx_amg_apply_dev ...) are again about its own block
	if (rc)
		dist->err = ctx->err;
	return rc;
}

// the hierarchy holds the partition's communicator: gone before the partition object is
void pfx_dist_amg_release(pfx_dist *dist)
{
	pfx_ctx *ctx = dist->ctx;
	if (ctx->amg && ctx->amg->dist == dist)
		pfx_amg_release(ctx);
	if (ctx->amg_dist == dist)
		ctx->amg_dist = nullptr;
}

namespace
{
	// the solver's four vectors [r | p | t | 1/diag]: the caller's workspace (pfx_cgs_set_workspace) or the context's own
	double *amg_workspace(pfx_ctx *ctx)
	{
		if (ctx->d_work_ext)
			return ctx->d_work_ext;
		if (!ctx->d_work && pfx_alloc(ctx, &ctx->d_work, 4 * ctx->n_dofs()))
			return nullptr;
		return ctx->d_work;
	}
} // namespace

// one application of the hierarchy that spans the ranks (d_r, d_z: local vectors [owned | halo]; the halo part of d_z is current on return)
pfx_status pfx_dist_amg_apply_run(pfx_dist *dist, const double *d_r, double *d_z)
{
	pfx_ctx *ctx = dist->ctx;
	if (ctx->d_work_ext)
		PFX_FAIL(PFX_ERR_UNSUPPORTED, "clear pfx_cgs_set_workspace first");
	ctx->amg_dist = pfx_comm_world(dist->comm) > 1 ? dist : nullptr;
	const int64_t nfull = ctx->n_dofs();
	double *base = amg_workspace(ctx);
	pfx_status rc = base ? pfx_amg_prepare(ctx, base + 3 * nfull, base + nfull, base + 2 * nfull) : PFX_ERR_HIP;
	if (!rc)
		rc = pfx_amg_apply(ctx, d_r, d_z, base + 2 * nfull, base + 3 * nfull);
	if (!rc && ctx->amg_dist && ctx->amg->lv.size() >= 2)
		rc = level_exchange(ctx, ctx->amg, ctx->amg->lv[0], d_z, ctx->bs);
	ctx->amg_dist = nullptr;
	if (rc)
		dist->err = ctx->err;
	return rc;
}

extern "C"
{
	void pfx_amg_default_params(pfx_amg_params *p)
	{
		if (!p)
			return;
		p->aggregate_passes = 4;
		p->relax_degree = 1;
		p->ncycle = 2;
		p->max_levels = 12;
		p->coarsest_dofs = 600;
		p->power_iters = 15;
		p->relax_lower = 0.1;
		p->coarse_scale = 1.6;
		p->mixed_precision = 1;
	}

	pfx_status pfx_amg_set_params(pfx_ctx *ctx, const pfx_amg_params *p)
	{
		if (!ctx || !p)
			return PFX_ERR_ARG;
		if (p->aggregate_passes < 1 || p->aggregate_passes > 8 || p->relax_degree < 1 || p->relax_degree > 16 || p->ncycle < 1 || p->ncycle > 4 || p->max_levels < 2
			|| p->coarsest_dofs < 6 || p->coarsest_dofs > 3000 || p->power_iters < 1 || !(p->relax_lower > 0.0 && p->relax_lower < 1.0) || !(p->coarse_scale >= 1.0 && p->coarse_scale < 2.0) || (p->mixed_precision != 0 && p->mixed_precision != 1))
			PFX_FAIL(PFX_ERR_ARG, "pfx_amg_set_params: parameter out of range");
		ctx->amg_params = *p;
		PFX_HIP(hipSetDevice(ctx->device));
		pfx_amg_release(ctx); // the hierarchy depends on the parameters
		return PFX_OK;
	}

	pfx_status pfx_amg_setup(pfx_ctx *ctx)
	{
		if (!ctx)
			return PFX_ERR_ARG;
		if (!ctx->d_node_ptr)
			PFX_FAIL(PFX_ERR_ARG, "pfx_build_pattern must be called first");
		if (ctx->ext && !ctx->ext_factorized)
			PFX_FAIL(PFX_ERR_ARG, "pfx_solver_factorize must succeed before the matrix is used");
		PFX_HIP(hipSetDevice(ctx->device));
		const int64_t nfull = ctx->n_dofs();
		double *base = amg_workspace(ctx);
		if (!base)
			return PFX_ERR_HIP;
		return pfx_amg_prepare(ctx, base + 3 * nfull, base + nfull, base + 2 * nfull);
	}

	pfx_status pfx_tune_placement(pfx_ctx *ctx)
	{
		if (!ctx)
			return PFX_ERR_ARG;
		if (!ctx->d_node_ptr || !ctx->d_values)
			PFX_FAIL(PFX_ERR_ARG, "no matrix: assemble or pfx_solver_factorize first");
		if (ctx->ext && !ctx->ext_factorized)
			PFX_FAIL(PFX_ERR_ARG, "pfx_solver_factorize must succeed before the matrix is used");
		PFX_HIP(hipSetDevice(ctx->device));
		pfx_status rc = pfx_sell_refresh(ctx);
		if (rc || (rc = pfx_sell_place(ctx)))
			return rc;
		// the single-precision copies of an existing hierarchy: level 0 and every level that multiplies through the slice format
		pfx_amg *A = ctx->amg;
		if (A && ctx->amg_params.mixed_precision)
			for (auto &L : A->lv)
				if (!L.small && L.m && L.m->d_sell_vals32 && (rc = pfx_sell_place32(L.m)))
				{
					if (L.m != ctx)
						ctx->err = L.m->err;
					return rc;
				}
		return PFX_OK;
	}

	pfx_status pfx_amg_get_info(pfx_ctx *ctx, pfx_amg_info *info)
	{
		if (!ctx || !info)
			return PFX_ERR_ARG;
		memset(info, 0, sizeof(*info));
		pfx_amg *A = ctx->amg;
		if (!A)
			PFX_FAIL(PFX_ERR_ARG, "no AMG hierarchy: call pfx_amg_setup or solve with PFX_PRECOND_AMG first");
		info->n_levels = (int)A->lv.size();
		for (size_t l = 0; l < A->lv.size() && l < 16; ++l)
		{
			info->n_vertices[l] = l == 0 ? ctx->n_owned : A->lv[l].n_overt;
			info->n_aggregates[l] = A->lv[l].n_agg;
			info->n_blocks[l] = A->lv[l].m->n_blocks;
			info->lambda_max[l] = A->lv[l].lmax;
		}
		info->operator_complexity = (double)A->work_blocks / (double)ctx->n_blocks;
		info->symbolic_s = A->symbolic_s;
		info->numeric_s = A->numeric_s;
		return PFX_OK;
	}

	pfx_status pfx_amg_get_aggregates(pfx_ctx *ctx, int level, int32_t *agg)
	{
		if (!ctx || !agg)
			return PFX_ERR_ARG;
		pfx_amg *A = ctx->amg;
		if (!A || level < 0 || level + 1 >= (int)A->lv.size())
			PFX_FAIL(PFX_ERR_ARG, "pfx_amg_get_aggregates: no such level");
		PFX_HIP(hipSetDevice(ctx->device));
		const int64_t n = level == 0 ? ctx->n_nodes : A->lv[(size_t)level].n_overt;
		PFX_HIP(hipMemcpyAsync(agg, A->lv[(size_t)level].d_agg, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
		PFX_HIP(hipStreamSynchronize(ctx->stream));
		return PFX_OK;
	}

	pfx_status pfx_amg_apply_dev(pfx_ctx *ctx, const double *d_r, double *d_z)
	{
		if (!ctx || !d_r || !d_z)
			return PFX_ERR_ARG;
		pfx_status rc = pfx_amg_setup(ctx);
		if (rc)
			return rc;
		const int64_t nfull = ctx->n_dofs();
		double *base = amg_workspace(ctx);
		// the preconditioner is that of the owned-owned block: the entries of halo nodes of z take part in the level-0
		// products as zeros
		if (ctx->n_owned < ctx->n_nodes)
			PFX_HIP(hipMemsetAsync(d_z + ctx->n_owned * ctx->bs, 0, sizeof(double) * (size_t)((ctx->n_nodes - ctx->n_owned) * ctx->bs), ctx->stream));
		return pfx_amg_apply(ctx, d_r, d_z, base + 2 * nfull, base + 3 * nfull);
	}
}
