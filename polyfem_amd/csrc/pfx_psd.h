// pfx_psd.h -- projection of the element Hessians onto the cone of positive semi-definite matrices.
//
// Reference: NLAssembler::assemble_hessian with project_to_psd = true (src/polyfem/assembler/Assembler.cpp:554-555)
// calls ipc::project_to_psd on every local (n_loc*3)^2 matrix: symmetric eigendecomposition, the matrix is returned
// unchanged if its smallest eigenvalue is >= 0, otherwise V max(D, 0) V^T.  The Newton solver switches this on when a
// plain Newton step fails (SparseNewtonDescentSolver.tpp:105-123, descent strategy 1).  ipc-toolkit and Eigen's
// SelfAdjointEigenSolver are not part of the reference tree; the eigenvalues are computed here with cyclic Jacobi
// rotations (round-robin ordering: n/2 independent rotations per step), which yields the same projection up to
// round-off.
//
// This is the only assembly path that materialises local matrices: one 64-lane workgroup per element builds the local
// Hessian in LDS from the contribution functor, diagonalises it there and writes the projected blocks of the pairs
// j <= i to HBM (n_loc(n_loc+1)/2 * 9 doubles per element); the row assembly then reads them through staged_block.
#pragma once
#include "pfx_internal.h"

template <int NLOC>
struct staged_block
{
	const double *kloc;
	static constexpr int LDS_DOUBLES = 1;
	__device__ __forceinline__ void load_tables(double *, int) const {}
	__device__ __forceinline__ void operator()(const double *, uint32_t e, int i, int j, double (&out)[9]) const
	{
		constexpr int64_t NPAIR = NLOC * (NLOC + 1) / 2;
		const double *p = kloc + ((int64_t)e * NPAIR + (i * (i + 1) / 2 + j)) * 9;
#pragma unroll
		for (int k = 0; k < 9; ++k)
			out[k] = p[k];
	}
};

namespace pfx_psd_detail
{
	__device__ __forceinline__ double wave_sum_all(double v)
	{
		for (int off = 32; off > 0; off >>= 1)
			v += __shfl_xor(v, off, 64);
		return v;
	}
	__device__ __forceinline__ double wave_min_all(double v)
	{
		for (int off = 32; off > 0; off >>= 1)
		{
			const double o = __shfl_xor(v, off, 64);
			v = o < v ? o : v;
		}
		return v;
	}
	// pair k of step t of the round-robin tournament on n players (n even): p < q
	__device__ __forceinline__ void rr_pair(int n, int t, int k, int &p, int &q)
	{
		int a, b;
		if (k == 0)
			a = n - 1, b = t;
		else
			a = (t + k) % (n - 1), b = (t - k + (n - 1)) % (n - 1);
		p = a < b ? a : b;
		q = a < b ? b : a;
	}
} // namespace pfx_psd_detail

template <int NLOC, class Op>
__global__ __launch_bounds__(64) void k_project_psd(Op op, int64_t n_el, double *__restrict__ kloc)
{
	using namespace pfx_psd_detail;
	constexpr int N = NLOC * 3, NPAIR = NLOC * (NLOC + 1) / 2, H = N / 2;
	static_assert(N % 2 == 0, "round-robin ordering needs an even size");
	__shared__ double A[N * N], V[N * N], cs[2 * H];
	__shared__ double optab[Op::LDS_DOUBLES];
	const int lane = threadIdx.x;
	const int64_t e = blockIdx.x;
	if (e >= n_el)
		return;
	op.load_tables(optab, lane);
	__syncthreads();
	// local Hessian, both triangles
	for (int q = lane; q < NPAIR; q += 64)
	{
		int i = 0;
		while ((i + 1) * (i + 2) / 2 <= q)
			++i;
		const int j = q - i * (i + 1) / 2;
		double b[9];
		op(optab, (uint32_t)e, i, j, b);
		for (int m = 0; m < 3; ++m)
			for (int n = 0; n < 3; ++n)
			{
				A[(3 * i + m) * N + 3 * j + n] = b[3 * m + n];
				A[(3 * j + n) * N + 3 * i + m] = b[3 * m + n];
			}
	}
	for (int t = lane; t < N * N; t += 64)
		V[t] = (t / N == t % N) ? 1.0 : 0.0;
	__syncthreads();
	double norm2 = 0;
	for (int t = lane; t < N * N; t += 64)
		norm2 += A[t] * A[t];
	norm2 = wave_sum_all(norm2);

	for (int sweep = 0; sweep < 60; ++sweep)
	{
		double off = 0;
		for (int t = lane; t < N * N; t += 64)
		{
			const int r = t / N, c = t - r * N;
			if (c > r)
				off += A[t] * A[t];
		}
		off = wave_sum_all(off);
		if (off <= 1e-32 * norm2)
			break;
		for (int t = 0; t < N - 1; ++t)
		{
			if (lane < H)
			{
				int p, q;
				rr_pair(N, t, lane, p, q);
				const double apq = A[p * N + q];
				double c = 1.0, s = 0.0;
				if (apq != 0.0)
				{
					const double theta = (A[q * N + q] - A[p * N + p]) / (2.0 * apq);
					const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
					c = 1.0 / sqrt(tt * tt + 1.0);
					s = tt * c;
				}
				cs[2 * lane] = c, cs[2 * lane + 1] = s;
			}
			__syncthreads();
			// columns p, q of A and V
			for (int idx = lane; idx < H * N; idx += 64)
			{
				const int k = idx / N, r = idx - k * N;
				int p, q;
				rr_pair(N, t, k, p, q);
				const double c = cs[2 * k], s = cs[2 * k + 1];
				const double ap = A[r * N + p], aq = A[r * N + q];
				A[r * N + p] = c * ap - s * aq;
				A[r * N + q] = s * ap + c * aq;
				const double vp = V[r * N + p], vq = V[r * N + q];
				V[r * N + p] = c * vp - s * vq;
				V[r * N + q] = s * vp + c * vq;
			}
			__syncthreads();
			// rows p, q of A
			for (int idx = lane; idx < H * N; idx += 64)
			{
				const int k = idx / N, col = idx - k * N;
				int p, q;
				rr_pair(N, t, k, p, q);
				const double c = cs[2 * k], s = cs[2 * k + 1];
				const double ap = A[p * N + col], aq = A[q * N + col];
				A[p * N + col] = c * ap - s * aq;
				A[q * N + col] = s * ap + c * aq;
			}
			__syncthreads();
		}
	}
	double lmin = 1e300;
	for (int k = lane; k < N; k += 64)
		lmin = A[k * N + k] < lmin ? A[k * N + k] : lmin;
	lmin = wave_min_all(lmin);
	double *out = kloc + e * (int64_t)(NPAIR * 9);
	if (lmin >= 0.0)
	{
		// already positive semi-definite: the matrix is kept as it is
		for (int q = lane; q < NPAIR; q += 64)
		{
			int i = 0;
			while ((i + 1) * (i + 2) / 2 <= q)
				++i;
			const int j = q - i * (i + 1) / 2;
			double b[9];
			op(optab, (uint32_t)e, i, j, b);
			for (int k = 0; k < 9; ++k)
				out[q * 9 + k] = b[k];
		}
		return;
	}
	for (int idx = lane; idx < NPAIR * 9; idx += 64)
	{
		const int q = idx / 9, mn = idx - q * 9;
		int i = 0;
		while ((i + 1) * (i + 2) / 2 <= q)
			++i;
		const int j = q - i * (i + 1) / 2;
		const int r = 3 * i + mn / 3, c = 3 * j + mn % 3;
		double s = 0;
		for (int k = 0; k < N; ++k)
		{
			const double l = A[k * N + k];
			if (l > 0.0)
				s += l * V[r * N + k] * V[c * N + k];
		}
		out[idx] = s;
	}
}
