// pfx_psd.h -- projection of the element Hessians onto the cone of positive semi-definite matrices.
//
// Reference: NLAssembler::assemble_hessian with project_to_psd = true (src/polyfem/assembler/Assembler.cpp:554-555)
// calls ipc::project_to_psd on every local (n_loc*3)^2 matrix: symmetric eigendecomposition, the matrix is returned
// unchanged if its smallest eigenvalue is >= 0, otherwise V max(D, 0) V^T.  The Newton solver switches this on when a
// plain Newton step fails (SparseNewtonDescentSolver.tpp:105-123, descent strategy 1).  ipc-toolkit and Eigen's
// SelfAdjointEigenSolver are not part of the reference tree; the eigenvalues are computed here with cyclic Jacobi
// rotations (round-robin ordering: n/2 independent rotations per step), which yields the same projection up to
// round-off.  The tournament schedule is tabulated in LDS once per element and the rotation passes give every lane a
// fixed matrix row, so the inner loops contain no integer division.  Elements whose Hessian is positive definite after
// a shift of 1e-12 of its largest diagonal entry skip the eigendecomposition (elimination test, see the kernel).
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
	static_assert(N % 2 == 0 && N <= 64, "round-robin ordering needs an even size; one lane per row");
	// lanes are laid out as (pair slot, row): RW rows per pair, PPI pairs per pass over the wave
	constexpr int RW = N <= 16 ? 16 : (N <= 32 ? 32 : 64), PPI = 64 / RW;
	constexpr int LD = N + 1; // odd row stride: the column passes (one row per lane) stay free of bank conflicts
	__shared__ double A[N * LD], V[N * LD], cs[N];
	__shared__ uint8_t pq[(N - 1) * H * 2]; // tournament schedule: pair k of step t
	__shared__ uint8_t ij[NPAIR * 2];       // (i, j) of the local block q = i(i+1)/2 + j
	__shared__ double optab[Op::LDS_DOUBLES];
	const int lane = threadIdx.x;
	const int64_t e = blockIdx.x;
	if (e >= n_el)
		return;
	op.load_tables(optab, lane);
	for (int idx = lane; idx < (N - 1) * H; idx += 64)
	{
		int p, q;
		rr_pair(N, idx / H, idx % H, p, q);
		pq[2 * idx] = (uint8_t)p, pq[2 * idx + 1] = (uint8_t)q;
	}
	for (int q = lane; q < NPAIR; q += 64)
	{
		int i = 0;
		while ((i + 1) * (i + 2) / 2 <= q)
			++i;
		ij[2 * q] = (uint8_t)i, ij[2 * q + 1] = (uint8_t)(q - i * (i + 1) / 2);
	}
	__syncthreads();
	// local Hessian, both triangles
	for (int q = lane; q < NPAIR; q += 64)
	{
		const int i = ij[2 * q], j = ij[2 * q + 1];
		double b[9];
		op(optab, (uint32_t)e, i, j, b);
		for (int m = 0; m < 3; ++m)
			for (int n = 0; n < 3; ++n)
			{
				A[(3 * i + m) * LD + 3 * j + n] = b[3 * m + n];
				A[(3 * j + n) * LD + 3 * i + m] = b[3 * m + n];
			}
	}
	const int row = lane & (RW - 1), slot = lane / RW; // this lane's matrix row and column / pair slot in the passes below
	double *out = kloc + e * (int64_t)(NPAIR * 9);

	// Fast path.  If A + shift I (shift = 1e-12 max diag) can be eliminated without pivoting with positive pivots, it is
	// positive definite, so lambda_min(A) > -shift and the projection leaves A as it is up to that shift -- which is the
	// size of the round-off that decides the sign of the zero eigenvalues (rigid translations) in the exact test anyway.
	// Costs N eliminations steps against some thirteen Jacobi sweeps of N - 1 steps; V is the work copy.
	{
		double dmax = lane < N ? A[lane * LD + lane] : 0.0;
		for (int o = 32; o > 0; o >>= 1)
		{
			const double other = __shfl_xor(dmax, o, 64);
			dmax = other > dmax ? other : dmax;
		}
		for (int t = lane; t < N * LD; t += 64)
			V[t] = A[t];
		__syncthreads();
		if (lane < N)
			V[lane * LD + lane] += 1e-12 * dmax;
		__syncthreads();
		bool definite = dmax > 0.0;
		for (int k = 0; k < N - 1 && definite; ++k)
		{
			const double d = V[k * LD + k];
			if (!(d > 0.0))
			{
				definite = false; // the same value in every lane
				break;
			}
			if (row > k && row < N)
			{
				const double l = V[row * LD + k] / d;
				for (int c = k + 1 + slot; c < N; c += PPI)
					V[row * LD + c] -= l * V[k * LD + c];
			}
			__syncthreads();
		}
		if (definite && V[(N - 1) * LD + N - 1] > 0.0)
		{
			for (int idx = lane; idx < NPAIR * 9; idx += 64)
			{
				const int q = idx / 9, mn = idx - q * 9;
				out[idx] = A[(3 * ij[2 * q] + mn / 3) * LD + 3 * ij[2 * q + 1] + mn % 3];
			}
			return;
		}
		__syncthreads();
	}

	for (int t = lane; t < N * LD; t += 64)
		V[t] = 0.0;
	__syncthreads();
	if (lane < N)
		V[lane * LD + lane] = 1.0;
	double norm2 = 0;
	if (lane < N)
		for (int c = 0; c < N; ++c)
			norm2 += A[lane * LD + c] * A[lane * LD + c];
	norm2 = wave_sum_all(norm2);
	__syncthreads();

	double prev_off = 1e300;
	for (int sweep = 0; sweep < 60; ++sweep)
	{
		// off-diagonal mass: lane r sums row r right of the diagonal
		double off = 0;
		if (lane < N)
			for (int c = lane + 1; c < N; ++c)
				off += A[lane * LD + c] * A[lane * LD + c];
		off = wave_sum_all(off);
		// converged, or stagnating at the round-off floor (N^2/2 entries of size eps |A| keep the sum near 1e-30 |A|^2:
		// without the second test every element would run all 60 sweeps)
		if (off <= 1e-32 * norm2 || (off <= 1e-24 * norm2 && off > 0.25 * prev_off))
			break;
		prev_off = off;
		for (int t = 0; t < N - 1; ++t)
		{
			const uint8_t *sched = pq + 2 * t * H;
			if (lane < H)
			{
				const int p = sched[2 * lane], q = sched[2 * lane + 1];
				const double apq = A[p * LD + q];
				double c = 1.0, s = 0.0;
				if (apq != 0.0)
				{
					const double theta = (A[q * LD + q] - A[p * LD + p]) / (2.0 * apq);
					const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
					c = 1.0 / sqrt(tt * tt + 1.0);
					s = tt * c;
				}
				cs[2 * lane] = c, cs[2 * lane + 1] = s;
			}
			__syncthreads();
			// columns p, q of A and V
			if (row < N)
				for (int k = slot; k < H; k += PPI)
				{
					const int p = sched[2 * k], q = sched[2 * k + 1];
					const double c = cs[2 * k], s = cs[2 * k + 1];
					const double ap = A[row * LD + p], aq = A[row * LD + q];
					A[row * LD + p] = c * ap - s * aq;
					A[row * LD + q] = s * ap + c * aq;
					const double vp = V[row * LD + p], vq = V[row * LD + q];
					V[row * LD + p] = c * vp - s * vq;
					V[row * LD + q] = s * vp + c * vq;
				}
			__syncthreads();
			// rows p, q of A
			if (row < N)
				for (int k = slot; k < H; k += PPI)
				{
					const int p = sched[2 * k], q = sched[2 * k + 1];
					const double c = cs[2 * k], s = cs[2 * k + 1];
					const double ap = A[p * LD + row], aq = A[q * LD + row];
					A[p * LD + row] = c * ap - s * aq;
					A[q * LD + row] = s * ap + c * aq;
				}
			__syncthreads();
		}
	}
	double lmin = lane < N ? A[lane * LD + lane] : 1e300;
	lmin = wave_min_all(lmin);
	if (lmin >= 0.0)
	{
		// already positive semi-definite: the matrix is kept as it is
		for (int q = lane; q < NPAIR; q += 64)
		{
			double b[9];
			op(optab, (uint32_t)e, ij[2 * q], ij[2 * q + 1], b);
			for (int k = 0; k < 9; ++k)
				out[q * 9 + k] = b[k];
		}
		return;
	}
	// V max(D, 0) V^T: scale the columns of the negative eigenvalues away once, then one dot product per entry
	if (lane < N)
		cs[lane] = A[lane * LD + lane] > 0.0 ? A[lane * LD + lane] : 0.0;
	__syncthreads();
	for (int q = lane; q < NPAIR; q += 64)
	{
		const int i = ij[2 * q], j = ij[2 * q + 1];
		double b[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
		for (int k = 0; k < N; ++k)
		{
			const double l = cs[k];
			const double vi0 = l * V[(3 * i) * LD + k], vi1 = l * V[(3 * i + 1) * LD + k], vi2 = l * V[(3 * i + 2) * LD + k];
			const double vj0 = V[(3 * j) * LD + k], vj1 = V[(3 * j + 1) * LD + k], vj2 = V[(3 * j + 2) * LD + k];
			b[0] += vi0 * vj0, b[1] += vi0 * vj1, b[2] += vi0 * vj2;
			b[3] += vi1 * vj0, b[4] += vi1 * vj1, b[5] += vi1 * vj2;
			b[6] += vi2 * vj0, b[7] += vi2 * vj1, b[8] += vi2 * vj2;
		}
		for (int k = 0; k < 9; ++k)
			out[q * 9 + k] = b[k];
	}
}
