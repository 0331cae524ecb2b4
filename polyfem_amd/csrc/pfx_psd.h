// pfx_psd.h -- projection of the element Hessians onto the cone of positive semi-definite matrices.
//
// Reference: NLAssembler::assemble_hessian with project_to_psd = true (src/polyfem/assembler/Assembler.cpp:554-555)
// calls ipc::project_to_psd on every local (n_loc*dim)^2 matrix: symmetric eigendecomposition, the matrix is returned
// unchanged if its smallest eigenvalue is >= 0, otherwise V max(D, 0) V^T.  The Newton solver switches this on when a
// plain Newton step fails (SparseNewtonDescentSolver.tpp:105-123, descent strategy 1).  ipc-toolkit and Eigen's
// SelfAdjointEigenSolver are not part of the reference tree.  Like that solver, the kernel reduces the matrix to
// tridiagonal form with Householder reflections and diagonalises it with implicitly shifted QL sweeps (the classical
// tred2 / tql2 pair), which yields the same projection up to round-off:
//   * Householder phase: one lane per matrix row, the dot products of a step are wave reductions;
//   * QL phase: the scalar recurrence of the rotations is evaluated redundantly by every lane (identical stores to d
//     and e), and lane k applies each rotation to row k of the eigenvector matrix -- no lane reads anything it has not
//     also written itself, so the phase needs no barrier and the elements resident on a CU hide each other's latencies.
// Elements whose Hessian is positive definite after a shift of 1e-12 of its largest diagonal entry skip the
// eigendecomposition (elimination test, see the kernel).
//
// This is the only assembly path that materialises local matrices: one 64-lane workgroup per element builds the local
// Hessian in LDS from the contribution functor, factorises it there and writes the (projected) blocks of the pairs
// j <= i to HBM (n_loc(n_loc+1)/2 * dim^2 doubles per element); the row assembly then reads them through staged_block.
#pragma once
#include "pfx_internal.h"

template <int NLOC, int BS = 3>
struct staged_block
{
	const double *kloc;
	static constexpr int LDS_DOUBLES = 1;
	__device__ __forceinline__ void load_tables(double *, int) const {}
	__device__ __forceinline__ void operator()(const double *, uint32_t e, int i, int j, double (&out)[BS * BS]) const
	{
		constexpr int64_t NPAIR = NLOC * (NLOC + 1) / 2;
		const double *p = kloc + ((int64_t)e * NPAIR + (i * (i + 1) / 2 + j)) * (BS * BS);
#pragma unroll
		for (int k = 0; k < BS * BS; ++k)
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
	__device__ __forceinline__ double wave_max_all(double v)
	{
		for (int off = 32; off > 0; off >>= 1)
		{
			const double o = __shfl_xor(v, off, 64);
			v = o > v ? o : v;
		}
		return v;
	}

	// Householder reduction of the symmetric N x N matrix in Z (row stride LD, lower triangle referenced) to tridiagonal
	// form; on return Z holds the orthogonal transformation, d the diagonal, e[1..N-1] the sub-diagonal.  Called by all
	// 64 lanes of a single-wave workgroup; lane t owns the rows / columns t, t + 64, ... (R of them).
	template <int N, int LD, int R>
	__device__ void householder_tridiagonal(double *Z, double *d, double *e, int lane)
	{
		for (int i = N - 1; i >= 1; --i)
		{
			const int l = i - 1;
			double h = 0.0;
			if (l > 0)
			{
				double part = 0.0;
#pragma unroll
				for (int rr = 0; rr < R; ++rr)
					if (lane + 64 * rr <= l)
						part += fabs(Z[i * LD + lane + 64 * rr]);
				const double scale = wave_sum_all(part);
				if (scale == 0.0)
				{
					if (lane == 0)
						e[i] = Z[i * LD + l];
				}
				else
				{
					part = 0.0;
#pragma unroll
					for (int rr = 0; rr < R; ++rr)
						if (lane + 64 * rr <= l)
						{
							const double v = Z[i * LD + lane + 64 * rr] / scale;
							Z[i * LD + lane + 64 * rr] = v;
							part += v * v;
						}
					h = wave_sum_all(part);
					__syncthreads();
					const double f0 = Z[i * LD + l];
					const double g0 = f0 >= 0.0 ? -sqrt(h) : sqrt(h);
					h -= f0 * g0;
					__syncthreads();
					if (lane == 0)
					{
						e[i] = scale * g0;
						Z[i * LD + l] = f0 - g0;
					}
					__syncthreads();
					// e[j] = (A u)_j / h for the rows j <= l, u = row i
					double fpart = 0.0;
#pragma unroll
					for (int rr = 0; rr < R; ++rr)
					{
						const int j = lane + 64 * rr;
						if (j <= l)
						{
							const double uj = Z[i * LD + j];
							Z[j * LD + i] = uj / h;
							double g = 0.0;
							for (int k = 0; k <= j; ++k)
								g += Z[j * LD + k] * Z[i * LD + k];
							for (int k = j + 1; k <= l; ++k)
								g += Z[k * LD + j] * Z[i * LD + k];
							const double ej = g / h;
							e[j] = ej;
							fpart += ej * uj;
						}
					}
					const double hh = wave_sum_all(fpart) / (h + h);
					__syncthreads();
#pragma unroll
					for (int rr = 0; rr < R; ++rr)
						if (lane + 64 * rr <= l)
							e[lane + 64 * rr] -= hh * Z[i * LD + lane + 64 * rr];
					__syncthreads();
#pragma unroll
					for (int rr = 0; rr < R; ++rr)
					{
						const int j = lane + 64 * rr;
						if (j <= l)
						{
							const double f = Z[i * LD + j], g = e[j];
							for (int k = 0; k <= j; ++k)
								Z[j * LD + k] -= f * e[k] + g * Z[i * LD + k];
						}
					}
				}
			}
			else if (lane == 0)
				e[i] = Z[i * LD + l];
			if (lane == 0)
				d[i] = h;
			__syncthreads();
		}
		if (lane == 0)
			d[0] = 0.0, e[0] = 0.0;
		__syncthreads();
		// accumulate the reflections
		for (int i = 0; i < N; ++i)
		{
			const int l = i - 1;
			if (d[i] != 0.0)
			{
#pragma unroll
				for (int rr = 0; rr < R; ++rr)
				{
					const int j = lane + 64 * rr;
					if (j <= l)
					{
						double g = 0.0;
						for (int k = 0; k <= l; ++k)
							g += Z[i * LD + k] * Z[k * LD + j];
						for (int k = 0; k <= l; ++k)
							Z[k * LD + j] -= g * Z[k * LD + i];
					}
				}
			}
			__syncthreads();
			if (lane == 0)
			{
				d[i] = Z[i * LD + i];
				Z[i * LD + i] = 1.0;
			}
#pragma unroll
			for (int rr = 0; rr < R; ++rr)
				if (lane + 64 * rr <= l)
					Z[(lane + 64 * rr) * LD + i] = 0.0, Z[i * LD + lane + 64 * rr] = 0.0;
			__syncthreads();
		}
	}

	// Implicit QL with Wilkinson shifts on the tridiagonal (d, e); the rotations are applied to the rows of Z, lane t
	// owning the rows t, t + 64, ...  Every lane runs the same scalar recurrence on d and e and stores the same values (a
	// lane only ever reads what it has also written itself), so the phase needs no barrier.  Returns false if an
	// eigenvalue needs more than 60 sweeps.
	template <int N, int LD, int R>
	__device__ bool ql_implicit(double *Z, double *dl, double *el, int lane)
	{
		for (int i = 1; i < N; ++i)
			el[i - 1] = el[i];
		el[N - 1] = 0.0;
		bool has_row[R];
		double *zr[R];
#pragma unroll
		for (int rr = 0; rr < R; ++rr)
		{
			has_row[rr] = lane + 64 * rr < N;
			zr[rr] = Z + (has_row[rr] ? lane + 64 * rr : 0) * LD;
		}
		for (int l = 0; l < N; ++l)
		{
			int iter = 0, m;
			do
			{
				// first m >= l whose sub-diagonal entry is negligible (N - 1 if there is none): one candidate per lane and pass
				m = N - 1;
#pragma unroll
				for (int rr = R - 1; rr >= 0; --rr)
				{
					const int t = lane + 64 * rr;
					const double dd = t < N - 1 ? fabs(dl[t]) + fabs(dl[t + 1]) : 0.0;
					const bool negligible = t < N - 1 && fabs(el[t]) + dd == dd;
					const unsigned long long hit = __ballot(negligible && t >= l);
					if (hit)
						m = 64 * rr + __ffsll((long long)hit) - 1;
				}
				if (m != l)
				{
					if (iter++ == 60)
						return false;
					double g = (dl[l + 1] - dl[l]) / (2.0 * el[l]);
					double r = sqrt(g * g + 1.0);
					g = dl[m] - dl[l] + el[l] / (g + (g >= 0.0 ? r : -r));
					double s = 1.0, c = 1.0, p = 0.0;
					int i;
					// The rotations form one dependent chain; everything that does not depend on the previous rotation
					// (e[i-1], d[i-1] and this lane's z[i-1] of the next step) is loaded one step ahead, and d[i], z[i]
					// are carried over in registers, so that only arithmetic is left on the chain.
					double e_i = el[m - 1], d_i = dl[m - 1], d_ip1 = dl[m];
					double z_i[R], z_ip1[R];
#pragma unroll
					for (int rr = 0; rr < R; ++rr)
						z_i[rr] = has_row[rr] ? zr[rr][m - 1] : 0.0, z_ip1[rr] = has_row[rr] ? zr[rr][m] : 0.0;
					for (i = m - 1; i >= l; --i)
					{
						const double e_nx = i > l ? el[i - 1] : 0.0, d_nx = i > l ? dl[i - 1] : 0.0;
						double z_nx[R];
#pragma unroll
						for (int rr = 0; rr < R; ++rr)
							z_nx[rr] = (has_row[rr] && i > l) ? zr[rr][i - 1] : 0.0;
						const double f = s * e_i;
						const double b = c * e_i;
						// |(f, g)| and its reciprocal from one reciprocal square root (entries of an element Hessian: no
						// overflow, and an underflow to zero deflates)
						const double t2 = f * f + g * g;
						const double inv = t2 > 0.0 ? rsqrt(t2) : 0.0;
						r = t2 * inv;
						el[i + 1] = r;
						if (r == 0.0)
						{
							dl[i + 1] = d_ip1 - p;
							el[m] = 0.0;
#pragma unroll
							for (int rr = 0; rr < R; ++rr)
								if (has_row[rr])
									zr[rr][i + 1] = z_ip1[rr];
							break;
						}
						s = f * inv;
						c = g * inv;
						g = d_ip1 - p;
						r = (d_i - g) * s + 2.0 * c * b;
						p = s * r;
						dl[i + 1] = g + p;
						g = c * r - b;
#pragma unroll
						for (int rr = 0; rr < R; ++rr)
						{
							if (has_row[rr])
								zr[rr][i + 1] = s * z_i[rr] + c * z_ip1[rr];
							z_ip1[rr] = c * z_i[rr] - s * z_ip1[rr]; // becomes z[i], the upper member of the next pair
							z_i[rr] = z_nx[rr];
						}
						d_ip1 = d_i, d_i = d_nx, e_i = e_nx;
					}
					if (i < l)
					{
#pragma unroll
						for (int rr = 0; rr < R; ++rr)
							if (has_row[rr])
								zr[rr][l] = z_ip1[rr];
					}
					if (r == 0.0 && i >= l)
						continue;
					dl[l] -= p;
					el[l] = g;
					el[m] = 0.0;
				}
			} while (m != l);
		}
		return true;
	}
} // namespace pfx_psd_detail

// W: N x LD doubles of dynamic shared memory (N = BS NLOC, LD = N rounded up to odd); BS = 3, or 2 for the triangles
template <int NLOC, int BS, class Op>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NLOC <= 10 ? 4 : 1, NLOC <= 10 ? 4 : 8))) void k_project_psd(Op op, int64_t n_el, double *__restrict__ kloc)
{
	using namespace pfx_psd_detail;
	constexpr int N = NLOC * BS, NPAIR = NLOC * (NLOC + 1) / 2, B2 = BS * BS;
	constexpr int R = (N + 63) / 64;                                   // matrix rows per lane
	constexpr int RW = N <= 16 ? 16 : (N <= 32 ? 32 : 64), PPI = 64 / RW; // elimination pass: lanes laid out as (column slot, row)
	constexpr int LD = N | 1;                                            // odd row stride: one row per lane without bank conflicts
	extern __shared__ double pfx_psd_dyn[];
	double *W = pfx_psd_dyn;
	__shared__ double d[N], e[N];
	__shared__ uint8_t ij[NPAIR * 2]; // (i, j) of the local block q = i(i+1)/2 + j
	__shared__ double optab[Op::LDS_DOUBLES];
	const int lane = threadIdx.x;
	const int64_t el = blockIdx.x;
	if (el >= n_el)
		return;
	op.load_tables(optab, lane);
	for (int q = lane; q < NPAIR; q += 64)
	{
		int i = 0;
		while ((i + 1) * (i + 2) / 2 <= q)
			++i;
		ij[2 * q] = (uint8_t)i, ij[2 * q + 1] = (uint8_t)(q - i * (i + 1) / 2);
	}
	__syncthreads();
	// local Hessian into W, both triangles (evaluated again whenever a fresh copy is needed: cheaper than a second
	// matrix in LDS, which would halve the number of elements a CU works on at a time)
	auto fill = [&]() {
		for (int q = lane; q < NPAIR; q += 64)
		{
			const int i = ij[2 * q], j = ij[2 * q + 1];
			double b[B2];
			op(optab, (uint32_t)el, i, j, b);
			for (int m = 0; m < BS; ++m)
				for (int n = 0; n < BS; ++n)
				{
					W[(BS * i + m) * LD + BS * j + n] = b[BS * m + n];
					W[(BS * j + n) * LD + BS * i + m] = b[BS * m + n];
				}
		}
		__syncthreads();
	};
	fill();
	const int row = lane & (RW - 1), slot = lane / RW;
	double *out = kloc + el * (int64_t)(NPAIR * B2);
	double dloc = 0.0;
#pragma unroll
	for (int rr = 0; rr < R; ++rr)
		if (lane + 64 * rr < N)
			dloc = fmax(dloc, W[(lane + 64 * rr) * LD + lane + 64 * rr]);
	const double dmax = wave_max_all(dloc);

	// Fast path.  If A + shift I (shift = 1e-12 max diag) can be eliminated without pivoting with positive pivots, it is
	// positive definite, so lambda_min(A) > -shift and the projection leaves A as it is up to that shift -- which is the
	// size of the round-off that decides the sign of the zero eigenvalues (rigid translations) in the exact test anyway.
	__syncthreads();
#pragma unroll
	for (int rr = 0; rr < R; ++rr)
		if (lane + 64 * rr < N)
			W[(lane + 64 * rr) * LD + lane + 64 * rr] += 1e-12 * dmax;
	__syncthreads();
	bool keep = dmax > 0.0;
	for (int k = 0; k < N - 1 && keep; ++k)
	{
		const double piv = W[k * LD + k];
		if (!(piv > 0.0))
		{
			keep = false; // the same value in every lane
			break;
		}
#pragma unroll
		for (int rr = 0; rr < R; ++rr)
		{
			const int r = row + 64 * rr;
			if (r > k && r < N)
			{
				const double l = W[r * LD + k] / piv;
				for (int c = k + 1 + slot; c < N; c += PPI)
					W[r * LD + c] -= l * W[k * LD + c];
			}
		}
		__syncthreads();
	}
	keep = keep && W[(N - 1) * LD + N - 1] > 0.0;
	__syncthreads();

	if (!keep)
	{
		fill();
		householder_tridiagonal<N, LD, R>(W, d, e, lane);
		const bool ok = ql_implicit<N, LD, R>(W, d, e, lane);
		__syncthreads();
		double lmin = 1e300;
#pragma unroll
		for (int rr = 0; rr < R; ++rr)
			if (lane + 64 * rr < N)
				lmin = fmin(lmin, d[lane + 64 * rr]);
		lmin = -wave_max_all(-lmin);
		keep = ok && lmin >= 0.0;
		if (!keep)
		{
			// V max(D, 0) V^T (not-a-number where the iteration did not converge, which finite input does not produce)
			__syncthreads();
#pragma unroll
			for (int rr = 0; rr < R; ++rr)
				if (lane + 64 * rr < N)
					e[lane + 64 * rr] = !ok ? 0.0 / 0.0 : (d[lane + 64 * rr] > 0.0 ? d[lane + 64 * rr] : 0.0);
			__syncthreads();
			for (int q = lane; q < NPAIR; q += 64)
			{
				const int i = ij[2 * q], j = ij[2 * q + 1];
				double b[B2];
#pragma unroll
				for (int k = 0; k < B2; ++k)
					b[k] = 0.0;
				for (int k = 0; k < N; ++k)
				{
					const double l = e[k];
					double vi[BS], vj[BS];
#pragma unroll
					for (int m = 0; m < BS; ++m)
						vi[m] = l * W[(BS * i + m) * LD + k], vj[m] = W[(BS * j + m) * LD + k];
#pragma unroll
					for (int m = 0; m < BS; ++m)
#pragma unroll
						for (int n = 0; n < BS; ++n)
							b[BS * m + n] += vi[m] * vj[n];
				}
				for (int k = 0; k < B2; ++k)
					out[q * B2 + k] = b[k];
			}
			return;
		}
	}
	// the matrix is kept as it is
	for (int q = lane; q < NPAIR; q += 64)
	{
		double b[B2];
		op(optab, (uint32_t)el, ij[2 * q], ij[2 * q + 1], b);
		for (int k = 0; k < B2; ++k)
			out[q * B2 + k] = b[k];
	}
}

// launches k_project_psd for all elements; the matrix lives in dynamic shared memory (more than the 64 KB a kernel may
// claim statically for 27-node elements)
template <int NLOC, int BS, class Op>
pfx_status pfx_launch_project_psd(pfx_ctx *ctx, Op op)
{
	constexpr int N = NLOC * BS, LD = N | 1;
	const size_t bytes = sizeof(double) * N * LD;
	if (bytes > 32 * 1024)
		PFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_project_psd<NLOC, BS, Op>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
	hipLaunchKernelGGL((k_project_psd<NLOC, BS, Op>), dim3((unsigned)ctx->n_el), dim3(64), bytes, ctx->stream, op, ctx->n_el, ctx->d_kloc);
	PFX_HIP(hipGetLastError());
	return PFX_OK;
}
