// pfx_local_linear.hip -- local stiffness blocks of the LINEAR assemblers (Laplacian, LinearElasticity).
//
// Reference: Assembler<L>::assemble element loop (src/polyfem/assembler/Assembler.cpp:127-209),
// ElementAssemblyValues::compute/finalize3d (ElementAssemblyValues.cpp:94-133, 205-351),
// Laplacian::assemble (Laplacian.cpp:7-18), LinearElasticity::assemble (LinearElasticity.cpp:21-54).
//
// THIS FILE IS COMPILED WITH -ffp-contract=off.  The reference drops local values with |v| < 1e-30
// (Assembler.cpp:170-173), so the sparsity pattern of the assembled matrix depends on which values are exact
// zeros; every expression below is therefore evaluated in the reference's operation order, one IEEE operation
// per source operation, and the 3x3 determinant / inverse follow Eigen 3.4's fixed-size routines.
//
// Work decomposition: TPE lanes cooperate on one element.
//   phase 1  lanes < n_pts            : Jacobian of the geometric map at a quadrature point -> det, J^{-T}
//   phase 2  lanes over (point,basis) : physical gradients grad * J^{-T}  -> LDS
//   phase 3  lanes over pairs j <= i  : local block by the quadrature loop -> LDS tile (mirrored for j < i)
//   phase 4  all lanes                : coalesced copy of the (n_loc*bs)^2 tile to the element's slot in HBM
#include "pfx_internal.h"

namespace
{
	template <int NLOC, int NGLOC, int NPTS, int BS>
	struct lin_smem
	{
		double xs[NGLOC * 3];
		double jt[NPTS * 9];
		double da[NPTS];
		double gt[NPTS * NLOC * 3];
	};

	__device__ __forceinline__ double det3_fixed(const double *m)
	{
		const double h0 = m[0] * (m[4] * m[8] - m[5] * m[7]);
		const double h1 = m[1] * (m[3] * m[8] - m[5] * m[6]);
		const double h2 = m[2] * (m[3] * m[7] - m[4] * m[6]);
		return h0 - h1 + h2;
	}

	__device__ __forceinline__ double cof3(const double *m, int i, int j)
	{
		const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
		return m[3 * i1 + j1] * m[3 * i2 + j2] - m[3 * i1 + j2] * m[3 * i2 + j1];
	}

	// writes (m^{-1})^T row-major into jt
	__device__ __forceinline__ void inv3_transposed(const double *m, double *jt)
	{
		const double c0 = cof3(m, 0, 0), c1 = cof3(m, 1, 0), c2 = cof3(m, 2, 0);
		const double det = (c0 * m[0] + c1 * m[3]) + c2 * m[6];
		const double invdet = 1.0 / det;
		// inverse(r,c) = cofactor(c,r) * invdet ; jt(r,c) = inverse(c,r) = cofactor(r,c) * invdet
		jt[0] = c0 * invdet;
		jt[3] = c1 * invdet;
		jt[6] = c2 * invdet;
		jt[1] = cof3(m, 0, 1) * invdet;
		jt[4] = cof3(m, 1, 1) * invdet;
		jt[7] = cof3(m, 2, 1) * invdet;
		jt[2] = cof3(m, 0, 2) * invdet;
		jt[5] = cof3(m, 1, 2) * invdet;
		jt[8] = cof3(m, 2, 2) * invdet;
	}

	template <int NLOC, int NGLOC, int NPTS, int BS, int TPE, bool STAGE>
	__global__ __launch_bounds__(256) void k_local_linear(const int32_t *__restrict__ gconn, const double *__restrict__ xyz,
														   const pfx_tables *__restrict__ tab, int64_t n_el, int model, double lambda, double mu,
														   double *__restrict__ kloc)
	{
		constexpr int EPB = 256 / TPE; // elements per block
		constexpr int N = NLOC * BS;
		constexpr int NPAIR = NLOC * (NLOC + 1) / 2;
		__shared__ lin_smem<NLOC, NGLOC, NPTS, BS> sm[EPB];
		__shared__ double tile[STAGE ? EPB * N * N : 1];

		const int grp = threadIdx.x / TPE, lane = threadIdx.x % TPE;
		const int64_t e = (int64_t)blockIdx.x * EPB + grp;
		const bool active = e < n_el;
		lin_smem<NLOC, NGLOC, NPTS, BS> &s = sm[grp];

		if (active)
			for (int t = lane; t < NGLOC * 3; t += TPE)
				s.xs[t] = xyz[3 * (int64_t)gconn[e * NGLOC + t / 3] + t % 3];
		__syncthreads();

		if (active)
			for (int k = lane; k < NPTS; k += TPE)
			{
				double tmp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
				for (int j = 0; j < NGLOC; ++j)
					for (int r = 0; r < 3; ++r)
						for (int c = 0; c < 3; ++c)
							tmp[3 * r + c] += tab->ggrad[k][j][r] * s.xs[3 * j + c] * 1.0;
				s.da[k] = det3_fixed(tmp) * tab->w[k];
				inv3_transposed(tmp, &s.jt[9 * k]);
			}
		__syncthreads();

		if (active)
			for (int t = lane; t < NPTS * NLOC; t += TPE)
			{
				const int k = t / NLOC, i = t % NLOC;
				const double g0 = tab->grad[k][i][0], g1 = tab->grad[k][i][1], g2 = tab->grad[k][i][2];
				const double *jt = &s.jt[9 * k];
				for (int c = 0; c < 3; ++c)
					s.gt[3 * t + c] = (g0 * jt[c] + g1 * jt[3 + c]) + g2 * jt[6 + c];
			}
		__syncthreads();

		double *dst = STAGE ? &tile[grp * N * N] : (kloc + e * (int64_t)(N * N));
		if (active)
			for (int q = lane; q < NPAIR; q += TPE)
			{
				int i = 0;
				while ((i + 1) * (i + 2) / 2 <= q)
					++i;
				const int j = q - i * (i + 1) / 2;
				if (BS == 1)
				{
					double res = 0;
					for (int k = 0; k < NPTS; ++k)
					{
						const double *gi = &s.gt[3 * (k * NLOC + i)], *gj = &s.gt[3 * (k * NLOC + j)];
						const double dot = (gi[0] * gj[0] + gi[1] * gj[1]) + gi[2] * gj[2];
						res += dot * s.da[k];
					}
					if (fabs(res) < 1e-30)
						res = 0.0;
					dst[i * N + j] = res;
					if (j < i)
						dst[j * N + i] = res;
				}
				else
				{
					double res[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
					for (int k = 0; k < NPTS; ++k)
					{
						const double *gi = &s.gt[3 * (k * NLOC + i)], *gj = &s.gt[3 * (k * NLOC + j)];
						double outer[9];
						for (int col = 0; col < 3; ++col)
							for (int row = 0; row < 3; ++row)
								outer[col * 3 + row] = gi[row] * gj[col];
						const double dot = (gi[0] * gj[0] + gi[1] * gj[1]) + gi[2] * gj[2];
						const double dak = s.da[k];
						for (int ii = 0; ii < 3; ++ii)
							for (int jj = 0; jj < 3; ++jj)
							{
								double rk = outer[ii * 3 + jj] * mu + outer[jj * 3 + ii] * lambda;
								if (ii == jj)
									rk += mu * dot;
								res[jj * 3 + ii] += rk * dak;
							}
					}
					// entry (row comp m, col comp n) of block (i,j) is res[n*3+m]  (Assembler.cpp:169,177,182)
					for (int m = 0; m < 3; ++m)
						for (int n = 0; n < 3; ++n)
						{
							double v = res[n * 3 + m];
							if (fabs(v) < 1e-30)
								v = 0.0;
							dst[(i * 3 + m) * N + (j * 3 + n)] = v;
							if (j < i)
								dst[(j * 3 + n) * N + (i * 3 + m)] = v;
						}
				}
			}
		if (STAGE)
		{
			__syncthreads();
			if (active)
			{
				double *out = kloc + e * (int64_t)(N * N);
				for (int t = lane; t < N * N; t += TPE)
					out[t] = tile[grp * N * N + t];
			}
		}
	}

	template <int NLOC, int NGLOC, int NPTS, int BS, int TPE, bool STAGE>
	void launch(pfx_ctx *ctx)
	{
		constexpr int EPB = 256 / TPE;
		const int64_t blocks = (ctx->n_el + EPB - 1) / EPB;
		hipLaunchKernelGGL((k_local_linear<NLOC, NGLOC, NPTS, BS, TPE, STAGE>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
						   ctx->d_gconn, ctx->d_xyz, ctx->d_tab, ctx->n_el, ctx->model, ctx->lambda, ctx->mu, ctx->d_kloc);
	}
} // namespace

pfx_status pfx_launch_local_linear(pfx_ctx *ctx)
{
	const int np = ctx->n_pts;
	const int bs = ctx->bs;
	bool ok = true;
	switch (ctx->kind)
	{
	case PFX_P1TET:
		if (np != 1)
			ok = false;
		else if (bs == 1)
			launch<4, 4, 1, 1, 16, true>(ctx);
		else
			launch<4, 4, 1, 3, 16, true>(ctx);
		break;
	case PFX_P2TET:
		if (np != 6)
			ok = false;
		else if (bs == 1)
			launch<10, 4, 6, 1, 64, true>(ctx);
		else
			launch<10, 4, 6, 3, 64, true>(ctx);
		break;
	case PFX_Q1HEX:
		if (np != 1)
			ok = false;
		else if (bs == 1)
			launch<8, 8, 1, 1, 64, true>(ctx);
		else
			launch<8, 8, 1, 3, 64, true>(ctx);
		break;
	case PFX_Q2HEX:
		if (np != 27)
			ok = false;
		else if (bs == 1)
			launch<27, 8, 27, 1, 256, true>(ctx);
		else
			launch<27, 8, 27, 3, 256, false>(ctx);
		break;
	default:
		ok = false;
	}
	if (!ok)
		PFX_FAIL(PFX_ERR_UNSUPPORTED, "linear assembly: unsupported element kind / quadrature order combination");
	PFX_HIP(hipGetLastError());
	return PFX_OK;
}
