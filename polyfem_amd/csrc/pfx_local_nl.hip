// pfx_local_nl.hip -- per-element NeoHookean energy, gradient and Hessian blocks.
//
// Reference: NLAssembler<NeoHookeanElasticity>::assemble / assemble_grad / assemble_hessian element loops
// (src/polyfem/assembler/Assembler.cpp:638-678, 427-504, 506-636) and the local kernels
// NeoHookeanElasticity::compute_energy_aux (NeoHookeanElasticity.cpp:250-331),
// compute_energy_aux_gradient_fast (:364-453), compute_energy_hessian_aux_fast (:455-565).
//
// With G_i = grad_i * J^{-T} (row i of delF_delU), F = I + sum_i u_i (x) G_i, dJ = cofactor matrix of F
// (columns cross(v,w), cross(w,u), cross(u,v)), w_i = dJ G_i and, per quadrature point,
//     c1 = (mu + lambda (1 - ln J)) / J^2 ,  c2 = (lambda ln J - mu) / J :
//   energy   += da [ mu/2 (|F|^2 - 3 - 2 ln J) + lambda/2 ln^2 J ]
//   grad_i   += da [ mu F G_i + c2 w_i ]                                   (= delF_delU * gradient_temp^T, :440-441)
//   H_ij     += da [ mu (G_i . G_j) I + c1 w_i w_j^T - c2 hat(F (G_i x G_j)) ]
// The last line is the reference's B^T (mu I + c1 g g^T + c2 d2J/dF2) B (:542-563) with the sparsity of B
// (three non-zeros per column) and the hat() structure of d2J/dF2 (:527-532) multiplied out; H_ji = H_ij^T.
//
// Work decomposition (TPE lanes per element, all intermediate data in LDS):
//   1 geometry nodes, connectivity, displacements -> LDS      4 F per point (9 lanes per point)
//   2 J^{-T}, det per quadrature point                       5 per-point scalars, cofactors, energy
//   3 G for every (point, basis)                              6 w for every (point, basis)
//   7 gradient rows (one lane per basis) / Hessian blocks (one lane per pair j <= i, mirrored), staged in an LDS
//     tile and written to the element's slot in HBM with coalesced stores.
#include "pfx_internal.h"

namespace
{
	template <int NLOC, int NGLOC, int NPTS>
	struct nl_smem
	{
		double xs[NGLOC * 3];
		double ul[NLOC * 3];
		double jt[NPTS * 9];
		double da[NPTS];
		double G[NPTS * NLOC * 3];
		double W[NPTS * NLOC * 3];
		double F[NPTS * 9];
		double dJ[NPTS * 9];
		double coef[NPTS * 4]; // mu*da, c1*da, c2*da, energy density * da
		int bad;
	};

	__device__ __forceinline__ void inv3_transposed_det(const double *m, double *jt, double *det)
	{
		const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
		const double c10 = m[2] * m[7] - m[1] * m[8], c11 = m[0] * m[8] - m[2] * m[6], c12 = m[1] * m[6] - m[0] * m[7];
		const double c20 = m[1] * m[5] - m[2] * m[4], c21 = m[2] * m[3] - m[0] * m[5], c22 = m[0] * m[4] - m[1] * m[3];
		const double d = m[0] * c00 + m[1] * c01 + m[2] * c02;
		const double id = 1.0 / d;
		// (m^{-1})^T (r,c) = cofactor(r,c) / det
		jt[0] = c00 * id, jt[1] = c01 * id, jt[2] = c02 * id;
		jt[3] = c10 * id, jt[4] = c11 * id, jt[5] = c12 * id;
		jt[6] = c20 * id, jt[7] = c21 * id, jt[8] = c22 * id;
		*det = d;
	}

	template <int NLOC, int NGLOC, int NPTS, int TPE, bool STAGE>
	__global__ __launch_bounds__(256) void k_local_neohookean(const int32_t *__restrict__ conn, const int32_t *__restrict__ gconn,
																   const double *__restrict__ xyz, const double *__restrict__ u,
																   const pfx_tables *__restrict__ tab, int64_t n_el, double lambda, double mu,
																   int want_e, int want_g, int want_h,
																   double *__restrict__ eloc, double *__restrict__ gloc, double *__restrict__ kloc,
																   int *__restrict__ flag)
	{
		constexpr int EPB = 256 / TPE;
		constexpr int N = NLOC * 3;
		constexpr int NPAIR = NLOC * (NLOC + 1) / 2;
		__shared__ nl_smem<NLOC, NGLOC, NPTS> sm[EPB];
		__shared__ double tile[STAGE ? EPB * N * N : 1];

		const int grp = threadIdx.x / TPE, lane = threadIdx.x % TPE;
		const int64_t e = (int64_t)blockIdx.x * EPB + grp;
		const bool active = e < n_el;
		nl_smem<NLOC, NGLOC, NPTS> &s = sm[grp];

		if (active)
		{
			for (int t = lane; t < NGLOC * 3; t += TPE)
				s.xs[t] = xyz[3 * (int64_t)gconn[e * NGLOC + t / 3] + t % 3];
			for (int t = lane; t < NLOC * 3; t += TPE)
				s.ul[t] = u[3 * (int64_t)conn[e * NLOC + t / 3] + t % 3];
			if (lane == 0)
				s.bad = 0;
		}
		__syncthreads();

		if (active)
			for (int k = lane; k < NPTS; k += TPE)
			{
				double tmp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
				for (int j = 0; j < NGLOC; ++j)
					for (int r = 0; r < 3; ++r)
						for (int c = 0; c < 3; ++c)
							tmp[3 * r + c] += tab->ggrad[k][j][r] * s.xs[3 * j + c];
				double det;
				inv3_transposed_det(tmp, &s.jt[9 * k], &det);
				s.da[k] = det * tab->w[k];
			}
		__syncthreads();

		if (active)
			for (int t = lane; t < NPTS * NLOC; t += TPE)
			{
				const int k = t / NLOC, i = t % NLOC;
				const double g0 = tab->grad[k][i][0], g1 = tab->grad[k][i][1], g2 = tab->grad[k][i][2];
				const double *jt = &s.jt[9 * k];
				for (int c = 0; c < 3; ++c)
					s.G[3 * t + c] = g0 * jt[c] + g1 * jt[3 + c] + g2 * jt[6 + c];
			}
		__syncthreads();

		if (active)
			for (int t = lane; t < NPTS * 9; t += TPE)
			{
				const int k = t / 9, d = (t % 9) / 3, c = t % 3;
				double acc = (d == c) ? 1.0 : 0.0;
				for (int i = 0; i < NLOC; ++i)
					acc += s.ul[3 * i + d] * s.G[3 * (k * NLOC + i) + c];
				s.F[t] = acc; // F(d,c) row-major
			}
		__syncthreads();

		if (active)
			for (int k = lane; k < NPTS; k += TPE)
			{
				const double *F = &s.F[9 * k];
				double *dJ = &s.dJ[9 * k];
				// cofactor matrix: dJ(r,c) = d det / d F(r,c)
				dJ[0] = F[4] * F[8] - F[5] * F[7], dJ[1] = F[5] * F[6] - F[3] * F[8], dJ[2] = F[3] * F[7] - F[4] * F[6];
				dJ[3] = F[2] * F[7] - F[1] * F[8], dJ[4] = F[0] * F[8] - F[2] * F[6], dJ[5] = F[1] * F[6] - F[0] * F[7];
				dJ[6] = F[1] * F[5] - F[2] * F[4], dJ[7] = F[2] * F[3] - F[0] * F[5], dJ[8] = F[0] * F[4] - F[1] * F[3];
				const double J = F[0] * dJ[0] + F[1] * dJ[1] + F[2] * dJ[2];
				if (!(J > 0.0))
					s.bad = 1;
				const double lj = log(J);
				const double da = s.da[k];
				double tr = 0;
				for (int q = 0; q < 9; ++q)
					tr += F[q] * F[q];
				s.coef[4 * k + 0] = mu * da;
				s.coef[4 * k + 1] = (mu + lambda * (1.0 - lj)) / (J * J) * da;
				s.coef[4 * k + 2] = (lambda * lj - mu) / J * da;
				s.coef[4 * k + 3] = (mu / 2 * (tr - 3 - 2 * lj) + lambda / 2 * lj * lj) * da;
			}
		__syncthreads();

		if (active && want_e && lane == 0)
		{
			double en = 0;
			for (int k = 0; k < NPTS; ++k)
				en += s.coef[4 * k + 3];
			eloc[e] = en;
		}
		if (active && lane == 0 && s.bad)
			atomicOr(flag, 1);

		if (active && (want_g || want_h))
			for (int t = lane; t < NPTS * NLOC; t += TPE)
			{
				const int k = t / NLOC;
				const double *dJ = &s.dJ[9 * k];
				const double *g = &s.G[3 * t];
				for (int m = 0; m < 3; ++m)
					s.W[3 * t + m] = dJ[3 * m] * g[0] + dJ[3 * m + 1] * g[1] + dJ[3 * m + 2] * g[2];
			}
		__syncthreads();

		if (active && want_g)
			for (int i = lane; i < NLOC; i += TPE)
			{
				double g[3] = {0, 0, 0};
				for (int k = 0; k < NPTS; ++k)
				{
					const double *F = &s.F[9 * k];
					const double *Gi = &s.G[3 * (k * NLOC + i)], *Wi = &s.W[3 * (k * NLOC + i)];
					const double a = s.coef[4 * k + 0], c2 = s.coef[4 * k + 2];
					for (int d = 0; d < 3; ++d)
						g[d] += a * (F[3 * d] * Gi[0] + F[3 * d + 1] * Gi[1] + F[3 * d + 2] * Gi[2]) + c2 * Wi[d];
				}
				for (int d = 0; d < 3; ++d)
					gloc[e * N + 3 * i + d] = g[d];
			}

		if (want_h)
		{
			double *dst = STAGE ? &tile[grp * N * N] : (kloc + e * (int64_t)(N * N));
			if (active)
				for (int q = lane; q < NPAIR; q += TPE)
				{
					int i = 0;
					while ((i + 1) * (i + 2) / 2 <= q)
						++i;
					const int j = q - i * (i + 1) / 2;
					double b[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
					for (int k = 0; k < NPTS; ++k)
					{
						const double *F = &s.F[9 * k];
						const double *Gi = &s.G[3 * (k * NLOC + i)], *Gj = &s.G[3 * (k * NLOC + j)];
						const double *Wi = &s.W[3 * (k * NLOC + i)], *Wj = &s.W[3 * (k * NLOC + j)];
						const double a = s.coef[4 * k + 0], c1 = s.coef[4 * k + 1], c2 = s.coef[4 * k + 2];
						const double dot = a * (Gi[0] * Gj[0] + Gi[1] * Gj[1] + Gi[2] * Gj[2]);
						const double cx = Gi[1] * Gj[2] - Gi[2] * Gj[1], cy = Gi[2] * Gj[0] - Gi[0] * Gj[2], cz = Gi[0] * Gj[1] - Gi[1] * Gj[0];
						// t = c2 * F (Gi x Gj);  block -= hat(t)
						const double t0 = c2 * (F[0] * cx + F[1] * cy + F[2] * cz);
						const double t1 = c2 * (F[3] * cx + F[4] * cy + F[5] * cz);
						const double t2 = c2 * (F[6] * cx + F[7] * cy + F[8] * cz);
						const double w0 = c1 * Wi[0], w1 = c1 * Wi[1], w2 = c1 * Wi[2];
						b[0] += dot + w0 * Wj[0];
						b[1] += w0 * Wj[1] + t2;
						b[2] += w0 * Wj[2] - t1;
						b[3] += w1 * Wj[0] - t2;
						b[4] += dot + w1 * Wj[1];
						b[5] += w1 * Wj[2] + t0;
						b[6] += w2 * Wj[0] + t1;
						b[7] += w2 * Wj[1] - t0;
						b[8] += dot + w2 * Wj[2];
					}
					for (int m = 0; m < 3; ++m)
						for (int n = 0; n < 3; ++n)
						{
							dst[(i * 3 + m) * N + (j * 3 + n)] = b[3 * m + n];
							if (j < i)
								dst[(j * 3 + n) * N + (i * 3 + m)] = b[3 * m + n];
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
	}

	template <int NLOC, int NGLOC, int NPTS, int TPE, bool STAGE>
	void launch(pfx_ctx *ctx, const double *d_u, bool en, bool gr, bool he)
	{
		constexpr int EPB = 256 / TPE;
		const int64_t blocks = (ctx->n_el + EPB - 1) / EPB;
		hipLaunchKernelGGL((k_local_neohookean<NLOC, NGLOC, NPTS, TPE, STAGE>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
						   ctx->d_conn, ctx->d_gconn, ctx->d_xyz, d_u, ctx->d_tab, ctx->n_el, ctx->lambda, ctx->mu,
						   (int)en, (int)gr, (int)he, ctx->d_eloc, ctx->d_gloc, ctx->d_kloc, ctx->d_flag);
	}
} // namespace

pfx_status pfx_launch_local_nl(pfx_ctx *ctx, const double *d_u, bool energy, bool grad, bool hess)
{
	if (ctx->model != PFX_NEOHOOKEAN)
		PFX_FAIL(PFX_ERR_UNSUPPORTED, "nonlinear assembly is implemented for the NeoHookean model");
	const int np = ctx->n_pts;
	bool ok = true;
	switch (ctx->kind)
	{
	case PFX_P1TET:
		if (np != 1)
			ok = false;
		else
			launch<4, 4, 1, 16, true>(ctx, d_u, energy, grad, hess);
		break;
	case PFX_P2TET:
		if (np != 6)
			ok = false;
		else
			launch<10, 4, 6, 64, true>(ctx, d_u, energy, grad, hess);
		break;
	case PFX_Q1HEX:
		if (np != 1)
			ok = false;
		else
			launch<8, 8, 1, 64, true>(ctx, d_u, energy, grad, hess);
		break;
	case PFX_Q2HEX:
		if (np != 27)
			ok = false;
		else
			launch<27, 8, 27, 256, false>(ctx, d_u, energy, grad, hess);
		break;
	default:
		ok = false;
	}
	if (!ok)
		PFX_FAIL(PFX_ERR_UNSUPPORTED, "nonlinear assembly: unsupported element kind / quadrature order combination");
	PFX_HIP(hipGetLastError());
	return PFX_OK;
}
