// pfx_local_nl.hip -- per-element energy, gradient and Hessian blocks of the nonlinear-assembler interface
// (NeoHookean, SaintVenant, and LinearElasticity used through that interface).
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
// LinearElasticity through the same interface (LinearElasticity.cpp:56-169; the reference differentiates
// compute_energy_aux with autodiff, here the derivatives are written out): with D = sum_i u_i (x) G_i,
// eps = (D + D^T)/2, sigma = 2 mu eps + lambda tr(eps) I:
//   energy += da [ mu eps:eps + lambda/2 tr(eps)^2 ]      grad_i += da sigma G_i
//   H_ij   += da [ mu (G_i . G_j) I + mu G_j G_i^T + lambda G_i G_j^T ]
//
// SaintVenantElasticity (SaintVenantElasticity.cpp:215-312, isotropic elasticity tensor of
// utils/ElasticityUtils.cpp:372-413; gradient and Hessian are autodiff in the reference, written out here): with
// E = (F^T F - I)/2, S = lambda tr(E) I + 2 mu E (second Piola-Kirchhoff stress) and f_m = row m of F:
//   energy += da/2 S:E        grad_i += da F S G_i
//   H_ij   += da [ (G_i^T S G_j) I + lambda (F G_i)(F G_j)^T + mu ( (G_i.G_j) F F^T + (F G_j)(F G_i)^T ) ]
//
// Element kernels: elements with at most ten nodes take one THREAD each, everything in registers (k_local_energy,
// k_local_reg); the 27-node hexahedra take half a wave each, a lane being a quadrature point and then a node
// (k_local_pts).  For the Hessian the per-point state (J^{-T}, F, mu da, c1 da, c2 da) is written out instead of any
// local matrix.
// The Hessian blocks are then computed straight into the global matrix by k_assemble_rows (pfx_rowasm.h) with the
// functor nl_block: one lane per contribution (element, i, j) evaluates H_ij from that state.
#include "pfx_internal.h"
#include "pfx_rowasm.h"
#include "pfx_psd.h"

namespace
{
	// material behind the kernels' MODEL parameter
	enum
	{
		M_LINEAR = 0,
		M_NEO = 1,
		M_STVK = 2,
		M_MULTI = 3, // one of the three per element (MultiModel, assembler/MultiModel.cpp:42-96): the branches become run-time tests
		M_STVKT = 4  // SaintVenant with the material's 6 x 6 elasticity tensor instead of (lambda, mu) (SaintVenantElasticity.cpp:64-68)
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

	// SaintVenant with a general elasticity tensor: S = C : E in the Voigt order xx yy zz yz xz xy with engineering shears
	// (SaintVenantElasticity::compute_energy_aux, assembler/SaintVenantElasticity.cpp:281-307); returns S : E
	__device__ __forceinline__ double stvk_tensor_stress(const double *__restrict__ C, const double *E, double *S)
	{
		const double eps[6] = {E[0], E[4], E[8], 2 * E[5], 2 * E[2], 2 * E[1]};
		double sg[6];
		for (int a = 0; a < 6; ++a)
		{
			double r = C[6 * a] * eps[0];
			for (int b = 1; b < 6; ++b)
				r += C[6 * a + b] * eps[b];
			sg[a] = r;
		}
		S[0] = sg[0], S[1] = sg[5], S[2] = sg[4];
		S[3] = sg[5], S[4] = sg[1], S[5] = sg[3];
		S[6] = sg[4], S[7] = sg[3], S[8] = sg[2];
		double se = 0;
		for (int q = 0; q < 9; ++q)
			se += S[q] * E[q];
		return se;
	}

	// doubles of Hessian state per element: affine geometric map (4-node tets): J^{-T} once, then per point F and the
	// three coefficients (12 doubles); otherwise per point J^{-T}, F and the coefficients
	template <int NGLOC, int NPTS>
	struct nl_state
	{
		// J^{-T} takes 10 doubles (one of padding) so that every group starts on a 16-byte boundary
		static constexpr bool affine = (NGLOC == 4 || NGLOC == 3); // tetrahedra and triangles with the linear geometric map
		static constexpr int stride = affine ? 10 + 12 * NPTS : 22 * NPTS;
		__host__ __device__ static constexpr int jt_at(int k) { return affine ? 0 : 22 * k; }
		__host__ __device__ static constexpr int f_at(int k) { return affine ? 10 + 12 * k : 22 * k + 10; }
	};

	// What the NeoHookean row functor reads instead of (J^-T, F): everything in its block that does not depend on the pair
	// (i, j) is formed once per point by the element kernel.  With g_i = J^-T grad phi_i, cof the cofactor matrix of F:
	//   g_i . g_j = grad phi_i^T S grad phi_j,        S = jt jt^T (symmetric; its six entries go into the slot of J^-T),
	//   cof g_i   = M grad phi_i,                     M = cof jt^T (into the slot of F),
	//   F (g_i x g_j) = (cof g_i) x (cof g_j) / det F (cofactor of the cofactor matrix = det F * F), so c2 is stored divided by det F.
	// jt is the row-major array with g_i[c] = sum_r grad phi_i[r] * jt[3 r + c].
	__device__ __forceinline__ void neo_state_header(const double *jt, double *out)
	{
		out[0] = jt[0] * jt[0] + jt[1] * jt[1] + jt[2] * jt[2];
		out[1] = jt[0] * jt[3] + jt[1] * jt[4] + jt[2] * jt[5];
		out[2] = jt[0] * jt[6] + jt[1] * jt[7] + jt[2] * jt[8];
		out[3] = jt[3] * jt[3] + jt[4] * jt[4] + jt[5] * jt[5];
		out[4] = jt[3] * jt[6] + jt[4] * jt[7] + jt[5] * jt[8];
		out[5] = jt[6] * jt[6] + jt[7] * jt[7] + jt[8] * jt[8];
		out[6] = out[7] = out[8] = 0.0;
	}
	__device__ __forceinline__ void neo_state_point(const double *jt, const double *F, const double *cof, double c0, double c1, double c2, double *out)
	{
		for (int a = 0; a < 3; ++a)
			for (int r = 0; r < 3; ++r)
				out[3 * a + r] = cof[3 * a] * jt[3 * r] + cof[3 * a + 1] * jt[3 * r + 1] + cof[3 * a + 2] * jt[3 * r + 2];
		const double J = F[0] * cof[0] + F[1] * cof[1] + F[2] * cof[2];
		out[9] = c0, out[10] = c1, out[11] = c2 / J;
	}

	// One quadrature point of the energy density and its derivative: F the deformation gradient (the displacement gradient
	// for M_LINEAR), da = det J * weight.  dJ receives what the gradient contracts with the physical basis gradients (NeoHookean:
	// the cofactor matrix of F; otherwise the first Piola-Kirchhoff / Cauchy stress), c0..c2 the coefficients the gradient
	// and the Hessian functor use; returns the point's energy.
	// DIM = 2 (triangles): F holds the 2 x 2 deformation gradient in its upper left, zeros beside it and a one in the corner
	// (zero for M_LINEAR), so that J, the cofactors and the in-plane part of every formula are those of the 2-D problem
	// (NeoHookeanElasticity.cpp:410-417, 501-512); only the trace term of the energy counts the corner and is corrected.
	template <int MODEL, int DIM = 3>
	__device__ __forceinline__ double nl_point(int model, const double *F, double lambda, double mu, const double *__restrict__ tensor, double da, double *dJ,
											   double &c0, double &c1, double &c2, bool &bad)
	{
		double en = 0.0;
		if (model == M_NEO)
		{
			dJ[0] = F[4] * F[8] - F[5] * F[7], dJ[1] = F[5] * F[6] - F[3] * F[8], dJ[2] = F[3] * F[7] - F[4] * F[6];
			dJ[3] = F[2] * F[7] - F[1] * F[8], dJ[4] = F[0] * F[8] - F[2] * F[6], dJ[5] = F[1] * F[6] - F[0] * F[7];
			dJ[6] = F[1] * F[5] - F[2] * F[4], dJ[7] = F[2] * F[3] - F[0] * F[5], dJ[8] = F[0] * F[4] - F[1] * F[3];
			const double J = F[0] * dJ[0] + F[1] * dJ[1] + F[2] * dJ[2];
			if (!(J > 0.0))
				bad = true;
			const double lj = log(J);
			double tr = 0;
			if (DIM == 2)
				tr = F[0] * F[0] + F[1] * F[1] + F[3] * F[3] + F[4] * F[4];
			else
				for (int q = 0; q < 9; ++q)
					tr += F[q] * F[q];
			c0 = mu * da;
			c1 = (mu + lambda * (1.0 - lj)) / (J * J) * da;
			c2 = (lambda * lj - mu) / J * da;
			en = (mu / 2 * (tr - DIM - 2 * lj) + lambda / 2 * lj * lj) * da;
		}
		else if (model == M_STVK || model == M_STVKT)
		{
			double E[9], S[9];
			for (int r = 0; r < 3; ++r)
				for (int c = 0; c < 3; ++c)
					E[3 * r + c] = (F[r] * F[c] + F[3 + r] * F[3 + c] + F[6 + r] * F[6 + c] - (r == c ? 1.0 : 0.0)) / 2;
			const double tr = E[0] + E[4] + E[8];
			double se = 0;
			if (MODEL == M_STVKT)
				se = stvk_tensor_stress(tensor, E, S);
			else
				for (int q = 0; q < 9; ++q)
				{
					S[q] = 2 * mu * E[q];
					if (q == 0 || q == 4 || q == 8)
						S[q] += lambda * tr;
					se += S[q] * E[q];
				}
			for (int r = 0; r < 3; ++r)
				for (int c = 0; c < 3; ++c)
					dJ[3 * r + c] = F[3 * r] * S[c] + F[3 * r + 1] * S[3 + c] + F[3 * r + 2] * S[6 + c];
			c0 = mu * da, c1 = lambda * da, c2 = da;
			en = se / 2 * da;
		}
		else
		{
			double eps[9], ee = 0;
			for (int r = 0; r < 3; ++r)
				for (int c = 0; c < 3; ++c)
				{
					eps[3 * r + c] = (F[3 * r + c] + F[3 * c + r]) / 2;
					ee += eps[3 * r + c] * eps[3 * r + c];
				}
			const double tr = eps[0] + eps[4] + eps[8];
			for (int q = 0; q < 9; ++q)
				dJ[q] = 2 * mu * eps[q];
			dJ[0] += lambda * tr, dJ[4] += lambda * tr, dJ[8] += lambda * tr;
			c0 = mu * da, c1 = lambda * da, c2 = da;
			en = (mu * ee + lambda / 2 * tr * tr) * da;
		}
		return en;
	}

	// Energy only (the line search asks for nothing else): one THREAD per element, everything in registers, the points
	// visited in order (the element's energy is the same sum in every kernel of this file).  No LDS, no barriers.  The reference-gradient tables are read through the constant path
	// (every lane reads the same entry).
	template <int NLOC, int NGLOC, int NPTS, int MODEL, bool PP>
	__global__ __launch_bounds__(256) void k_local_energy(const int32_t *__restrict__ conn, const int32_t *__restrict__ gconn, const double *__restrict__ xyz,
															   const double *__restrict__ u, const pfx_tables *__restrict__ tab, int64_t n_el, double lambda_u,
															   double mu_u, const double *__restrict__ lambda_e, const double *__restrict__ mu_e,
															   const uint8_t *__restrict__ model_e, const double *__restrict__ tensor, double *__restrict__ eloc, int *__restrict__ flag)
	{
		constexpr int DIM = (NGLOC == 3) ? 2 : 3;           // triangles: two displacement components per node
		constexpr bool AFF = (NGLOC == 4 || NGLOC == 3);    // linear geometric map: one Jacobian per element
		const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
		if (e >= n_el)
			return;
		const int model = (MODEL == M_MULTI) ? (int)model_e[e] : MODEL;
		double xs[NGLOC * 3];
		for (int j = 0; j < NGLOC; ++j)
		{
			const int64_t gv = gconn[e * NGLOC + j];
			xs[3 * j] = xyz[3 * gv], xs[3 * j + 1] = xyz[3 * gv + 1], xs[3 * j + 2] = xyz[3 * gv + 2];
		}
		int32_t nd[NLOC];
		for (int i = 0; i < NLOC; ++i)
			nd[i] = conn[e * NLOC + i];
		double en = 0.0;
		bool bad = false;
		double jt[9], det = 0.0;
		for (int k = 0; k < NPTS; ++k)
		{
			if (k == 0 || !AFF) // affine geometry: one Jacobian per element
			{
				double tmp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
				for (int j = 0; j < NGLOC; ++j)
					for (int r = 0; r < 3; ++r)
						for (int c = 0; c < 3; ++c)
							tmp[3 * r + c] += tab->ggrad[AFF ? 0 : k][j][r] * xs[3 * j + c];
				if (DIM == 2)
					tmp[8] = 1.0; // the 2 x 2 Jacobian in the upper left of a 3 x 3 matrix: same determinant, same inverse there
				inv3_transposed_det(tmp, jt, &det);
			}
			const double da = det * tab->w[k];
			double F[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
			if (model != M_LINEAR)
				F[0] = F[4] = F[8] = 1.0;
			for (int i = 0; i < NLOC; ++i)
			{
				const double g0 = tab->grad[k][i][0], g1 = tab->grad[k][i][1], g2 = tab->grad[k][i][2];
				double G[3];
				for (int c = 0; c < 3; ++c)
					G[c] = g0 * jt[c] + g1 * jt[3 + c] + g2 * jt[6 + c];
				const double *ui = u + DIM * (int64_t)nd[i];
				for (int d = 0; d < DIM; ++d)
					for (int c = 0; c < 3; ++c)
						F[3 * d + c] += ui[d] * G[c];
			}
			const double lambda = lambda_e ? lambda_e[PP ? e * NPTS + k : e] : lambda_u, mu = mu_e ? mu_e[PP ? e * NPTS + k : e] : mu_u;
			if (model == M_NEO)
			{
				const double J = F[0] * (F[4] * F[8] - F[5] * F[7]) + F[1] * (F[5] * F[6] - F[3] * F[8]) + F[2] * (F[3] * F[7] - F[4] * F[6]);
				if (!(J > 0.0))
					bad = true;
				const double lj = log(J);
				double tr = 0;
				if (DIM == 2)
					tr = F[0] * F[0] + F[1] * F[1] + F[3] * F[3] + F[4] * F[4];
				else
					for (int q = 0; q < 9; ++q)
						tr += F[q] * F[q];
				en += (mu / 2 * (tr - DIM - 2 * lj) + lambda / 2 * lj * lj) * da;
			}
			else if (model == M_STVK || model == M_STVKT)
			{
				double tr = 0, se = 0;
				double E[9];
				for (int r = 0; r < 3; ++r)
					for (int c = 0; c < 3; ++c)
						E[3 * r + c] = (F[r] * F[c] + F[3 + r] * F[3 + c] + F[6 + r] * F[6 + c] - (r == c ? 1.0 : 0.0)) / 2;
				tr = E[0] + E[4] + E[8];
				if (MODEL == M_STVKT)
				{
					double St[9];
					se = stvk_tensor_stress(tensor, E, St);
				}
				else
					for (int q = 0; q < 9; ++q)
					{
						double S = 2 * mu * E[q];
						if (q == 0 || q == 4 || q == 8)
							S += lambda * tr;
						se += S * E[q];
					}
				en += se / 2 * da;
			}
			else
			{
				double ee = 0, eps[9];
				for (int r = 0; r < 3; ++r)
					for (int c = 0; c < 3; ++c)
					{
						eps[3 * r + c] = (F[3 * r + c] + F[3 * c + r]) / 2;
						ee += eps[3 * r + c] * eps[3 * r + c];
					}
				const double tr = eps[0] + eps[4] + eps[8];
				en += (mu * ee + lambda / 2 * tr * tr) * da;
			}
		}
		eloc[e] = en;
		if (bad)
			atomicOr(flag, 1);
	}

	// The same register-only scheme for the gradient (WG) and / or the per-point state the row assembly builds the Hessian
	// from (WH): one thread per element; the energy comes for free.  The 27-node elements, whose 81 gradient
	// accumulators do not fit in registers, take k_local_pts.
	template <int NLOC, int NGLOC, int NPTS, int MODEL, bool PP, bool WG, bool WH>
	__global__ __launch_bounds__(256) void k_local_reg(const int32_t *__restrict__ conn, const int32_t *__restrict__ gconn, const double *__restrict__ xyz,
															const double *__restrict__ u, const pfx_tables *__restrict__ tab, int64_t n_el, double lambda_u,
															double mu_u, const double *__restrict__ lambda_e, const double *__restrict__ mu_e,
															const uint8_t *__restrict__ model_e, const double *__restrict__ tensor, double *__restrict__ eloc,
															double *__restrict__ gloc, double *__restrict__ estate, int *__restrict__ flag)
	{
		typedef nl_state<NGLOC, NPTS> ST;
		constexpr int DIM = (NGLOC == 3) ? 2 : 3;           // triangles: two displacement components per node
		constexpr bool AFF = (NGLOC == 4 || NGLOC == 3);    // linear geometric map: one Jacobian per element
		const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
		if (e >= n_el)
			return;
		const int model = (MODEL == M_MULTI) ? (int)model_e[e] : MODEL;
		double xs[NGLOC * 3];
		for (int j = 0; j < NGLOC; ++j)
		{
			const int64_t gv = gconn[e * NGLOC + j];
			xs[3 * j] = xyz[3 * gv], xs[3 * j + 1] = xyz[3 * gv + 1], xs[3 * j + 2] = xyz[3 * gv + 2];
		}
		int32_t nd[NLOC];
		for (int i = 0; i < NLOC; ++i)
			nd[i] = conn[e * NLOC + i];
		double g[WG ? NLOC * 3 : 1];
		if (WG)
			for (int t = 0; t < NLOC * 3; ++t)
				g[t] = 0.0;
		double *st = WH ? estate + e * (int64_t)ST::stride : nullptr;
		double en = 0.0;
		bool bad = false;
		double jt[9], det = 0.0;
		for (int k = 0; k < NPTS; ++k)
		{
			if (k == 0 || !AFF)
			{
				double tmp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
				for (int j = 0; j < NGLOC; ++j)
					for (int r = 0; r < 3; ++r)
						for (int c = 0; c < 3; ++c)
							tmp[3 * r + c] += tab->ggrad[AFF ? 0 : k][j][r] * xs[3 * j + c];
				if (DIM == 2)
					tmp[8] = 1.0; // the 2 x 2 Jacobian in the upper left of a 3 x 3 matrix: same determinant, same inverse there
				inv3_transposed_det(tmp, jt, &det);
				if (WH)
				{
					if (model == M_NEO)
						neo_state_header(jt, st + ST::jt_at(k));
					else
						for (int q = 0; q < 9; ++q)
							st[ST::jt_at(k) + q] = jt[q];
					st[ST::jt_at(k) + 9] = (double)model;
				}
			}
			const double da = det * tab->w[k];
			double F[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
			if (model != M_LINEAR)
				F[0] = F[4] = F[8] = 1.0;
			for (int i = 0; i < NLOC; ++i)
			{
				const double g0 = tab->grad[k][i][0], g1 = tab->grad[k][i][1], g2 = tab->grad[k][i][2];
				double G[3];
				for (int c = 0; c < 3; ++c)
					G[c] = g0 * jt[c] + g1 * jt[3 + c] + g2 * jt[6 + c];
				const double *ui = u + DIM * (int64_t)nd[i];
				for (int d = 0; d < DIM; ++d)
					for (int c = 0; c < 3; ++c)
						F[3 * d + c] += ui[d] * G[c];
			}
			const double lambda = lambda_e ? lambda_e[PP ? e * NPTS + k : e] : lambda_u, mu = mu_e ? mu_e[PP ? e * NPTS + k : e] : mu_u;
			double dJ[9], c0, c1, c2;
			en += nl_point<MODEL, DIM>(model, F, lambda, mu, tensor, da, dJ, c0, c1, c2, bad);
			if (WH)
			{
				if (model == M_NEO)
					neo_state_point(jt, F, dJ, c0, c1, c2, st + ST::f_at(k));
				else
				{
					for (int q = 0; q < 9; ++q)
						st[ST::f_at(k) + q] = F[q];
					st[ST::f_at(k) + 9] = c0, st[ST::f_at(k) + 10] = c1, st[ST::f_at(k) + 11] = c2;
				}
			}
			if (WG)
				for (int i = 0; i < NLOC; ++i)
				{
					const double g0 = tab->grad[k][i][0], g1 = tab->grad[k][i][1], g2 = tab->grad[k][i][2];
					double G[3];
					for (int c = 0; c < 3; ++c)
						G[c] = g0 * jt[c] + g1 * jt[3 + c] + g2 * jt[6 + c];
					for (int d = 0; d < 3; ++d)
					{
						const double sg = dJ[3 * d] * G[0] + dJ[3 * d + 1] * G[1] + dJ[3 * d + 2] * G[2];
						if (model == M_NEO)
							g[3 * i + d] += c0 * (F[3 * d] * G[0] + F[3 * d + 1] * G[1] + F[3 * d + 2] * G[2]) + c2 * sg;
						else
							g[3 * i + d] += c2 * sg;
					}
				}
		}
		eloc[e] = en;
		if (WG)
			for (int i = 0; i < NLOC; ++i)
				for (int d = 0; d < DIM; ++d)
					gloc[e * (NLOC * DIM) + i * DIM + d] = g[3 * i + d];
		if (bad)
			atomicOr(flag, 1);
	}

	// Gradient and / or Hessian state of the elements with many nodes and points (27-node hexahedra, 27 points): a lane is a
	// quadrature point while the point values are formed and a node while the gradient is summed, two elements per wave (lanes
	// 0..31 and 32..63).  The physical basis gradients G_i = J^{-T} grad phi_i are never formed: with H = sum_i u_i (x)
	// grad phi_i (reference gradients, one table in LDS for the whole workgroup) F = I + H J^{-1}, and the gradient
	// sum_k X_k G_i(k) is sum_k (X_k J_k^{-1}) grad phi_i(xi_k).  Per element the energy is the sum over the points in order
	// and the gradient the sum over the points in order.
	// PP: the material arrays hold one value per (element, point) instead of one per element
	template <int NLOC, int NGLOC, int NPTS, int MODEL, bool PP, int WAVES>
	__global__ __launch_bounds__(64 * WAVES) void k_local_pts(const int32_t *__restrict__ conn, const int32_t *__restrict__ gconn, const double *__restrict__ xyz,
																 const double *__restrict__ u, const pfx_tables *__restrict__ tab, int64_t n_el, double lambda_u,
																 double mu_u, const double *__restrict__ lambda_e, const double *__restrict__ mu_e,
																 const uint8_t *__restrict__ model_e, const double *__restrict__ tensor, int want_g, int want_h,
																 double *__restrict__ eloc, double *__restrict__ gloc, double *__restrict__ estate,
																 int *__restrict__ flag)
	{
		static_assert(NLOC <= 32 && NPTS <= 32 && NGLOC <= 32, "a half wave per element");
		typedef nl_state<NGLOC, NPTS> ST;
		constexpr int EPB = 2 * WAVES;
		__shared__ double s_grad[NPTS * NLOC * 3];   // grad phi_i(xi_k), [k][i][3]
		__shared__ double s_ggrad[NPTS * NGLOC * 3]; // the same for the geometric map
		__shared__ double s_w[NPTS];
		__shared__ double s_u[EPB][NLOC * 3];
		__shared__ double s_xs[EPB][NGLOC * 3];
		__shared__ double s_q[EPB][NPTS * 9]; // X_k J_k^{-1}
		__shared__ double s_en[EPB][NPTS];
		for (int t = threadIdx.x; t < NPTS * NLOC * 3; t += 64 * WAVES)
			s_grad[t] = tab->grad[t / (NLOC * 3)][(t / 3) % NLOC][t % 3];
		for (int t = threadIdx.x; t < NPTS * NGLOC * 3; t += 64 * WAVES)
			s_ggrad[t] = tab->ggrad[t / (NGLOC * 3)][(t / 3) % NGLOC][t % 3];
		for (int t = threadIdx.x; t < NPTS; t += 64 * WAVES)
			s_w[t] = tab->w[t];
		const int l = threadIdx.x & 31;
		const int slot = threadIdx.x >> 5; // element of the workgroup
		const int64_t e = (int64_t)blockIdx.x * EPB + slot;
		const bool active = e < n_el;
		if (active)
		{
			if (l < NLOC)
			{
				const double *up = u + 3 * (int64_t)conn[e * NLOC + l];
				s_u[slot][3 * l] = up[0], s_u[slot][3 * l + 1] = up[1], s_u[slot][3 * l + 2] = up[2];
			}
			if (l < NGLOC)
			{
				const double *xp = xyz + 3 * (int64_t)gconn[e * NGLOC + l];
				s_xs[slot][3 * l] = xp[0], s_xs[slot][3 * l + 1] = xp[1], s_xs[slot][3 * l + 2] = xp[2];
			}
		}
		__syncthreads();
		// from here on the two elements of a wave only use their own slots
		auto wave_sync = [] {
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		};
		const int model = (MODEL == M_MULTI) ? (active ? (int)model_e[e] : (int)M_LINEAR) : MODEL;
		bool bad = false;
		if (active && l < NPTS)
		{
			const int k = l;
			double tmp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 2
			for (int j = 0; j < NGLOC; ++j)
				for (int r = 0; r < 3; ++r)
					for (int c = 0; c < 3; ++c)
						tmp[3 * r + c] += s_ggrad[(k * NGLOC + j) * 3 + r] * s_xs[slot][3 * j + c];
			double jt[9], det;
			inv3_transposed_det(tmp, jt, &det);
			const double da = det * s_w[k];
			double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 3
			for (int i = 0; i < NLOC; ++i)
			{
				const double *gr = s_grad + (k * NLOC + i) * 3;
				const double *ui = s_u[slot] + 3 * i;
				for (int d = 0; d < 3; ++d)
					for (int r = 0; r < 3; ++r)
						H[3 * d + r] += ui[d] * gr[r];
			}
			double F[9];
			for (int d = 0; d < 3; ++d)
				for (int c = 0; c < 3; ++c)
					F[3 * d + c] = ((model != M_LINEAR && d == c) ? 1.0 : 0.0) + (H[3 * d] * jt[c] + H[3 * d + 1] * jt[3 + c] + H[3 * d + 2] * jt[6 + c]);
			const double lambda = lambda_e ? lambda_e[PP ? e * NPTS + k : e] : lambda_u, mu = mu_e ? mu_e[PP ? e * NPTS + k : e] : mu_u;
			double dJ[9], c0, c1, c2;
			s_en[slot][k] = nl_point<MODEL>(model, F, lambda, mu, tensor, da, dJ, c0, c1, c2, bad);
			if (want_g)
				for (int d = 0; d < 3; ++d)
				{
					double X[3];
					for (int c = 0; c < 3; ++c)
						X[c] = (model == M_NEO) ? c0 * F[3 * d + c] + c2 * dJ[3 * d + c] : c2 * dJ[3 * d + c];
					for (int r = 0; r < 3; ++r)
						s_q[slot][9 * k + 3 * d + r] = X[0] * jt[3 * r] + X[1] * jt[3 * r + 1] + X[2] * jt[3 * r + 2];
				}
			if (want_h)
			{
				double *st = estate + e * (int64_t)ST::stride;
				if (model == M_NEO)
				{
					neo_state_header(jt, st + ST::jt_at(k));
					neo_state_point(jt, F, dJ, c0, c1, c2, st + ST::f_at(k));
				}
				else
				{
					for (int q = 0; q < 9; ++q)
						st[ST::jt_at(k) + q] = jt[q];
					for (int q = 0; q < 9; ++q)
						st[ST::f_at(k) + q] = F[q];
					st[ST::f_at(k) + 9] = c0, st[ST::f_at(k) + 10] = c1, st[ST::f_at(k) + 11] = c2;
				}
				st[ST::jt_at(k) + 9] = (double)model; // the tenth slot tells the row functor the element's model
			}
		}
		wave_sync();
		if (active && l == 0)
		{
			double en = 0;
			for (int k = 0; k < NPTS; ++k)
				en += s_en[slot][k];
			eloc[e] = en;
		}
		if (active && want_g && l < NLOC)
		{
			const int i = l;
			double g[3] = {0, 0, 0};
#pragma unroll 3
			for (int k = 0; k < NPTS; ++k)
			{
				const double *q = s_q[slot] + 9 * k;
				const double *gr = s_grad + (k * NLOC + i) * 3;
				for (int d = 0; d < 3; ++d)
					g[d] += q[3 * d] * gr[0] + q[3 * d + 1] * gr[1] + q[3 * d + 2] * gr[2];
			}
			double *go = gloc + e * (NLOC * 3) + 3 * i;
			go[0] = g[0], go[1] = g[1], go[2] = g[2];
		}
		if (bad)
			atomicOr(flag, 1);
	}

	// H_ij for the pair (i, j), j <= i : out[m*3+n] = d2 E / d u_{i,m} d u_{j,n}
#ifndef PFX_NL_UNROLL
#define PFX_NL_UNROLL 3 // quadrature points per trip of the point loop: more keeps more state loads in flight, and more registers
#endif
// NeoHookean reads a state that leaves little arithmetic per point (neo_state_point): one point per trip in 106 registers
// lets four waves share a SIMD, which pays more than loads kept in flight by one wave (10 M P2 tets: 23.8 ms; three
// points per trip at three waves 32.9 ms, two at four 28.3 ms, one at five 28.5 ms)
#ifndef PFX_NEO_UNROLL
#define PFX_NEO_UNROLL 1
#endif
#ifndef PFX_NEO_MINWAVES
#define PFX_NEO_MINWAVES 4
#endif
	template <int NLOC, int NGLOC, int NPTS, int MODEL>
	struct nl_block
	{
		const double *estate;
		const pfx_tables *tab;
		const double *tensor = nullptr;                                                   // M_STVKT: the 6 x 6 elasticity tensor
		static constexpr int LDS_DOUBLES = NPTS * NLOC * 3 + (MODEL == M_STVKT ? 36 : 0); // grad[k][i][3], tensor
		static constexpr int MIN_WAVES = MODEL == M_STVKT ? 2 : MODEL == M_NEO ? PFX_NEO_MINWAVES : PFX_ROWASM_MINWAVES; // the tensor products want the registers of a third wave
		static constexpr int BS = (NGLOC == 3) ? 2 : 3; // triangles: the in-plane 2 x 2 part of the block computed below

		__device__ __forceinline__ void load_tables(double *lds, int lane) const
		{
			for (int t = lane; t < NPTS * NLOC * 3; t += 64)
				lds[t] = tab->grad[t / (NLOC * 3)][(t / 3) % NLOC][t % 3];
			if (MODEL == M_STVKT && lane < 36)
				lds[NPTS * NLOC * 3 + lane] = tensor[lane];
		}

		__device__ __forceinline__ void operator()(const double *lds, uint32_t e, int i, int j, double (&out)[BS * BS]) const
		{
			typedef nl_state<NGLOC, NPTS> ST;
			typedef double d2 __attribute__((ext_vector_type(2)));
			const d2 *st = reinterpret_cast<const d2 *>(estate + (int64_t)e * ST::stride);
			double jt[10], F[12];
			if (ST::affine)
				for (int q = 0; q < 5; ++q)
				{
					const d2 v = st[q];
					jt[2 * q] = v.x, jt[2 * q + 1] = v.y;
				}
			double b[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll(MODEL == M_STVKT ? 1 : MODEL == M_MULTI ? 2 : MODEL == M_NEO ? PFX_NEO_UNROLL : PFX_NL_UNROLL)
			for (int k = 0; k < NPTS; ++k)
			{
				if (!ST::affine)
					for (int q = 0; q < 5; ++q)
					{
						const d2 v = st[ST::jt_at(k) / 2 + q];
						jt[2 * q] = v.x, jt[2 * q + 1] = v.y;
					}
				for (int q = 0; q < 6; ++q)
				{
					const d2 v = st[ST::f_at(k) / 2 + q];
					F[2 * q] = v.x, F[2 * q + 1] = v.y;
				}
				const double a = F[9], c1 = F[10], c2 = F[11];
				const double *ri = lds + (k * NLOC + i) * 3, *rj = lds + (k * NLOC + j) * 3;
				const int model = (MODEL == M_MULTI) ? (int)jt[9] : MODEL;
				if (model == M_NEO)
				{
					// the state holds S (jt[0..5]: xx xy xz yy yz zz), M (F[0..8]) and c2 / det F (neo_state_header / _point):
					// block = a (ri^T S rj) I + c1 (M ri)(M rj)^T - hat(c2 (M ri) x (M rj))
					const double s0 = jt[0] * rj[0] + jt[1] * rj[1] + jt[2] * rj[2];
					const double s1 = jt[1] * rj[0] + jt[3] * rj[1] + jt[4] * rj[2];
					const double s2 = jt[2] * rj[0] + jt[4] * rj[1] + jt[5] * rj[2];
					const double dot = a * (ri[0] * s0 + ri[1] * s1 + ri[2] * s2);
					const double vi0 = F[0] * ri[0] + F[1] * ri[1] + F[2] * ri[2];
					const double vi1 = F[3] * ri[0] + F[4] * ri[1] + F[5] * ri[2];
					const double vi2 = F[6] * ri[0] + F[7] * ri[1] + F[8] * ri[2];
					const double wj0 = F[0] * rj[0] + F[1] * rj[1] + F[2] * rj[2];
					const double wj1 = F[3] * rj[0] + F[4] * rj[1] + F[5] * rj[2];
					const double wj2 = F[6] * rj[0] + F[7] * rj[1] + F[8] * rj[2];
					const double t0 = c2 * (vi1 * wj2 - vi2 * wj1), t1 = c2 * (vi2 * wj0 - vi0 * wj2), t2 = c2 * (vi0 * wj1 - vi1 * wj0);
					const double wi0 = c1 * vi0, wi1 = c1 * vi1, wi2 = c1 * vi2;
					b[0] += dot + wi0 * wj0;
					b[1] += wi0 * wj1 + t2;
					b[2] += wi0 * wj2 - t1;
					b[3] += wi1 * wj0 - t2;
					b[4] += dot + wi1 * wj1;
					b[5] += wi1 * wj2 + t0;
					b[6] += wi2 * wj0 + t1;
					b[7] += wi2 * wj1 - t0;
					b[8] += dot + wi2 * wj2;
					continue;
				}
				double Gi[3], Gj[3];
				for (int c = 0; c < 3; ++c)
				{
					Gi[c] = ri[0] * jt[c] + ri[1] * jt[3 + c] + ri[2] * jt[6 + c];
					Gj[c] = rj[0] * jt[c] + rj[1] * jt[3 + c] + rj[2] * jt[6 + c];
				}
				const double dot = a * (Gi[0] * Gj[0] + Gi[1] * Gj[1] + Gi[2] * Gj[2]);
				if (model == M_LINEAR)
				{
					// a = mu da, c1 = lambda da
					for (int m = 0; m < 3; ++m)
						for (int n = 0; n < 3; ++n)
							b[3 * m + n] += a * Gi[n] * Gj[m] + c1 * Gi[m] * Gj[n] + (m == n ? dot : 0.0);
					continue;
				}
				if (MODEL == M_STVKT)
				{
					// c2 = da; block(m, n) = da [ voigt(f_m, Gi)^T C voigt(f_n, Gj) + delta_mn Gi^T S Gj ], f_m = row m of F,
					// voigt(f, g) = the Voigt strain (engineering shears) of sym(f (x) g), S = C : E
					const double *C = lds + NPTS * NLOC * 3;
					double E[9], S[9];
					for (int r = 0; r < 3; ++r)
						for (int c = r; c < 3; ++c)
							E[3 * r + c] = E[3 * c + r] = (F[r] * F[c] + F[3 + r] * F[3 + c] + F[6 + r] * F[6 + c] - (r == c ? 1.0 : 0.0)) / 2;
					stvk_tensor_stress(C, E, S);
					const double sg0 = S[0] * Gj[0] + S[1] * Gj[1] + S[2] * Gj[2];
					const double sg1 = S[3] * Gj[0] + S[4] * Gj[1] + S[5] * Gj[2];
					const double sg2 = S[6] * Gj[0] + S[7] * Gj[1] + S[8] * Gj[2];
					const double gsg = Gi[0] * sg0 + Gi[1] * sg1 + Gi[2] * sg2;
					double cw[3][6];
					for (int n = 0; n < 3; ++n)
					{
						const double *f = F + 3 * n;
						const double w[6] = {f[0] * Gj[0], f[1] * Gj[1], f[2] * Gj[2], f[1] * Gj[2] + Gj[1] * f[2], f[0] * Gj[2] + Gj[0] * f[2], f[0] * Gj[1] + Gj[0] * f[1]};
						for (int r = 0; r < 6; ++r)
						{
							double acc = C[6 * r] * w[0];
							for (int c = 1; c < 6; ++c)
								acc += C[6 * r + c] * w[c];
							cw[n][r] = acc;
						}
					}
					for (int m = 0; m < 3; ++m)
					{
						const double *f = F + 3 * m;
						const double v[6] = {f[0] * Gi[0], f[1] * Gi[1], f[2] * Gi[2], f[1] * Gi[2] + Gi[1] * f[2], f[0] * Gi[2] + Gi[0] * f[2], f[0] * Gi[1] + Gi[0] * f[1]};
						for (int n = 0; n < 3; ++n)
						{
							double acc = (m == n) ? gsg : 0.0;
							for (int r = 0; r < 6; ++r)
								acc += v[r] * cw[n][r];
							b[3 * m + n] += acc * c2;
						}
					}
					continue;
				}
				if (model == M_STVK)
				{
					// a = mu da, c1 = lambda da; S da = c1 tr(E) I + 2 a E from F
					double E[6]; // xx, yy, zz, yz, xz, xy of (F^T F - I)/2
					E[0] = (F[0] * F[0] + F[3] * F[3] + F[6] * F[6] - 1.0) / 2;
					E[1] = (F[1] * F[1] + F[4] * F[4] + F[7] * F[7] - 1.0) / 2;
					E[2] = (F[2] * F[2] + F[5] * F[5] + F[8] * F[8] - 1.0) / 2;
					E[3] = (F[1] * F[2] + F[4] * F[5] + F[7] * F[8]) / 2;
					E[4] = (F[0] * F[2] + F[3] * F[5] + F[6] * F[8]) / 2;
					E[5] = (F[0] * F[1] + F[3] * F[4] + F[6] * F[7]) / 2;
					const double ltr = c1 * (E[0] + E[1] + E[2]);
					const double s0 = ltr + 2 * a * E[0], s1 = ltr + 2 * a * E[1], s2 = ltr + 2 * a * E[2];
					const double s3 = 2 * a * E[3], s4 = 2 * a * E[4], s5 = 2 * a * E[5];
					const double sg0 = s0 * Gj[0] + s5 * Gj[1] + s4 * Gj[2];
					const double sg1 = s5 * Gj[0] + s1 * Gj[1] + s3 * Gj[2];
					const double sg2 = s4 * Gj[0] + s3 * Gj[1] + s2 * Gj[2];
					const double gsg = Gi[0] * sg0 + Gi[1] * sg1 + Gi[2] * sg2;
					double fgi[3], fgj[3];
					for (int m = 0; m < 3; ++m)
					{
						fgi[m] = F[3 * m] * Gi[0] + F[3 * m + 1] * Gi[1] + F[3 * m + 2] * Gi[2];
						fgj[m] = F[3 * m] * Gj[0] + F[3 * m + 1] * Gj[1] + F[3 * m + 2] * Gj[2];
					}
					for (int m = 0; m < 3; ++m)
						for (int n = 0; n < 3; ++n)
						{
							const double fft = F[3 * m] * F[3 * n] + F[3 * m + 1] * F[3 * n + 1] + F[3 * m + 2] * F[3 * n + 2];
							b[3 * m + n] += c1 * fgi[m] * fgj[n] + dot * fft + a * fgj[m] * fgi[n] + (m == n ? gsg : 0.0);
						}
					continue;
				}
			}
			for (int m = 0; m < BS; ++m)
				for (int n = 0; n < BS; ++n)
					out[m * BS + n] = b[3 * m + n];
		}
	};

	template <int NLOC, int NGLOC, int NPTS>
	void launch(pfx_ctx *ctx, const double *d_u, bool en, bool gr, bool he, double *eloc)
	{
		if (en && !gr && !he && NLOC <= 10) // energy alone: the register-only kernel
		{
			const unsigned eb = (unsigned)((ctx->n_el + 255) / 256);
#define PFX_LOCAL_E_PP(MODEL_, PP_)                                                                                                                   \
	hipLaunchKernelGGL((k_local_energy<NLOC, NGLOC, NPTS, MODEL_, PP_>), dim3(eb), dim3(256), 0, ctx->stream, ctx->d_conn, ctx->d_gconn, ctx->d_xyz, d_u, \
					   ctx->d_tab, ctx->n_el, ctx->lambda, ctx->mu, ctx->d_lambda_e, ctx->d_mu_e, ctx->d_model_e, ctx->d_tensor, eloc, ctx->d_flag)
#define PFX_LOCAL_E(MODEL_)                            \
	do                                                 \
	{                                                  \
		if (ctx->mat_per_point && ctx->d_lambda_e)     \
			PFX_LOCAL_E_PP(MODEL_, true);              \
		else                                           \
			PFX_LOCAL_E_PP(MODEL_, false);             \
	} while (0)
			if (ctx->model == PFX_NEOHOOKEAN)
				PFX_LOCAL_E(M_NEO);
			else if (ctx->model == PFX_SAINT_VENANT && ctx->stvk_tensor)
				PFX_LOCAL_E(M_STVKT);
			else if (ctx->model == PFX_SAINT_VENANT)
				PFX_LOCAL_E(M_STVK);
			else if (ctx->model == PFX_MULTI_MODEL)
				PFX_LOCAL_E(M_MULTI);
			else
				PFX_LOCAL_E(M_LINEAR);
#undef PFX_LOCAL_E
#undef PFX_LOCAL_E_PP
			return;
		}
		if ((gr || he) && NLOC <= 10)
		{
			const unsigned eb = (unsigned)((ctx->n_el + 255) / 256);
#define PFX_LOCAL_R3(MODEL_, PP_, WG_, WH_)                                                                                                                   \
	hipLaunchKernelGGL((k_local_reg<NLOC, NGLOC, NPTS, MODEL_, PP_, WG_, WH_>), dim3(eb), dim3(256), 0, ctx->stream, ctx->d_conn, ctx->d_gconn, ctx->d_xyz, d_u, \
					   ctx->d_tab, ctx->n_el, ctx->lambda, ctx->mu, ctx->d_lambda_e, ctx->d_mu_e, ctx->d_model_e, ctx->d_tensor, eloc, ctx->d_gloc, ctx->d_estate,     \
					   ctx->d_flag)
#define PFX_LOCAL_R2(MODEL_, PP_)              \
	do                                         \
	{                                          \
		if (gr && he)                          \
			PFX_LOCAL_R3(MODEL_, PP_, true, true);  \
		else if (gr)                           \
			PFX_LOCAL_R3(MODEL_, PP_, true, false); \
		else                                   \
			PFX_LOCAL_R3(MODEL_, PP_, false, true); \
	} while (0)
#define PFX_LOCAL_R(MODEL_)                            \
	do                                                 \
	{                                                  \
		if (ctx->mat_per_point && ctx->d_lambda_e)     \
			PFX_LOCAL_R2(MODEL_, true);                \
		else                                           \
			PFX_LOCAL_R2(MODEL_, false);               \
	} while (0)
			if (ctx->model == PFX_NEOHOOKEAN)
				PFX_LOCAL_R(M_NEO);
			else if (ctx->model == PFX_SAINT_VENANT && ctx->stvk_tensor)
				PFX_LOCAL_R(M_STVKT);
			else if (ctx->model == PFX_SAINT_VENANT)
				PFX_LOCAL_R(M_STVK);
			else if (ctx->model == PFX_MULTI_MODEL)
				PFX_LOCAL_R(M_MULTI);
			else
				PFX_LOCAL_R(M_LINEAR);
#undef PFX_LOCAL_R
#undef PFX_LOCAL_R2
#undef PFX_LOCAL_R3
			return;
		}
#define PFX_LOCAL_NL_PP(MODEL_, PP_)                                                                                                              \
	hipLaunchKernelGGL((k_local_pts<NLOC, NGLOC, NPTS, MODEL_, PP_, 8>), dim3((unsigned)((ctx->n_el + 15) / 16)), dim3(512), 0, ctx->stream, ctx->d_conn, ctx->d_gconn, \
					   ctx->d_xyz, d_u, ctx->d_tab, ctx->n_el, ctx->lambda, ctx->mu, ctx->d_lambda_e, ctx->d_mu_e, ctx->d_model_e, ctx->d_tensor, (int)gr, (int)he,     \
					   eloc, ctx->d_gloc, ctx->d_estate, ctx->d_flag)
#define PFX_LOCAL_NL(MODEL_)                               \
	do                                                     \
	{                                                      \
		if (ctx->mat_per_point && ctx->d_lambda_e)         \
			PFX_LOCAL_NL_PP(MODEL_, true);                 \
		else                                               \
			PFX_LOCAL_NL_PP(MODEL_, false);                \
	} while (0)
		if constexpr (NLOC > 10)
		{
			if (ctx->model == PFX_NEOHOOKEAN)
				PFX_LOCAL_NL(M_NEO);
			else if (ctx->model == PFX_SAINT_VENANT && ctx->stvk_tensor)
				PFX_LOCAL_NL(M_STVKT);
			else if (ctx->model == PFX_SAINT_VENANT)
				PFX_LOCAL_NL(M_STVK);
			else if (ctx->model == PFX_MULTI_MODEL)
				PFX_LOCAL_NL(M_MULTI);
			else
				PFX_LOCAL_NL(M_LINEAR);
		}
#undef PFX_LOCAL_NL
#undef PFX_LOCAL_NL_PP
	}

	template <int NLOC, int NGLOC, int NPTS>
	void launch_rows(pfx_ctx *ctx)
	{
		constexpr int BS = (NGLOC == 3) ? 2 : 3; // triangles: 2 x 2 blocks
		if (ctx->model == PFX_NEOHOOKEAN)
			pfx_launch_assemble_rows<NLOC, BS>(ctx, nl_block<NLOC, NGLOC, NPTS, M_NEO>{ctx->d_estate, ctx->d_tab}, false);
		else if (ctx->model == PFX_SAINT_VENANT && ctx->stvk_tensor)
			pfx_launch_assemble_rows<NLOC, BS>(ctx, nl_block<NLOC, NGLOC, NPTS, M_STVKT>{ctx->d_estate, ctx->d_tab, ctx->d_tensor}, false);
		else if (ctx->model == PFX_SAINT_VENANT)
			pfx_launch_assemble_rows<NLOC, BS>(ctx, nl_block<NLOC, NGLOC, NPTS, M_STVK>{ctx->d_estate, ctx->d_tab}, false);
		else if (ctx->model == PFX_MULTI_MODEL)
			pfx_launch_assemble_rows<NLOC, BS>(ctx, nl_block<NLOC, NGLOC, NPTS, M_MULTI>{ctx->d_estate, ctx->d_tab}, false);
		else
			pfx_launch_assemble_rows<NLOC, BS>(ctx, nl_block<NLOC, NGLOC, NPTS, M_LINEAR>{ctx->d_estate, ctx->d_tab}, false);
	}
	template <int NLOC, int NGLOC, int NPTS>
	pfx_status launch_psd(pfx_ctx *ctx)
	{
		pfx_status rc;
		constexpr int BS = (NGLOC == 3) ? 2 : 3; // triangles: 2 x 2 blocks
		if (ctx->model == PFX_NEOHOOKEAN)
			rc = pfx_launch_project_psd<NLOC, BS>(ctx, nl_block<NLOC, NGLOC, NPTS, M_NEO>{ctx->d_estate, ctx->d_tab});
		else if (ctx->model == PFX_SAINT_VENANT && ctx->stvk_tensor)
			rc = pfx_launch_project_psd<NLOC, BS>(ctx, nl_block<NLOC, NGLOC, NPTS, M_STVKT>{ctx->d_estate, ctx->d_tab, ctx->d_tensor});
		else if (ctx->model == PFX_SAINT_VENANT)
			rc = pfx_launch_project_psd<NLOC, BS>(ctx, nl_block<NLOC, NGLOC, NPTS, M_STVK>{ctx->d_estate, ctx->d_tab});
		else if (ctx->model == PFX_MULTI_MODEL)
			rc = pfx_launch_project_psd<NLOC, BS>(ctx, nl_block<NLOC, NGLOC, NPTS, M_MULTI>{ctx->d_estate, ctx->d_tab});
		else
			rc = pfx_launch_project_psd<NLOC, BS>(ctx, nl_block<NLOC, NGLOC, NPTS, M_LINEAR>{ctx->d_estate, ctx->d_tab});
		if (rc)
			return rc;
		pfx_launch_assemble_rows<NLOC, BS>(ctx, staged_block<NLOC, BS>{ctx->d_kloc}, false);
		return PFX_OK;
	}
} // namespace

pfx_status pfx_launch_local_nl(pfx_ctx *ctx, const double *d_u, bool energy, bool grad, bool hess, double *d_eloc_out)
{
	double *const eloc = d_eloc_out ? d_eloc_out : ctx->d_eloc;
	if (ctx->model != PFX_NEOHOOKEAN && ctx->model != PFX_LINEAR_ELASTICITY && ctx->model != PFX_SAINT_VENANT && ctx->model != PFX_MULTI_MODEL)
		PFX_FAIL(PFX_ERR_UNSUPPORTED, "the nonlinear-assembler interface is implemented for NeoHookean, SaintVenant, LinearElasticity and MultiModel");
	if (ctx->model == PFX_MULTI_MODEL && !ctx->d_model_e)
		PFX_FAIL(PFX_ERR_ARG, "pfx_set_element_models must follow pfx_set_material");
	const int np = ctx->n_pts;
	bool ok = true;
	switch (ctx->kind)
	{
	case PFX_P1TET:
		if (np != 1)
			ok = false;
		else
			launch<4, 4, 1>(ctx, d_u, energy, grad, hess, eloc);
		break;
	case PFX_P2TET:
		if (np != 6)
			ok = false;
		else
			launch<10, 4, 6>(ctx, d_u, energy, grad, hess, eloc);
		break;
	case PFX_Q1HEX:
		if (np != 1)
			ok = false;
		else
			launch<8, 8, 1>(ctx, d_u, energy, grad, hess, eloc);
		break;
	case PFX_Q2HEX:
		if (np != 27)
			ok = false;
		else
			launch<27, 8, 27>(ctx, d_u, energy, grad, hess, eloc);
		break;
	case PFX_P1TRI:
		if (np != 1)
			ok = false;
		else
			launch<3, 3, 1>(ctx, d_u, energy, grad, hess, eloc);
		break;
	case PFX_P2TRI:
		if (np != 4)
			ok = false;
		else
			launch<6, 3, 4>(ctx, d_u, energy, grad, hess, eloc);
		break;
	default:
		ok = false;
	}
	if (!ok)
		PFX_FAIL(PFX_ERR_UNSUPPORTED, "nonlinear assembly: unsupported element kind / quadrature order combination");
	if (hess)
		ctx->estate_kind = PFX_ESTATE_NL;
	PFX_HIP(hipGetLastError());
	return PFX_OK;
}

// global Hessian from the state left by pfx_launch_local_nl(..., hess = true)
pfx_status pfx_launch_rows_nl(pfx_ctx *ctx)
{
	if (ctx->estate_kind != PFX_ESTATE_NL)
		PFX_FAIL(PFX_ERR_UNSUPPORTED, "Hessian assembly: element state missing");
	switch (ctx->kind)
	{
	case PFX_P1TET:
		launch_rows<4, 4, 1>(ctx);
		break;
	case PFX_P2TET:
		launch_rows<10, 4, 6>(ctx);
		break;
	case PFX_Q1HEX:
		launch_rows<8, 8, 1>(ctx);
		break;
	case PFX_Q2HEX:
		launch_rows<27, 8, 27>(ctx);
		break;
	case PFX_P1TRI:
		launch_rows<3, 3, 1>(ctx);
		break;
	case PFX_P2TRI:
		launch_rows<6, 3, 4>(ctx);
		break;
	default:
		PFX_FAIL(PFX_ERR_UNSUPPORTED, "Hessian assembly: unsupported element kind");
	}
	PFX_HIP(hipGetLastError());
	return PFX_OK;
}

// global Hessian with every element Hessian projected to PSD (Assembler.cpp:554-555), from the state left by
// pfx_launch_local_nl(..., hess = true)
pfx_status pfx_launch_rows_nl_psd(pfx_ctx *ctx)
{
	if (ctx->estate_kind != PFX_ESTATE_NL)
		PFX_FAIL(PFX_ERR_UNSUPPORTED, "Hessian assembly: element state missing");
	const int64_t need = ctx->n_el * (int64_t)(ctx->n_loc * (ctx->n_loc + 1) / 2) * (ctx->bs * ctx->bs);
	if (ctx->kloc_count < need)
	{
		if (pfx_alloc(ctx, &ctx->d_kloc, need))
			return PFX_ERR_HIP;
		ctx->kloc_count = need;
	}
	pfx_status rc;
	switch (ctx->kind)
	{
	case PFX_P1TET:
		rc = launch_psd<4, 4, 1>(ctx);
		break;
	case PFX_P2TET:
		rc = launch_psd<10, 4, 6>(ctx);
		break;
	case PFX_Q1HEX:
		rc = launch_psd<8, 8, 1>(ctx);
		break;
	case PFX_Q2HEX:
		rc = launch_psd<27, 8, 27>(ctx);
		break;
	case PFX_P1TRI:
		rc = launch_psd<3, 3, 1>(ctx);
		break;
	case PFX_P2TRI:
		rc = launch_psd<6, 3, 4>(ctx);
		break;
	default:
		PFX_FAIL(PFX_ERR_UNSUPPORTED, "Hessian assembly: unsupported element kind");
	}
	if (rc)
		return rc;
	PFX_HIP(hipGetLastError());
	return PFX_OK;
}
