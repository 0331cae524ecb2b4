// pfx_local_linear.hip -- local stiffness blocks of the LINEAR assemblers (Laplacian, LinearElasticity,
// HookeLinearElasticity).
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
// Two kernels:
//   k_element_geometry  one lane per (element, quadrature point): Jacobian of the geometric map -> J^{-T} and det,
//                       10 doubles per point (per element when the geometric map is affine, i.e. 4-node tets, where
//                       every point yields bitwise the same Jacobian);
//   k_assemble_rows     (pfx_rowasm.h) with the functor below: one lane per contribution computes the block of the
//                       pair (i, j), j <= i, by the quadrature loop: physical gradients grad * J^{-T} of the two
//                       bases, then Laplacian::assemble / LinearElasticity::assemble.
#include "pfx_internal.h"
#include "pfx_rowasm.h"

namespace
{
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

	// geo[(e*SP + k)*10 ..] = J^{-T} (9, row-major) and det J at point k; SP = 1 for affine maps, n_pts otherwise
	template <int NGLOC, int SP>
	__global__ __launch_bounds__(256) void k_element_geometry(const int32_t *__restrict__ gconn, const double *__restrict__ xyz,
															   const pfx_tables *__restrict__ tab, int64_t n_el, double *__restrict__ geo)
	{
		const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
		if (t >= n_el * SP)
			return;
		const int64_t e = t / SP;
		const int k = (int)(t - e * SP);
		double tmp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
		for (int j = 0; j < NGLOC; ++j)
		{
			const double *x = xyz + 3 * (int64_t)gconn[e * NGLOC + j];
			for (int r = 0; r < 3; ++r)
				for (int c = 0; c < 3; ++c)
					tmp[3 * r + c] += tab->ggrad[k][j][r] * x[c] * 1.0;
		}
		// Triangles (ElementAssemblyValues::finalize2d, ElementAssemblyValues.cpp:161-197): the third row and column of tmp
		// are exact zeros; with a one in the corner the 3 x 3 cofactor formulas below reduce, operation by operation, to
		// Eigen's 2 x 2 determinant m00 m11 - m10 m01 and inverse (adjugate times 1/det): every extra term is a product
		// with an exact zero or one
		if (NGLOC == 3)
			tmp[8] = 1.0;
		double out[10];
		inv3_transposed(tmp, out);
		out[9] = det3_fixed(tmp);
		for (int q = 0; q < 10; ++q)
			geo[t * 10 + q] = out[q];
	}

	// block of the pair (i, j), j <= i : out[m*BS+n] = (row component m of basis i, column component n of basis j),
	// values with |v| < 1e-30 replaced by an exact zero (they are absent from the reference's matrix)
	// PP: the material arrays hold one value per (element, point) instead of one per element
#ifndef PFX_LIN_UNROLL
#define PFX_LIN_UNROLL 1 // quadrature points per trip of the functors' point loops
#endif
	template <int NLOC, int NPTS, int BS, int SP, bool PP = false>
	struct linear_block
	{
		const double *geo;
		const pfx_tables *tab;
		double lambda, mu;                // uniform Lame parameters, or
		const double *lambda_e, *mu_e;    // one value per element or per (element, point) (LameParameters::lambda_mu, utils/ElasticityUtils.cpp:490-517)
		static constexpr int LDS_DOUBLES = NPTS * NLOC * 3 + NPTS; // grad[k][i][3], then w[k]

		__device__ __forceinline__ void load_tables(double *lds, int lane) const
		{
			for (int t = lane; t < NPTS * NLOC * 3; t += 64)
				lds[t] = tab->grad[t / (NLOC * 3)][(t / 3) % NLOC][t % 3];
			for (int t = lane; t < NPTS; t += 64)
				lds[NPTS * NLOC * 3 + t] = tab->w[t];
		}

		__device__ __forceinline__ void operator()(const double *lds, uint32_t e, int i, int j, double (&out)[BS * BS]) const
		{
			typedef double d2 __attribute__((ext_vector_type(2)));
			const d2 *ge = reinterpret_cast<const d2 *>(geo + (int64_t)e * (SP * 10));
			double jt[10];
			if (SP == 1)
				for (int q = 0; q < 5; ++q)
				{
					const d2 v = ge[q];
					jt[2 * q] = v.x, jt[2 * q + 1] = v.y;
				}
			double lambda = lambda_e ? lambda_e[PP ? (int64_t)e * NPTS : e] : this->lambda, mu = mu_e ? mu_e[PP ? (int64_t)e * NPTS : e] : this->mu;
			double res[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll(PFX_LIN_UNROLL)
			for (int k = 0; k < NPTS; ++k)
			{
				if (SP != 1)
					for (int q = 0; q < 5; ++q)
					{
						const d2 v = ge[k * 5 + q];
						jt[2 * q] = v.x, jt[2 * q + 1] = v.y;
					}
				const double dak = jt[9] * lds[NPTS * NLOC * 3 + k]; // det * weight
				if (PP && k > 0)
					lambda = lambda_e[(int64_t)e * NPTS + k], mu = mu_e[(int64_t)e * NPTS + k];
				const double *ri = lds + (k * NLOC + i) * 3, *rj = lds + (k * NLOC + j) * 3;
				double gi[3], gj[3];
				for (int c = 0; c < 3; ++c)
				{
					gi[c] = (ri[0] * jt[c] + ri[1] * jt[3 + c]) + ri[2] * jt[6 + c];
					gj[c] = (rj[0] * jt[c] + rj[1] * jt[3 + c]) + rj[2] * jt[6 + c];
				}
				const double dot = (gi[0] * gj[0] + gi[1] * gj[1]) + gi[2] * gj[2];
				if (BS == 1)
					res[0] += dot * dak;
				else
				{
					double outer[9];
					for (int col = 0; col < 3; ++col)
						for (int row = 0; row < 3; ++row)
							outer[col * 3 + row] = gi[row] * gj[col];
					for (int ii = 0; ii < 3; ++ii)
						for (int jj = 0; jj < 3; ++jj)
						{
							double rk = outer[ii * 3 + jj] * mu + outer[jj * 3 + ii] * lambda;
							if (ii == jj)
								rk += mu * dot;
							res[jj * 3 + ii] += rk * dak;
						}
				}
			}
			if (BS == 1)
				out[0] = fabs(res[0]) < 1e-30 ? 0.0 : res[0];
			else
				// entry (row comp m, col comp n) of block (i,j) is res[n*3+m]  (Assembler.cpp:169,177,182)
				for (int m = 0; m < BS; ++m)
					for (int n = 0; n < BS; ++n)
					{
						const double v = res[n * 3 + m];
						out[m * BS + n] = fabs(v) < 1e-30 ? 0.0 : v;
					}
		}
	};

	// HookeLinearElasticity::assemble (assembler/HookeLinearElasticity.cpp:75-190, 3-D branch) for the pair (i, j): per
	// point, strain<3>() puts the physical gradient into row `coo` of a zero matrix and symmetrises it, the Voigt strain
	// (engineering shears) goes through ElasticityTensor::compute_stress (utils/ElasticityUtils.cpp:469-477), and entry
	// (coo_j * 3 + coo_i) is trace(sigma_i * eps_j).  Every operation of the reference is kept (including the products
	// with the structural zeros), in its order.
	// BS = 2 (triangles, HookeLinearElasticity.cpp:92-122): the tensor arrives with the 3 x 3 one of the 2-D problem in its
	// xx, yy, xy rows and columns and zeros elsewhere, the gradients have a zero third component; every extra term of the
	// 3-D expressions is then an exact zero added to the 2-D sums, which are formed in the reference's order
	template <int NLOC, int NPTS, int SP, int BS = 3>
	struct hooke_block
	{
		const double *geo;
		const pfx_tables *tab;
		const double *tensor; // device, 36 doubles, symmetric 6 x 6
		static constexpr int LDS_DOUBLES = NPTS * NLOC * 3 + NPTS + 36; // grad[k][i][3], w[k], tensor
		static constexpr int MIN_WAVES = 2; // the 6 x 6 tensor product of one point wants more registers than three waves leave

		__device__ __forceinline__ void load_tables(double *lds, int lane) const
		{
			for (int t = lane; t < NPTS * NLOC * 3; t += 64)
				lds[t] = tab->grad[t / (NLOC * 3)][(t / 3) % NLOC][t % 3];
			for (int t = lane; t < NPTS; t += 64)
				lds[NPTS * NLOC * 3 + t] = tab->w[t];
			if (lane < 36)
				lds[NPTS * NLOC * 3 + NPTS + lane] = tensor[lane];
		}

		__device__ __forceinline__ static void strain(const double *g, int coo, double *eps)
		{
			double J[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
			for (int c = 0; c < 3; ++c)
				J[3 * coo + c] = g[c];
			for (int r = 0; r < 3; ++r)
				for (int c = 0; c < 3; ++c)
					eps[3 * r + c] = (J[3 * r + c] + J[3 * c + r]) * 0.5;
		}

		__device__ __forceinline__ void operator()(const double *lds, uint32_t e, int i, int j, double (&out)[BS * BS]) const
		{
			const double *C = lds + NPTS * NLOC * 3 + NPTS;
			const double *ge = geo + (int64_t)e * (SP * 10);
			double res[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll(PFX_LIN_UNROLL)
			for (int k = 0; k < NPTS; ++k)
			{
				const double *jt = ge + (SP == 1 ? 0 : k * 10);
				const double dak = jt[9] * lds[NPTS * NLOC * 3 + k];
				const double *ri = lds + (k * NLOC + i) * 3, *rj = lds + (k * NLOC + j) * 3;
				double gi[3], gj[3];
				for (int c = 0; c < 3; ++c)
				{
					gi[c] = (ri[0] * jt[c] + ri[1] * jt[3 + c]) + ri[2] * jt[6 + c];
					gj[c] = (rj[0] * jt[c] + rj[1] * jt[3 + c]) + rj[2] * jt[6 + c];
				}
				double sigma[3][9], epsj[3][9];
#pragma unroll
				for (int coo = 0; coo < 3; ++coo)
				{
					double ei[9];
					strain(gi, coo, ei);
					strain(gj, coo, epsj[coo]);
					const double e6[6] = {ei[0], ei[4], ei[8], 2 * ei[5], 2 * ei[2], 2 * ei[1]};
					double s6[6];
#pragma unroll
					for (int a = 0; a < 6; ++a)
					{
						double r = 0;
#pragma unroll
						for (int b = 0; b < 6; ++b)
							r += C[a * 6 + b] * e6[b];
						s6[a] = r;
					}
					double *sg = sigma[coo];
					sg[0] = s6[0], sg[1] = s6[5], sg[2] = s6[4];
					sg[3] = s6[5], sg[4] = s6[1], sg[5] = s6[3];
					sg[6] = s6[4], sg[7] = s6[3], sg[8] = s6[2];
				}
#pragma unroll
				for (int ci = 0; ci < 3; ++ci)
#pragma unroll
					for (int cj = 0; cj < 3; ++cj)
					{
						const double *sg = sigma[ci], *ej = epsj[cj];
						const double d0 = (sg[0] * ej[0] + sg[1] * ej[3]) + sg[2] * ej[6];
						const double d1 = (sg[3] * ej[1] + sg[4] * ej[4]) + sg[5] * ej[7];
						const double d2 = (sg[6] * ej[2] + sg[7] * ej[5]) + sg[8] * ej[8];
						res[cj * 3 + ci] += ((d0 + d1) + d2) * dak;
					}
			}
			// entry (row comp m, col comp n) of block (i,j) is res[n*3+m]  (Assembler.cpp:169,177,182)
			for (int m = 0; m < BS; ++m)
				for (int n = 0; n < BS; ++n)
				{
					const double v = res[n * 3 + m];
					out[m * BS + n] = fabs(v) < 1e-30 ? 0.0 : v;
				}
		}
	};

	// Mass::assemble (assembler/Mass.cpp:5-20) for the pair (i, j): rho * phi_i * phi_j integrated with the mass
	// quadrature, on the diagonal of the bs x bs block; |v| < 1e-30 replaced by an exact zero as for the stiffness
	template <int NLOC, int NPTS, int BS, int SP>
	struct mass_block
	{
		const double *geo;
		const pfx_tables *tab; // tables of the mass quadrature rule
		double density;
		const double *density_e;
		const double *density_pts; // one density per (element, point): Density::operator() (utils/ElasticityUtils.cpp:561-570)
		static constexpr int LDS_DOUBLES = NPTS * NLOC + NPTS; // val[k][i], then w[k]

		__device__ __forceinline__ void load_tables(double *lds, int lane) const
		{
			for (int t = lane; t < NPTS * NLOC; t += 64)
				lds[t] = tab->val[t / NLOC][t % NLOC];
			for (int t = lane; t < NPTS; t += 64)
				lds[NPTS * NLOC + t] = tab->w[t];
		}

		__device__ __forceinline__ void operator()(const double *lds, uint32_t e, int i, int j, double (&out)[BS * BS]) const
		{
			const double *ge = geo + (int64_t)e * (SP * 10);
			const double rho = density_e ? density_e[e] : density;
			double det = ge[9];
			double tmp = 0;
#pragma unroll(PFX_LIN_UNROLL)
			for (int k = 0; k < NPTS; ++k)
			{
				if (SP != 1)
					det = ge[k * 10 + 9];
				const double dak = det * lds[NPTS * NLOC + k];
				const double rho_k = density_pts ? density_pts[(int64_t)e * NPTS + k] : rho;
				tmp += rho_k * lds[k * NLOC + i] * lds[k * NLOC + j] * dak;
			}
			if (fabs(tmp) < 1e-30)
				tmp = 0.0;
			for (int m = 0; m < BS; ++m)
				for (int n = 0; n < BS; ++n)
					out[m * BS + n] = (m == n) ? tmp : 0.0;
		}
	};

	// RhsAssembler::assemble (assembler/RhsAssembler.cpp:55-98) for a constant force density: one lane per
	// (element, local basis); gloc[e][i][d] = sum_q (f_d * (det w rho)) * phi_i(q)
	template <int NLOC, int NPTS, int BS, int SP>
	__global__ __launch_bounds__(256) void k_local_rhs(const double *__restrict__ geo, const pfx_tables *__restrict__ tab, int64_t n_el,
													   double f0, double f1, double f2, const double *__restrict__ force_pts, double density,
													   const double *__restrict__ density_e, const double *__restrict__ density_pts, double *__restrict__ gloc)
	{
		const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
		if (t >= n_el * NLOC)
			return;
		const int64_t e = t / NLOC;
		const int i = (int)(t - e * NLOC);
		const double *ge = geo + e * (SP * 10);
		const double rho = density_e ? density_e[e] : density;
		const double f[3] = {f0, f1, f2};
		double acc[3] = {0, 0, 0};
		for (int q = 0; q < NPTS; ++q)
		{
			const double det = ge[(SP == 1 ? 0 : q) * 10 + 9];
			const double s = det * tab->w[q] * (density_pts ? density_pts[e * NPTS + q] : rho);
			const double v = tab->val[q][i];
			// force density: one constant vector, or one value per (element, point) (what problem.rhs() returns in the reference)
			const double *fq = force_pts ? force_pts + (e * NPTS + q) * BS : f;
			for (int d = 0; d < BS; ++d)
				acc[d] += (fq[d] * s) * v;
		}
		for (int d = 0; d < BS; ++d)
			gloc[t * BS + d] = acc[d];
	}

	// physical position of every quadrature point: sum_j N_j^geo(xi_q) x_j  (ElementAssemblyValues.cpp:328-345)
	template <int NGLOC, int NPTS>
	__global__ __launch_bounds__(256) void k_quadrature_points(const int32_t *__restrict__ gconn, const double *__restrict__ xyz,
															   const pfx_tables *__restrict__ tab, int64_t n_el, double *__restrict__ pts)
	{
		const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
		if (t >= n_el * NPTS)
			return;
		const int64_t e = t / NPTS;
		const int q = (int)(t - e * NPTS);
		double p[3] = {0, 0, 0};
		for (int j = 0; j < NGLOC; ++j)
		{
			const double nv = tab->gval[q][j];
			const double *node = xyz + 3 * (int64_t)gconn[e * NGLOC + j];
			for (int c = 0; c < 3; ++c)
				p[c] += nv * node[c];
		}
		for (int c = 0; c < 3; ++c)
			pts[t * 3 + c] = p[c];
	}

	template <int NGLOC, int NPTS>
	void launch_geometry(pfx_ctx *ctx, const pfx_tables *d_tab)
	{
		constexpr int SP = (NGLOC == 4 || NGLOC == 3) ? 1 : NPTS; // affine geometry: one Jacobian per element
		const int64_t n = ctx->n_el * SP;
		hipLaunchKernelGGL((k_element_geometry<NGLOC, SP>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_gconn, ctx->d_xyz,
						   d_tab, ctx->n_el, ctx->d_estate);
	}

	template <int NLOC, int NGLOC, int NPTS>
	void launch_mass_rows(pfx_ctx *ctx, double density, const double *d_density_e, const double *d_density_pts)
	{
		constexpr int SP = (NGLOC == 4 || NGLOC == 3) ? 1 : NPTS; // affine geometry: one Jacobian per element
		constexpr int DIM = (NGLOC == 3) ? 2 : 3;
		launch_geometry<NGLOC, NPTS>(ctx, ctx->d_tab_mass);
		if (ctx->bs == 1)
			pfx_launch_assemble_rows<NLOC, 1>(ctx, mass_block<NLOC, NPTS, 1, SP>{ctx->d_estate, ctx->d_tab_mass, density, d_density_e, d_density_pts}, true);
		else
			pfx_launch_assemble_rows<NLOC, DIM>(ctx, mass_block<NLOC, NPTS, DIM, SP>{ctx->d_estate, ctx->d_tab_mass, density, d_density_e, d_density_pts}, true);
	}

	template <int NLOC, int NGLOC, int NPTS>
	void launch_rhs(pfx_ctx *ctx, const double *f, const double *d_force_pts, double density, const double *d_density_e, const double *d_density_pts)
	{
		constexpr int SP = (NGLOC == 4 || NGLOC == 3) ? 1 : NPTS; // affine geometry: one Jacobian per element
		constexpr int DIM = (NGLOC == 3) ? 2 : 3;
		launch_geometry<NGLOC, NPTS>(ctx, ctx->d_tab);
		const int64_t n = ctx->n_el * NLOC;
		if (ctx->bs == 1)
			hipLaunchKernelGGL((k_local_rhs<NLOC, NPTS, 1, SP>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_estate, ctx->d_tab,
							   ctx->n_el, f ? f[0] : 0.0, 0.0, 0.0, d_force_pts, density, d_density_e, d_density_pts, ctx->d_gloc);
		else
			hipLaunchKernelGGL((k_local_rhs<NLOC, NPTS, DIM, SP>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_estate, ctx->d_tab,
							   ctx->n_el, f ? f[0] : 0.0, f ? f[1] : 0.0, (f && DIM == 3) ? f[2] : 0.0, d_force_pts, density, d_density_e, d_density_pts, ctx->d_gloc);
	}

	template <int NLOC, int NGLOC, int NPTS>
	void launch_rows(pfx_ctx *ctx, bool write_mask)
	{
		constexpr int SP = (NGLOC == 4 || NGLOC == 3) ? 1 : NPTS; // affine geometry: one Jacobian per element
		constexpr int DIM = (NGLOC == 3) ? 2 : 3;
		if (ctx->model == PFX_HOOKE_LINEAR)
		{
			pfx_launch_assemble_rows<NLOC, DIM>(ctx, hooke_block<NLOC, NPTS, SP, DIM>{ctx->d_estate, ctx->d_tab, ctx->d_tensor}, write_mask);
		}
		else if (ctx->bs == 1)
			pfx_launch_assemble_rows<NLOC, 1>(ctx, linear_block<NLOC, NPTS, 1, SP>{ctx->d_estate, ctx->d_tab, ctx->lambda, ctx->mu, ctx->d_lambda_e, ctx->d_mu_e}, write_mask);
		else if (ctx->mat_per_point && ctx->d_lambda_e)
			pfx_launch_assemble_rows<NLOC, DIM>(ctx, linear_block<NLOC, NPTS, DIM, SP, true>{ctx->d_estate, ctx->d_tab, ctx->lambda, ctx->mu, ctx->d_lambda_e, ctx->d_mu_e}, write_mask);
		else
			pfx_launch_assemble_rows<NLOC, DIM>(ctx, linear_block<NLOC, NPTS, DIM, SP>{ctx->d_estate, ctx->d_tab, ctx->lambda, ctx->mu, ctx->d_lambda_e, ctx->d_mu_e}, write_mask);
	}

	bool supported(const pfx_ctx *ctx)
	{
		switch (ctx->kind)
		{
		case PFX_P1TET:
		case PFX_Q1HEX:
			return ctx->n_pts == 1;
		case PFX_P2TET:
			return ctx->n_pts == 6;
		case PFX_Q2HEX:
			return ctx->n_pts == 27;
		case PFX_P1TRI:
			return ctx->n_pts == 1;
		case PFX_P2TRI:
			return ctx->n_pts == 4;
		}
		return false;
	}
} // namespace

// per-element geometry for the linear assemblers
pfx_status pfx_launch_local_linear(pfx_ctx *ctx)
{
	if (!supported(ctx))
		PFX_FAIL(PFX_ERR_UNSUPPORTED, "linear assembly: unsupported element kind / quadrature order combination");
	switch (ctx->kind)
	{
	case PFX_P1TET:
		launch_geometry<4, 1>(ctx, ctx->d_tab);
		break;
	case PFX_P2TET:
		launch_geometry<4, 6>(ctx, ctx->d_tab);
		break;
	case PFX_Q1HEX:
		launch_geometry<8, 1>(ctx, ctx->d_tab);
		break;
	case PFX_P1TRI:
		launch_geometry<3, 1>(ctx, ctx->d_tab);
		break;
	case PFX_P2TRI:
		launch_geometry<3, 4>(ctx, ctx->d_tab);
		break;
	default:
		launch_geometry<8, 27>(ctx, ctx->d_tab);
	}
	ctx->estate_kind = PFX_ESTATE_LINEAR;
	PFX_HIP(hipGetLastError());
	return PFX_OK;
}

// global matrix from the state left by pfx_launch_local_linear
pfx_status pfx_launch_rows_linear(pfx_ctx *ctx, bool write_mask)
{
	if (!supported(ctx) || ctx->estate_kind != PFX_ESTATE_LINEAR)
		PFX_FAIL(PFX_ERR_UNSUPPORTED, "linear assembly: element state missing or unsupported element kind");
	switch (ctx->kind)
	{
	case PFX_P1TET:
		launch_rows<4, 4, 1>(ctx, write_mask);
		break;
	case PFX_P2TET:
		launch_rows<10, 4, 6>(ctx, write_mask);
		break;
	case PFX_Q1HEX:
		launch_rows<8, 8, 1>(ctx, write_mask);
		break;
	case PFX_P1TRI:
		launch_rows<3, 3, 1>(ctx, write_mask);
		break;
	case PFX_P2TRI:
		launch_rows<6, 3, 4>(ctx, write_mask);
		break;
	default:
		launch_rows<27, 8, 27>(ctx, write_mask);
	}
	PFX_HIP(hipGetLastError());
	return PFX_OK;
}

// mass matrix into the value array (Assembler<Mass>::assemble with is_mass = true, Assembler.cpp:135-139): geometry at
// the points of the mass quadrature rule, then the row assembly with mass_block
pfx_status pfx_launch_mass(pfx_ctx *ctx, double density, const double *d_density_e, const double *d_density_pts)
{
	static const int expect[6] = {6, 14, 27, 125, 4, 7}; // mass rules: tet order 3 / 5, hex order 3 / 5, triangle order 3 / 5
	if (ctx->kind < 0 || ctx->kind > 5 || ctx->h_tab_mass.n_pts != expect[ctx->kind])
		PFX_FAIL(PFX_ERR_UNSUPPORTED, "mass assembly: unsupported element kind");
	switch (ctx->kind)
	{
	case PFX_P1TET:
		launch_mass_rows<4, 4, 6>(ctx, density, d_density_e, d_density_pts);
		break;
	case PFX_P2TET:
		launch_mass_rows<10, 4, 14>(ctx, density, d_density_e, d_density_pts);
		break;
	case PFX_Q1HEX:
		launch_mass_rows<8, 8, 27>(ctx, density, d_density_e, d_density_pts);
		break;
	case PFX_P1TRI:
		launch_mass_rows<3, 3, 4>(ctx, density, d_density_e, d_density_pts);
		break;
	case PFX_P2TRI:
		launch_mass_rows<6, 3, 7>(ctx, density, d_density_e, d_density_pts);
		break;
	default:
		launch_mass_rows<27, 8, 125>(ctx, density, d_density_e, d_density_pts);
	}
	ctx->estate_kind = PFX_ESTATE_NONE;
	PFX_HIP(hipGetLastError());
	return PFX_OK;
}

// per-element body-force vectors into ctx->d_gloc (stiffness quadrature rule)
pfx_status pfx_launch_local_rhs(pfx_ctx *ctx, const double *force, const double *d_force_pts, double density, const double *d_density_e, const double *d_density_pts)
{
	if (!supported(ctx))
		PFX_FAIL(PFX_ERR_UNSUPPORTED, "rhs assembly: unsupported element kind / quadrature order combination");
	switch (ctx->kind)
	{
	case PFX_P1TET:
		launch_rhs<4, 4, 1>(ctx, force, d_force_pts, density, d_density_e, d_density_pts);
		break;
	case PFX_P2TET:
		launch_rhs<10, 4, 6>(ctx, force, d_force_pts, density, d_density_e, d_density_pts);
		break;
	case PFX_Q1HEX:
		launch_rhs<8, 8, 1>(ctx, force, d_force_pts, density, d_density_e, d_density_pts);
		break;
	case PFX_P1TRI:
		launch_rhs<3, 3, 1>(ctx, force, d_force_pts, density, d_density_e, d_density_pts);
		break;
	case PFX_P2TRI:
		launch_rhs<6, 3, 4>(ctx, force, d_force_pts, density, d_density_e, d_density_pts);
		break;
	default:
		launch_rhs<27, 8, 27>(ctx, force, d_force_pts, density, d_density_e, d_density_pts);
	}
	ctx->estate_kind = PFX_ESTATE_NONE;
	PFX_HIP(hipGetLastError());
	return PFX_OK;
}

// positions of the points of the mass quadrature rule (n_el x n_pts_mass x 3)
pfx_status pfx_launch_mass_quadrature_points(pfx_ctx *ctx, double *d_pts)
{
	static const int expect[6] = {6, 14, 27, 125, 4, 7};
	if (ctx->kind < 0 || ctx->kind > 5 || ctx->h_tab_mass.n_pts != expect[ctx->kind])
		PFX_FAIL(PFX_ERR_UNSUPPORTED, "mass quadrature points: unsupported element kind");
	const int64_t n = ctx->n_el * ctx->h_tab_mass.n_pts;
	const unsigned grid = (unsigned)((n + 255) / 256 > 0 ? (n + 255) / 256 : 1);
	switch (ctx->kind)
	{
	case PFX_P1TET:
		hipLaunchKernelGGL((k_quadrature_points<4, 6>), dim3(grid), dim3(256), 0, ctx->stream, ctx->d_gconn, ctx->d_xyz, ctx->d_tab_mass, ctx->n_el, d_pts);
		break;
	case PFX_P2TET:
		hipLaunchKernelGGL((k_quadrature_points<4, 14>), dim3(grid), dim3(256), 0, ctx->stream, ctx->d_gconn, ctx->d_xyz, ctx->d_tab_mass, ctx->n_el, d_pts);
		break;
	case PFX_Q1HEX:
		hipLaunchKernelGGL((k_quadrature_points<8, 27>), dim3(grid), dim3(256), 0, ctx->stream, ctx->d_gconn, ctx->d_xyz, ctx->d_tab_mass, ctx->n_el, d_pts);
		break;
	case PFX_P1TRI:
		hipLaunchKernelGGL((k_quadrature_points<3, 4>), dim3(grid), dim3(256), 0, ctx->stream, ctx->d_gconn, ctx->d_xyz, ctx->d_tab_mass, ctx->n_el, d_pts);
		break;
	case PFX_P2TRI:
		hipLaunchKernelGGL((k_quadrature_points<3, 7>), dim3(grid), dim3(256), 0, ctx->stream, ctx->d_gconn, ctx->d_xyz, ctx->d_tab_mass, ctx->n_el, d_pts);
		break;
	default:
		hipLaunchKernelGGL((k_quadrature_points<8, 125>), dim3(grid), dim3(256), 0, ctx->stream, ctx->d_gconn, ctx->d_xyz, ctx->d_tab_mass, ctx->n_el, d_pts);
	}
	PFX_HIP(hipGetLastError());
	return PFX_OK;
}

pfx_status pfx_launch_quadrature_points(pfx_ctx *ctx, double *d_pts)
{
	if (!supported(ctx))
		PFX_FAIL(PFX_ERR_UNSUPPORTED, "quadrature points: unsupported element kind / quadrature order combination");
	const int64_t n = ctx->n_el * ctx->n_pts;
	const unsigned grid = (unsigned)((n + 255) / 256 > 0 ? (n + 255) / 256 : 1);
	switch (ctx->kind)
	{
	case PFX_P1TET:
		hipLaunchKernelGGL((k_quadrature_points<4, 1>), dim3(grid), dim3(256), 0, ctx->stream, ctx->d_gconn, ctx->d_xyz, ctx->d_tab, ctx->n_el, d_pts);
		break;
	case PFX_P2TET:
		hipLaunchKernelGGL((k_quadrature_points<4, 6>), dim3(grid), dim3(256), 0, ctx->stream, ctx->d_gconn, ctx->d_xyz, ctx->d_tab, ctx->n_el, d_pts);
		break;
	case PFX_Q1HEX:
		hipLaunchKernelGGL((k_quadrature_points<8, 1>), dim3(grid), dim3(256), 0, ctx->stream, ctx->d_gconn, ctx->d_xyz, ctx->d_tab, ctx->n_el, d_pts);
		break;
	case PFX_P1TRI:
		hipLaunchKernelGGL((k_quadrature_points<3, 1>), dim3(grid), dim3(256), 0, ctx->stream, ctx->d_gconn, ctx->d_xyz, ctx->d_tab, ctx->n_el, d_pts);
		break;
	case PFX_P2TRI:
		hipLaunchKernelGGL((k_quadrature_points<3, 4>), dim3(grid), dim3(256), 0, ctx->stream, ctx->d_gconn, ctx->d_xyz, ctx->d_tab, ctx->n_el, d_pts);
		break;
	default:
		hipLaunchKernelGGL((k_quadrature_points<8, 27>), dim3(grid), dim3(256), 0, ctx->stream, ctx->d_gconn, ctx->d_xyz, ctx->d_tab, ctx->n_el, d_pts);
	}
	PFX_HIP(hipGetLastError());
	return PFX_OK;
}
