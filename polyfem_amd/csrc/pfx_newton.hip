// pfx_newton.hip -- Newton's method with line search, iterate resident on the device.
//
// Reference: cppoptlib::NonlinearSolver::minimize (src/polyfem/solver/NonlinearSolver.tpp:33-275),
// SparseNewtonDescentSolver (src/polyfem/solver/SparseNewtonDescentSolver.tpp:40-47 increase_descent_strategy,
// :63-100 compute_update_direction, :105-123 assemble_hessian, :128-155 solve_linear_system, :160-193
// check_direction; constants SparseNewtonDescentSolver.hpp:39-42), BacktrackingLineSearch
// (src/polyfem/solver/line_search/BacktrackingLineSearch.hpp:29-147, :157-255; LineSearch.tpp:90-145) and the
// objective NLProblem (src/polyfem/solver/NLProblem.cpp:79-99) = ElasticForm + BodyForm (value -x.rhs).
// cppoptlib's checkConvergence (meta.h) is not part of the reference tree; it is restated from the published
// library: iteration limit, then f delta, then gradient norm.
//
// Transient steps: with an inertia term enabled (pfx_inertia_enable) the objective is
// w (E(x) - rhs.x) + 1/2 (x - x_tilde)^T M (x - x_tilde), i.e. InertiaForm (src/polyfem/solver/forms/InertiaForm.cpp:14-31)
// beside ElasticForm and BodyForm weighted with the integrator's acceleration_scaling
// (src/polyfem/state/StateSolveNonlinear.cpp:78-84); pfx_implicit_euler_run adds the ImplicitEuler bookkeeping
// (src/polyfem/time_integrator/ImplicitEuler.cpp:5-33).
//
// The reference keeps x, grad and delta_x in host Eigen vectors and crosses to the GPU for every energy evaluation
// (its USE_NONLINEAR_GPU path copies x and delta_x per line-search step, BacktrackingLineSearch.cu).  Here the three
// vectors live on the device for the whole solve; the host only sees the scalars that steer the iteration.
//
// Over a partition (pfx_dist_newton_run: one context per rank holding its part, local numbering [owned | halo]) the same
// control flow runs on every rank with every decision taken on all-reduced scalars, so all ranks branch alike: sums over
// the owned dofs / owned elements + one all-reduce per evaluation, the linear solve is the partitioned CG, which hands
// the direction back with a current halo, so that every trial point x + a dx is halo-consistent without further exchanges.
#include "pfx_dist.h"
#include "pfx_internal.h"

#include <cmath>

namespace
{
	constexpr int NB = 512; // blocks of the reduction kernels

	// g = gradient of the objective restricted to the free dofs: (g - rhs), 0 on constrained dofs
	__global__ __launch_bounds__(256) void k_objective_gradient(double *__restrict__ g, const double *__restrict__ rhs,
																const uint8_t *__restrict__ fixed, int64_t n, double w,
																const double *__restrict__ md, double alw, const double *__restrict__ alm,
																const double *__restrict__ alt, const double *__restrict__ x)
	{
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
		{
			double v = g[k];
			if (rhs)
				v -= rhs[k];
			if (md) // transient step: weighted elastic + body forms, plus M (x - x_tilde)
				v = w * v + md[k];
			if (alm) // augmented-Lagrangian penalty on the Dirichlet dofs: weight * m (x - target)
				v += alw * alm[k] * (x[k] - alt[k]);
			if (fixed && fixed[k])
				v = 0.0;
			g[k] = v;
		}
	}

	// mass[b] = rho phi_i phi_j of node block b, read from the (0,0) entry of the assembled mass matrix
	__global__ __launch_bounds__(256) void k_capture_mass(const double *__restrict__ values, int64_t n_blocks, int bs, double *__restrict__ mass)
	{
		for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_blocks; b += (int64_t)gridDim.x * blockDim.x)
			mass[b] = values[b * (bs * bs)];
	}

	// md = M v on the rows [0, n_rows), M = mass (x) I_bs (bs = 3, or 2 for triangle meshes).  Eight lanes share a node row: they walk the row's blocks side by
	// side (mass and column index are read in 64-byte runs) and combine their partial sums with shuffles.
	__global__ __launch_bounds__(256) void k_mass_apply(const int64_t *__restrict__ node_ptr, const int32_t *__restrict__ node_col,
														const double *__restrict__ mass, const double *__restrict__ v, int64_t n_rows, int bs,
														double *__restrict__ md)
	{
		const int sub = threadIdx.x & 7;
		const int64_t stride = (int64_t)gridDim.x * (blockDim.x >> 3);
		for (int64_t g0 = (int64_t)blockIdx.x * (blockDim.x >> 3); g0 < n_rows; g0 += stride) // uniform trip count per wave
		{
			const int64_t g = g0 + (threadIdx.x >> 3);
			double a0 = 0, a1 = 0, a2 = 0;
			if (g < n_rows)
			{
				const int64_t b1 = node_ptr[g + 1];
				for (int64_t b = node_ptr[g] + sub; b < b1; b += 8)
				{
					const int64_t c = bs * (int64_t)node_col[b];
					const double m = mass[b];
					a0 += m * v[c];
					a1 += m * v[c + 1];
					if (bs == 3)
						a2 += m * v[c + 2];
				}
			}
			for (int o = 4; o > 0; o >>= 1)
			{
				a0 += __shfl_xor(a0, o, 64);
				a1 += __shfl_xor(a1, o, 64);
				a2 += __shfl_xor(a2, o, 64);
			}
			if (g < n_rows && sub < bs)
				md[bs * g + sub] = sub == 0 ? a0 : (sub == 1 ? a1 : a2);
		}
	}

	// q = sqrt(m) (x - target): the penalty is weight/2 |q|^2 (ALForm::value_unweighted, solver/forms/ALForm.cpp:50-63)
	__global__ __launch_bounds__(256) void k_al_residual(const double *__restrict__ x, const double *__restrict__ m, const double *__restrict__ t,
														 int64_t n, double *__restrict__ q)
	{
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
			q[k] = sqrt(m[k]) * (x[k] - t[k]);
	}

	// Hessian of the penalty: weight * m on the diagonal (ALForm::second_derivative_unweighted, :70-73); one thread per node
	__global__ __launch_bounds__(256) void k_al_add_diag(const int64_t *__restrict__ node_ptr, const int32_t *__restrict__ node_col, int64_t n_nodes, int bs,
														 double alw, const double *__restrict__ m, double *__restrict__ values)
	{
		for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_nodes; g += (int64_t)gridDim.x * blockDim.x)
		{
			const double m0 = m[bs * g], m1 = m[bs * g + 1], m2 = bs == 3 ? m[bs * g + 2] : 0.0;
			if (m0 == 0.0 && m1 == 0.0 && m2 == 0.0)
				continue;
			int64_t lo = node_ptr[g], hi = node_ptr[g + 1];
			while (lo < hi)
			{
				const int64_t mid = (lo + hi) >> 1;
				if (node_col[mid] < g)
					lo = mid + 1;
				else
					hi = mid;
			}
			if (lo < node_ptr[g + 1] && node_col[lo] == g)
			{
				const int64_t at = lo * (bs * bs);
				values[at] += alw * m0;
				values[at + bs + 1] += alw * m1;
				if (bs == 3)
					values[at + 8] += alw * m2;
			}
		}
	}

	// utils::lump_matrix of the mass matrix: row sums on the diagonal; every component of a node gets the node's sum
	__global__ __launch_bounds__(256) void k_lump_mass(const int64_t *__restrict__ node_ptr, const double *__restrict__ mass, int64_t n_nodes, int bs,
													   double *__restrict__ out)
	{
		for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_nodes; g += (int64_t)gridDim.x * blockDim.x)
		{
			double s = 0.0;
			for (int64_t b = node_ptr[g]; b < node_ptr[g + 1]; ++b)
				s += mass[b];
			for (int m = 0; m < bs; ++m)
				out[bs * g + m] = s;
		}
	}

	// Hessian of the transient objective: values = w * values + M (M sits on the diagonal of every node block)
	__global__ __launch_bounds__(256) void k_add_inertia(double *__restrict__ values, int64_t nnz, int bs, double w, const double *__restrict__ mass)
	{
		const int b2 = bs * bs;
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x)
		{
			const int64_t b = k / b2;
			const int r = (int)(k - b * b2);
			double v = w * values[k];
			if (r % (bs + 1) == 0) // diagonal entries of the block
				v += mass[b];
			values[k] = v;
		}
	}

	// ImplicitEuler::update_quantities: v = (x - x_prev) / dt, a = (v - v_prev) / dt
	__global__ __launch_bounds__(256) void k_euler_update(const double *__restrict__ x, const double *__restrict__ x_prev, double dt, int64_t n,
														  double *__restrict__ v, double *__restrict__ a)
	{
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
		{
			const double vn = (x[k] - x_prev[k]) / dt;
			if (a)
				a[k] = (vn - v[k]) / dt;
			v[k] = vn;
		}
	}

	// out = x + a * d (out may be x)
	__global__ __launch_bounds__(256) void k_step(const double *x, const double *__restrict__ d, double a, int64_t n, double *out)
	{
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
			out[k] = x[k] + a * d[k];
	}

	// out = -in
	__global__ __launch_bounds__(256) void k_negate(const double *__restrict__ in, int64_t n, double *__restrict__ out)
	{
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
			out[k] = -in[k];
	}

	// partial[blockIdx] = sum a.b ; partial[NB + blockIdx] = sum (a+c).(a+c) ; partial[2NB + blockIdx] = max |a|
	// (c may be null).  Fixed order inside a block and over the blocks: results are reproducible.
	__global__ __launch_bounds__(256) void k_reduce3(const double *__restrict__ a, const double *__restrict__ b, const double *__restrict__ c,
													 int64_t n, double *__restrict__ partial)
	{
		__shared__ double sh[3][256];
		double s0 = 0, s1 = 0, s2 = 0;
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
		{
			const double av = a[k];
			s0 += av * b[k];
			const double t = c ? av + c[k] : av;
			s1 += t * t;
			const double m = fabs(av);
			s2 = m > s2 ? m : s2;
		}
		sh[0][threadIdx.x] = s0, sh[1][threadIdx.x] = s1, sh[2][threadIdx.x] = s2;
		__syncthreads();
		for (int s = 128; s > 0; s >>= 1)
		{
			if ((int)threadIdx.x < s)
			{
				sh[0][threadIdx.x] += sh[0][threadIdx.x + s];
				sh[1][threadIdx.x] += sh[1][threadIdx.x + s];
				const double o = sh[2][threadIdx.x + s];
				if (o > sh[2][threadIdx.x])
					sh[2][threadIdx.x] = o;
			}
			__syncthreads();
		}
		if (threadIdx.x == 0)
		{
			partial[blockIdx.x] = sh[0][0];
			partial[NB + blockIdx.x] = sh[1][0];
			partial[2 * NB + blockIdx.x] = sh[2][0];
		}
	}

	__global__ __launch_bounds__(256) void k_reduce3_finish(const double *__restrict__ partial, double *__restrict__ out)
	{
		__shared__ double sh[3][256];
		double s0 = 0, s1 = 0, s2 = 0;
		for (int k = threadIdx.x; k < NB; k += 256)
		{
			s0 += partial[k];
			s1 += partial[NB + k];
			const double m = partial[2 * NB + k];
			s2 = m > s2 ? m : s2;
		}
		sh[0][threadIdx.x] = s0, sh[1][threadIdx.x] = s1, sh[2][threadIdx.x] = s2;
		__syncthreads();
		for (int s = 128; s > 0; s >>= 1)
		{
			if ((int)threadIdx.x < s)
			{
				sh[0][threadIdx.x] += sh[0][threadIdx.x + s];
				sh[1][threadIdx.x] += sh[1][threadIdx.x + s];
				const double o = sh[2][threadIdx.x + s];
				if (o > sh[2][threadIdx.x])
					sh[2][threadIdx.x] = o;
			}
			__syncthreads();
		}
		if (threadIdx.x == 0)
			out[0] = sh[0][0], out[1] = sh[1][0], out[2] = sh[2][0];
	}

	// buf[0] = s[0], buf[1] = s[1] (sums), buf[2 + r] = s[2] for r == rank, 0 for the other ranks: summed over the ranks,
	// buf[2 ..] holds every rank's maximum
	__global__ void k_pack_reduce3(const double *__restrict__ s, int rank, int world, double *__restrict__ buf)
	{
		const int k = threadIdx.x;
		if (k < 2)
			buf[k] = s[k];
		else if (k < 2 + world)
			buf[k] = (k - 2 == rank) ? s[2] : 0.0;
	}
	__global__ void k_unpack_reduce3(const double *__restrict__ buf, int world, double *__restrict__ s)
	{
		if (threadIdx.x == 0)
		{
			double m = 0;
			for (int r = 0; r < world; ++r)
				m = buf[2 + r] > m ? buf[2 + r] : m;
			s[0] = buf[0], s[1] = buf[1], s[2] = m;
		}
	}
	// buf[0] = energy of the owned elements, buf[1] = 1 if an element of this rank is inverted
	__global__ void k_pack_energy(const double *__restrict__ e, const int *__restrict__ flag, double *__restrict__ buf)
	{
		if (threadIdx.x == 0)
			buf[0] = e ? *e : 0.0, buf[1] = *flag ? 1.0 : 0.0;
	}

	constexpr int DIST_BUF = 40; // doubles of the all-reduce staging area: 2 sums + one maximum per rank
	struct newton
	{
		pfx_ctx *ctx;
		pfx_dist *dist = nullptr; // the partition this context is one part of (null: the whole problem is here)
		int64_t n_red = 0;        // dofs the reductions run over: the owned ones
		double *dbuf = nullptr;   // DIST_BUF doubles
		const pfx_newton_params *par;
		pfx_newton_info *info;
		int64_t n;
		const double *rhs; // device, may be null
		const uint8_t *fixed;
		double *g, *dx, *xn, *w, *gls; // n each
		double *partial, *scal;        // 3*NB, 4
		// inertia term (null mass: static problem)
		double iw = 1.0;               // weight of the elastic and body forms
		const double *mass = nullptr, *xt = nullptr;
		double *d = nullptr, *md = nullptr; // n each: x - x_tilde and M (x - x_tilde)
		// augmented-Lagrangian penalty on the Dirichlet dofs (null mass: none)
		double alw = 0.0;
		const double *alm = nullptr, *alt = nullptr;
		double *alq = nullptr; // n

		// weight/2 (x - target)^T m (x - target)
		pfx_status al_value(const double *x, double *val)
		{
			hipLaunchKernelGGL(k_al_residual, dim3(NB), dim3(256), 0, ctx->stream, x, alm, alt, n, alq);
			double r[3];
			pfx_status rc = reduce(alq, alq, nullptr, r);
			*val = 0.5 * alw * r[0];
			return rc;
		}

		// d = x - x_tilde, md = M d
		void mass_apply(const double *x)
		{
			hipLaunchKernelGGL(k_step, dim3(NB), dim3(256), 0, ctx->stream, x, xt, -1.0, n, d);
			// owned rows (all of them on an unpartitioned mesh); d carries a current halo because x and x_tilde do
			hipLaunchKernelGGL(k_mass_apply, dim3(NB * 8), dim3(256), 0, ctx->stream, ctx->d_node_ptr, ctx->d_node_col, mass, (const double *)d, ctx->n_owned, ctx->bs, md);
		}

		// a.b, |a + c|^2 (c may be null), max|a|
		pfx_status reduce(const double *a, const double *b, const double *c, double out[3])
		{
			hipStream_t st = ctx->stream;
			hipLaunchKernelGGL(k_reduce3, dim3(NB), dim3(256), 0, st, a, b, c, n_red, partial);
			hipLaunchKernelGGL(k_reduce3_finish, dim3(1), dim3(256), 0, st, partial, scal);
			if (dist)
			{
				const int world = pfx_comm_world(dist->comm);
				hipLaunchKernelGGL(k_pack_reduce3, dim3(1), dim3(64), 0, st, scal, pfx_comm_rank(dist->comm), world, dbuf);
				if (pfx_comm_all_reduce(dist->comm, st, dbuf, 2 + world))
					PFX_FAIL(PFX_ERR_HIP, pfx_comm_error(dist->comm));
				hipLaunchKernelGGL(k_unpack_reduce3, dim3(1), dim3(64), 0, st, dbuf, world, scal);
			}
			PFX_HIP(hipMemcpyAsync(out, scal, 3 * sizeof(double), hipMemcpyDeviceToHost, st));
			PFX_HIP(hipStreamSynchronize(st));
			return PFX_OK;
		}

		// the energy a kernel left at `d_e` (null: none) and the inverted-element flag, over all ranks; the flag is cleared
		pfx_status read_energy_flag(const double *d_e, double *e, int *flag)
		{
			hipStream_t st = ctx->stream;
			*e = 0;
			if (dist)
			{
				hipLaunchKernelGGL(k_pack_energy, dim3(1), dim3(64), 0, st, d_e, ctx->d_flag, dbuf);
				if (pfx_comm_all_reduce(dist->comm, st, dbuf, 2))
					PFX_FAIL(PFX_ERR_HIP, pfx_comm_error(dist->comm));
				double h[2];
				PFX_HIP(hipMemcpyAsync(h, dbuf, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
				PFX_HIP(hipStreamSynchronize(st));
				*e = h[0], *flag = h[1] != 0.0;
			}
			else
			{
				if (d_e)
					PFX_HIP(hipMemcpyAsync(e, d_e, sizeof(double), hipMemcpyDeviceToHost, st));
				PFX_HIP(hipMemcpyAsync(flag, ctx->d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
				PFX_HIP(hipStreamSynchronize(st));
			}
			if (*flag)
				PFX_HIP(hipMemsetAsync(ctx->d_flag, 0, sizeof(int), st));
			return PFX_OK;
		}

		// Values at the current iterate, from one fused element-kernel launch (energy, gradient rows and the per-element
		// state the Hessian is assembled from); the line search asks for the same numbers again and gets them from here.
		double x_energy = 0.0, x_gn2 = 0.0;
		bool x_valid = false, x_state_valid = false;
		// last trial point of the line search: step size and objective
		double trial_step = -1.0, trial_energy = 0.0;

		// objective and gradient at the iterate x (into g), Hessian state left on the device
		pfx_status eval_iterate(const double *x, double *val, double *norm2)
		{
			hipStream_t st = ctx->stream;
			pfx_status rc = pfx_eval_all_dev(ctx, x, scal + 3, g);
			if (rc)
				return rc;
			double e = 0;
			int flag = 0;
			if ((rc = read_energy_flag(scal + 3, &e, &flag)))
				return rc;
			if (flag)
				e = std::nan("");
			double r[3];
			if (rhs && std::isfinite(e))
			{
				if ((rc = reduce(x, rhs, nullptr, r)))
					return rc;
				e -= r[0];
			}
			if (mass)
				mass_apply(x);
			if (mass && std::isfinite(e))
			{
				if ((rc = reduce(d, md, nullptr, r)))
					return rc;
				e = iw * e + 0.5 * r[0];
			}
			if (alm && std::isfinite(e))
			{
				double pen = 0;
				if ((rc = al_value(x, &pen)))
					return rc;
				e += pen;
			}
			hipLaunchKernelGGL(k_objective_gradient, dim3(NB), dim3(256), 0, st, g, rhs, fixed, n, iw, (const double *)(mass ? md : nullptr), alw, alm, alt, x);
			if ((rc = reduce(g, g, nullptr, r)))
				return rc;
			*val = e;
			*norm2 = flag ? std::nan("") : r[0];
			x_energy = *val, x_gn2 = *norm2;
			x_valid = x_state_valid = true;
			return PFX_OK;
		}

		// NLProblem::value: elastic energy - rhs.x ; nan when an element is inverted (the reference's log of a
		// non-positive determinant)
		pfx_status value(const double *x, double *val)
		{
			pfx_status rc = pfx_energy_dev(ctx, x, scal + 3);
			if (rc)
				return rc;
			double e = 0;
			int flag = 0;
			if ((rc = read_energy_flag(scal + 3, &e, &flag)))
				return rc;
			if (flag)
				e = std::nan("");
			if (rhs && std::isfinite(e))
			{
				double r[3];
				if ((rc = reduce(x, rhs, nullptr, r)))
					return rc;
				e -= r[0];
			}
			if (mass && std::isfinite(e))
			{
				double r[3];
				mass_apply(x);
				if ((rc = reduce(d, md, nullptr, r)))
					return rc;
				e = iw * e + 0.5 * r[0];
			}
			if (alm && std::isfinite(e))
			{
				double pen = 0;
				if ((rc = al_value(x, &pen)))
					return rc;
				e += pen;
			}
			*val = e;
			return PFX_OK;
		}

		// NLProblem::gradient into `out`; *norm2 = its squared norm (nan for an inverted element)
		pfx_status gradient(const double *x, double *out, double *norm2)
		{
			hipStream_t st = ctx->stream;
			pfx_status rc = pfx_gradient_dev(ctx, x, out);
			if (rc)
				return rc;
			if (mass)
				mass_apply(x);
			hipLaunchKernelGGL(k_objective_gradient, dim3(NB), dim3(256), 0, st, out, rhs, fixed, n, iw, (const double *)(mass ? md : nullptr), alw, alm, alt, x);
			double r[3];
			if ((rc = reduce(out, out, nullptr, r)))
				return rc;
			int flag = 0;
			double unused = 0;
			if ((rc = read_energy_flag(nullptr, &unused, &flag)))
				return rc;
			if (flag)
				r[0] = std::nan("");
			*norm2 = r[0];
			return PFX_OK;
		}

		pfx_status step(const double *x, double a, double *out)
		{
			hipLaunchKernelGGL(k_step, dim3(NB), dim3(256), 0, ctx->stream, x, dx, a, n, out);
			PFX_HIP(hipGetLastError());
			return PFX_OK;
		}

		// BacktrackingLineSearch::line_search without contact; *rate = nan on failure
		pfx_status line_search(const double *x, double *rate)
		{
			const double min_step_size = 0;
			const int max_step_size_iter = 100;
			int cur_iter = 0;
			pfx_status rc;
			double old_energy_in = 0;
			*rate = std::nan("");
			trial_step = -1.0;
			if (x_valid)
				old_energy_in = x_energy; // evaluated with the gradient at the top of the iteration
			else if ((rc = value(x, &old_energy_in)))
				return rc;
			info->line_search_iterations++;
			if (std::isnan(old_energy_in))
				return PFX_OK;
			double step_size = 1;
			// finite-energy step (LineSearch.tpp:90-145); ElasticForm::is_step_valid (gradient free of nan) holds
			// exactly when the energy is finite for the materials implemented here
			if ((rc = step(x, step_size, xn)))
				return rc;
			while (step_size > min_step_size && cur_iter < max_step_size_iter)
			{
				double energy = 0;
				if ((rc = value(xn, &energy)))
					return rc;
				trial_step = step_size, trial_energy = energy;
				info->line_search_iterations++;
				if (!std::isfinite(energy))
				{
					step_size *= 0.5;
					if ((rc = step(x, step_size, xn)))
						return rc;
				}
				else
					break;
				cur_iter++;
			}
			if (cur_iter >= max_step_size_iter || step_size <= min_step_size)
				return PFX_OK;
			// descent step (BacktrackingLineSearch.hpp:157-255)
			double gn2 = 0;
			if (x_valid)
				gn2 = x_gn2; // |gradient(x)|^2, known from the top of the iteration
			else if ((rc = gradient(x, gls, &gn2)))
				return rc;
			const bool use_grad_norm = std::sqrt(gn2) < par->use_grad_norm_tol;
			const double old_energy = use_grad_norm ? gn2 : old_energy_in;
			double cur_energy = std::nan("");
			while (step_size > min_step_size && cur_iter < max_step_size_iter)
			{
				if (!use_grad_norm && step_size == trial_step)
					cur_energy = trial_energy; // the point the finite-energy loop ended on: same objective value
				else
				{
					if ((rc = step(x, step_size, xn)))
						return rc;
					if (use_grad_norm)
						rc = gradient(xn, gls, &cur_energy);
					else
						rc = value(xn, &cur_energy);
					if (rc)
						return rc;
				}
				info->line_search_iterations++;
				if (!std::isfinite(cur_energy) || cur_energy > old_energy)
					step_size /= 2.0;
				else
					break;
				cur_iter++;
			}
			if (cur_iter >= max_step_size_iter || step_size <= min_step_size)
				return PFX_OK;
			*rate = step_size;
			return PFX_OK;
		}
	};

	int check_convergence(const pfx_newton_params *stop, int iterations, double f_delta, double grad_norm, double grad_tol)
	{
		if (stop->max_iterations > 0 && iterations > stop->max_iterations)
			return PFX_NEWTON_ITERATION_LIMIT;
		if (stop->f_delta > 0 && f_delta < stop->f_delta) // false for nan
			return PFX_NEWTON_F_DELTA;
		if (grad_tol > 0 && grad_norm < grad_tol)
			return PFX_NEWTON_GRAD_NORM;
		return PFX_NEWTON_CONTINUE;
	}
} // namespace

static pfx_status newton_impl(pfx_ctx *ctx, pfx_dist *dist, const double *d_rhs, double *d_x, const pfx_newton_params *par, pfx_newton_info *info)
{
	const int64_t n = ctx->n_dofs();
	if (ctx->newton_count < 8 * n + 3 * NB + 8 + DIST_BUF)
	{
		if (pfx_alloc(ctx, &ctx->d_newton, 8 * n + 3 * NB + 8 + DIST_BUF))
			return PFX_ERR_HIP;
		ctx->newton_count = 8 * n + 3 * NB + 8 + DIST_BUF;
	}
	if (!dist && ctx->n_owned != ctx->n_nodes)
		PFX_FAIL(PFX_ERR_ARG, "a partitioned context is solved through pfx_dist_newton_dev");
	if (dist && 2 + pfx_comm_world(dist->comm) > DIST_BUF)
		PFX_FAIL(PFX_ERR_UNSUPPORTED, "the partitioned Newton solve is limited to 38 ranks");
	newton S;
	S.ctx = ctx, S.dist = dist, S.par = par, S.info = info, S.n = n, S.n_red = ctx->n_owned * ctx->bs, S.rhs = d_rhs;
	S.fixed = ctx->n_dirichlet > 0 ? ctx->d_fixed : nullptr;
	S.g = ctx->d_newton, S.dx = S.g + n, S.xn = S.dx + n, S.w = S.xn + n, S.gls = S.w + n;
	S.d = S.gls + n, S.md = S.d + n, S.alq = S.md + n;
	S.partial = S.alq + n, S.scal = S.partial + 3 * NB, S.dbuf = S.scal + 8;
	if (ctx->inertia_weight > 0.0 && ctx->d_x_tilde && ctx->mass_count > 0)
	{
		if (ctx->bs < 2)
			PFX_FAIL(PFX_ERR_UNSUPPORTED, "the inertia term needs a vector-valued model");
		S.iw = ctx->inertia_weight, S.mass = ctx->d_mass_blk, S.xt = ctx->d_x_tilde;
	}
	if (ctx->al_weight > 0.0 && ctx->d_al_mass && ctx->d_al_target)
	{
		if (ctx->bs < 2 || dist)
			PFX_FAIL(PFX_ERR_UNSUPPORTED, "the augmented-Lagrangian term needs a vector-valued model and an unpartitioned mesh");
		S.alw = ctx->al_weight, S.alm = ctx->d_al_mass, S.alt = ctx->d_al_target;
	}
	hipStream_t st = ctx->stream;
	*info = pfx_newton_info{};
	PFX_HIP(hipMemsetAsync(S.dx, 0, sizeof(double) * (size_t)n, st));
	PFX_HIP(hipMemsetAsync(ctx->d_flag, 0, sizeof(int), st));
	const double saved_shift = ctx->diag_shift;
	if (dist)
	{
		// the caller's iterate with a current halo: from here on every update adds a halo-consistent direction
		if (pfx_halo_start(dist->comm, &dist->halo, st, d_x, ctx->bs) || pfx_halo_finish(dist->comm, &dist->halo, st))
			PFX_FAIL(PFX_ERR_HIP, pfx_comm_error(dist->comm));
	}

	pfx_status rc;
	double old_energy = std::nan(""), energy = std::nan(""), gn2 = 0;
	if ((rc = S.gradient(d_x, S.g, &gn2)))
		return rc;
	const double first_grad_norm = std::sqrt(gn2);
	double grad_norm = first_grad_norm;
	int status = PFX_NEWTON_CONTINUE, iterations = 0, strategy = 0;
	double reg_weight = 0;
	const double reg_min = 1e-8, reg_max = 1e8, reg_inc = 10, reg_dec = 2;
	auto increase_descent_strategy = [&]() {
		if (strategy == 0 || reg_weight > reg_max)
			strategy++;
		else
			reg_weight = reg_inc * reg_weight > reg_min ? reg_inc * reg_weight : reg_min;
	};
	auto finish = [&](pfx_status code, const char *msg) {
		ctx->diag_shift = saved_shift;
		info->status = status;
		info->iterations = iterations;
		info->energy = energy;
		info->grad_norm = grad_norm;
		if (code != PFX_OK)
			ctx->err = msg;
		return code;
	};

	if (std::isnan(first_grad_norm))
	{
		status = PFX_NEWTON_FAILED;
		return finish(PFX_ERR_NUMERIC, "Gradient is nan; stopping");
	}
	if (check_convergence(par, 0, std::nan(""), first_grad_norm / (par->relative_gradient ? first_grad_norm : 1), par->first_grad_norm_tol) != PFX_NEWTON_CONTINUE)
	{
		if ((rc = S.value(d_x, &energy)))
			return finish(rc, ctx->err.c_str());
		status = PFX_NEWTON_NOT_STARTED;
		return finish(PFX_OK, "");
	}

	do
	{
		if (S.x_valid) // the iterate did not move (a descent strategy was escalated): same values
			energy = S.x_energy, gn2 = S.x_gn2;
		else if ((rc = S.eval_iterate(d_x, &energy, &gn2)))
			return finish(rc, ctx->err.c_str());
		if (!std::isfinite(energy))
		{
			status = PFX_NEWTON_FAILED;
			return finish(PFX_ERR_NUMERIC, "f(x) is nan or inf; stopping");
		}
		grad_norm = std::sqrt(gn2);
		if (std::isnan(grad_norm))
		{
			status = PFX_NEWTON_FAILED;
			return finish(PFX_ERR_NUMERIC, "Gradient is nan; stopping");
		}

		// ---- compute_update_direction ----
		double dg = 0, dx_max = 0;
		bool have_direction = false;
		while (!have_direction)
		{
			if (strategy > 2)
			{
				status = PFX_NEWTON_FAILED;
				return finish(PFX_ERR_NUMERIC, "no descent direction found");
			}
			if (strategy == 2)
			{
				hipLaunchKernelGGL(k_negate, dim3(NB), dim3(256), 0, st, S.g, n, S.dx);
				if (dist && (pfx_halo_start(dist->comm, &dist->halo, st, S.dx, ctx->bs) || pfx_halo_finish(dist->comm, &dist->halo, st)))
				{
					ctx->err = pfx_comm_error(dist->comm);
					return finish(PFX_ERR_HIP, ctx->err.c_str());
				}
				double r[3];
				if ((rc = S.reduce(S.dx, S.g, nullptr, r)))
					return finish(rc, ctx->err.c_str());
				dg = r[0], dx_max = r[2];
				have_direction = true;
				break;
			}
			// assemble_hessian (:105-123): strategy 1 projects the element Hessians to PSD
			if ((rc = S.x_state_valid ? pfx_hessian_from_state(ctx, strategy == 1) : (strategy == 1 ? pfx_hessian_psd_dev(ctx, d_x) : pfx_hessian_dev(ctx, d_x))))
				return finish(rc, ctx->err.c_str());
			if (S.mass)
			{
				hipLaunchKernelGGL(k_add_inertia, dim3(NB * 8), dim3(256), 0, st, ctx->d_values, ctx->nnz, ctx->bs, S.iw, S.mass);
				ctx->values_gen++;
			}
			if (S.alm)
			{
				hipLaunchKernelGGL(k_al_add_diag, dim3(NB), dim3(256), 0, st, ctx->d_node_ptr, ctx->d_node_col, ctx->n_nodes, ctx->bs, S.alw, S.alm, ctx->d_values);
				ctx->values_gen++;
			}
			ctx->diag_shift = reg_weight;
			hipLaunchKernelGGL(k_negate, dim3(NB), dim3(256), 0, st, S.g, n, S.w);
			int its = 0;
			double err = 0;
			// the previous direction is the initial guess (the iterative solver is called with solveWithGuess)
			if (dist)
			{
				// the partitioned solve returns the direction with a current halo
				rc = par->cg_precond == PFX_PRECOND_AMG ? pfx_dist_pcg_amg_run(dist, S.w, S.dx, par->cg_max_iter, par->cg_tol, 0, &its, &err)
														: pfx_dist_cg_run(dist, S.w, S.dx, par->cg_precond, par->cg_max_iter, par->cg_tol, 0, &its, &err);
				if (rc)
					ctx->err = dist->err;
			}
			else
				rc = pfx_cg_dev(ctx, S.w, S.dx, par->cg_precond, par->cg_max_iter, par->cg_tol, &its, &err);
			if (rc)
				return finish(rc, ctx->err.c_str());
			info->cg_iterations += its;
			// check_direction: || H dx + g || and g.dx
			if ((rc = pfx_spmv_dev(ctx, S.dx, S.w)))
				return finish(rc, ctx->err.c_str());
			double r[3], r2[3];
			if ((rc = S.reduce(S.w, S.w, S.g, r)) || (rc = S.reduce(S.dx, S.g, nullptr, r2)))
				return finish(rc, ctx->err.c_str());
			ctx->diag_shift = saved_shift;
			const double residual = std::sqrt(r[1]);
			const double lim = 1e-8 * grad_norm > 1e-5 ? 1e-8 * grad_norm : 1e-5;
			dg = r2[0], dx_max = r2[2];
			if (std::isnan(residual) || residual > lim || dg >= 0)
			{
				increase_descent_strategy();
				continue;
			}
			reg_weight /= reg_dec;
			if (reg_weight < reg_min)
				reg_weight = 0;
			have_direction = true;
		}

		if (grad_norm != 0 && dg >= 0)
		{
			increase_descent_strategy();
			status = PFX_NEWTON_CONTINUE;
			continue;
		}
		if (std::isnan(dx_max))
		{
			increase_descent_strategy();
			status = PFX_NEWTON_CONTINUE;
			continue;
		}
		const double cur_grad_norm = par->use_grad_norm ? grad_norm / (par->relative_gradient ? first_grad_norm : 1) : dx_max;
		status = check_convergence(par, iterations, std::fabs(old_energy - energy), cur_grad_norm, par->grad_norm);
		old_energy = energy;

		double rate = 1;
		if (par->line_search)
		{
			if ((rc = S.line_search(d_x, &rate)))
				return finish(rc, ctx->err.c_str());
			if (std::isnan(rate))
			{
				if (strategy < 2)
				{
					increase_descent_strategy();
					status = PFX_NEWTON_CONTINUE;
					continue;
				}
				status = PFX_NEWTON_FAILED;
				return finish(PFX_ERR_NUMERIC, "Line search failed on gradient descent");
			}
		}
		hipLaunchKernelGGL(k_step, dim3(NB), dim3(256), 0, st, d_x, S.dx, rate, n, d_x);
		S.x_valid = S.x_state_valid = false;
		info->descent_strategy = strategy;
		strategy = 0;
		++iterations;
	} while (status == PFX_NEWTON_CONTINUE);

	PFX_HIP(hipStreamSynchronize(st));
	PFX_HIP(hipGetLastError());
	if (status == PFX_NEWTON_ITERATION_LIMIT)
		return finish(PFX_ERR_NUMERIC, "Reached iteration limit");
	return finish(PFX_OK, "");
}

pfx_status pfx_newton_run(pfx_ctx *ctx, const double *d_rhs, double *d_x, const pfx_newton_params *par, pfx_newton_info *info)
{
	return newton_impl(ctx, nullptr, d_rhs, d_x, par, info);
}

pfx_status pfx_dist_newton_run(pfx_dist *dist, const double *d_rhs, double *d_x, const pfx_newton_params *par, pfx_newton_info *info)
{
	pfx_status rc = newton_impl(dist->ctx, dist, d_rhs, d_x, par, info);
	if (rc)
		dist->err = dist->ctx->err;
	return rc;
}

pfx_status pfx_capture_mass(pfx_ctx *ctx)
{
	if (ctx->mass_count != ctx->n_blocks)
	{
		if (pfx_alloc(ctx, &ctx->d_mass_blk, ctx->n_blocks))
			return PFX_ERR_HIP;
		ctx->mass_count = ctx->n_blocks;
	}
	hipLaunchKernelGGL(k_capture_mass, dim3(NB * 8), dim3(256), 0, ctx->stream, ctx->d_values, ctx->n_blocks, ctx->bs, ctx->d_mass_blk);
	PFX_HIP(hipGetLastError());
	return PFX_OK;
}

// one implicit Euler step; over a partition (dist != null) x and v come with current halos and leave with current halos
static pfx_status euler_impl(pfx_ctx *ctx, pfx_dist *dist, double dt, const double *d_rhs, double *d_x, double *d_v, double *d_a,
							 const pfx_newton_params *params, pfx_newton_info *info)
{
	const int64_t n = ctx->n_dofs();
	hipStream_t st = ctx->stream;
	if (ctx->tstep_count < 2 * n)
	{
		if (pfx_alloc(ctx, &ctx->d_tstep, 2 * n))
			return PFX_ERR_HIP;
		ctx->tstep_count = 2 * n;
	}
	double *x_prev = ctx->d_tstep, *x_tilde = ctx->d_tstep + n;
	PFX_HIP(hipMemcpyAsync(x_prev, d_x, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, st));
	// ImplicitEuler::x_tilde = x_prev + dt v_prev
	hipLaunchKernelGGL(k_step, dim3(NB), dim3(256), 0, st, d_x, d_v, dt, n, x_tilde);
	const double saved_w = ctx->inertia_weight;
	const double *saved_xt = ctx->d_x_tilde;
	ctx->inertia_weight = dt * dt; // ImplicitEuler::acceleration_scaling
	ctx->d_x_tilde = x_tilde;
	const pfx_status rc = newton_impl(ctx, dist, d_rhs, d_x, params, info);
	ctx->inertia_weight = saved_w;
	ctx->d_x_tilde = saved_xt;
	if (rc)
		return rc;
	hipLaunchKernelGGL(k_euler_update, dim3(NB), dim3(256), 0, st, d_x, x_prev, dt, n, d_v, d_a);
	PFX_HIP(hipGetLastError());
	PFX_HIP(hipStreamSynchronize(st));
	return PFX_OK;
}

pfx_status pfx_implicit_euler_run(pfx_ctx *ctx, double dt, const double *d_rhs, double *d_x, double *d_v, double *d_a,
								  const pfx_newton_params *params, pfx_newton_info *info)
{
	return euler_impl(ctx, nullptr, dt, d_rhs, d_x, d_v, d_a, params, info);
}

pfx_status pfx_dist_implicit_euler_run(pfx_dist *dist, double dt, const double *d_rhs, double *d_x, double *d_v, double *d_a,
									   const pfx_newton_params *params, pfx_newton_info *info)
{
	const pfx_status rc = euler_impl(dist->ctx, dist, dt, d_rhs, d_x, d_v, d_a, params, info);
	if (rc)
		dist->err = dist->ctx->err;
	return rc;
}

// y[owned rows] = M v with the captured mass matrix (v needs a current halo on a partitioned mesh)
pfx_status pfx_lumped_mass_run(pfx_ctx *ctx, double *d_out)
{
	hipLaunchKernelGGL(k_lump_mass, dim3(NB), dim3(256), 0, ctx->stream, ctx->d_node_ptr, ctx->d_mass_blk, ctx->n_nodes, ctx->bs, d_out);
	PFX_HIP(hipGetLastError());
	return PFX_OK;
}

pfx_status pfx_mass_apply_run(pfx_ctx *ctx, const double *d_v, double *d_y)
{
	hipLaunchKernelGGL(k_mass_apply, dim3(NB * 8), dim3(256), 0, ctx->stream, ctx->d_node_ptr, ctx->d_node_col, ctx->d_mass_blk, d_v, ctx->n_owned, ctx->bs, d_y);
	PFX_HIP(hipGetLastError());
	return PFX_OK;
}

// values = w * values + M on the resident matrix
pfx_status pfx_add_mass_run(pfx_ctx *ctx, double w)
{
	hipLaunchKernelGGL(k_add_inertia, dim3(NB * 8), dim3(256), 0, ctx->stream, ctx->d_values, ctx->nnz, ctx->bs, w, ctx->d_mass_blk);
	ctx->values_gen++;
	ctx->have_mask = false;
	PFX_HIP(hipGetLastError());
	return PFX_OK;
}
