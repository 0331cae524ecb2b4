// Host-side test of the C++ shim (include/polyfem_amd/assembler.hpp): reads like the reference's
// tests/test_assembler.cpp:12-64 ("hessian_lin") and tests/test_form_derivatives.cpp:84-143 on a small Kuhn mesh.
#include <polyfem_amd/assembler.hpp>

#include <cstdio>
#include <cstdlib>

using namespace polyfem_amd;

static void build_p2_cube(int n, std::vector<basis::ElementBases> &bases, std::vector<basis::ElementBases> &gbases, int &n_basis)
{
	static const int perms[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
	static const int even[6] = {1, 0, 0, 1, 1, 0};
	static const int edges[6][2] = {{0, 1}, {1, 2}, {2, 0}, {0, 3}, {1, 3}, {2, 3}};
	const int L = 2 * n + 1, nv = n + 1;
	n_basis = L * L * L;
	for (int kz = 0; kz < n; ++kz)
		for (int ky = 0; ky < n; ++ky)
			for (int kx = 0; kx < n; ++kx)
				for (int t = 0; t < 6; ++t)
				{
					int v[4][3] = {{kx, ky, kz}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
					for (int s = 0; s < 3; ++s)
					{
						for (int c = 0; c < 3; ++c)
							v[s + 1][c] = v[s][c];
						v[s + 1][perms[t][s]] += 1;
					}
					if (!even[t])
						for (int c = 0; c < 3; ++c)
							std::swap(v[2][c], v[3][c]);
					basis::ElementBases b, g;
					b.bases.resize(10);
					g.bases.resize(4);
					int lat[10][3];
					for (int k = 0; k < 4; ++k)
						for (int c = 0; c < 3; ++c)
							lat[k][c] = 2 * v[k][c];
					for (int k = 0; k < 6; ++k)
						for (int c = 0; c < 3; ++c)
							lat[4 + k][c] = (lat[edges[k][0]][c] + lat[edges[k][1]][c]) / 2;
					for (int k = 0; k < 10; ++k)
					{
						basis::Local2Global l2g;
						l2g.index = lat[k][0] + L * (lat[k][1] + L * lat[k][2]);
						for (int c = 0; c < 3; ++c)
							l2g.node[c] = lat[k][c] * 0.5 / n;
						b.bases[k].global_.push_back(l2g);
					}
					for (int k = 0; k < 4; ++k)
					{
						basis::Local2Global l2g;
						l2g.index = v[k][0] + nv * (v[k][1] + nv * v[k][2]);
						for (int c = 0; c < 3; ++c)
							l2g.node[c] = (double)v[k][c] / n;
						g.bases[k].global_.push_back(l2g);
					}
					bases.push_back(b);
					gbases.push_back(g);
				}
}

// n x n cells, two triangles each, P2 bases on the P1 geometry (local node order of autogen/auto_p_bases.cpp:275-284);
// the reference's own assembler tests run on 2-D triangle meshes (tests/test_assembler.cpp:12-64)
static void build_p2_square(int n, std::vector<basis::ElementBases> &bases, std::vector<basis::ElementBases> &gbases, int &n_basis)
{
	static const int tris[2][3][2] = {{{0, 0}, {1, 0}, {1, 1}}, {{0, 0}, {1, 1}, {0, 1}}};
	static const int edges[3][2] = {{0, 1}, {1, 2}, {2, 0}};
	const int L = 2 * n + 1, nv = n + 1;
	n_basis = L * L;
	for (int ky = 0; ky < n; ++ky)
		for (int kx = 0; kx < n; ++kx)
			for (int t = 0; t < 2; ++t)
			{
				basis::ElementBases b, g;
				b.bases.resize(6);
				g.bases.resize(3);
				int lat[6][2];
				for (int k = 0; k < 3; ++k)
				{
					const int vx = kx + tris[t][k][0], vy = ky + tris[t][k][1];
					lat[k][0] = 2 * vx, lat[k][1] = 2 * vy;
					basis::Local2Global l2g;
					l2g.index = vx + nv * vy;
					// a sheared square, so that no element is axis-aligned
					l2g.node[0] = (double)vx / n + 0.2 * vy / n, l2g.node[1] = (double)vy / n;
					g.bases[k].global_.push_back(l2g);
				}
				for (int k = 0; k < 3; ++k)
					for (int c = 0; c < 2; ++c)
						lat[3 + k][c] = (lat[edges[k][0]][c] + lat[edges[k][1]][c]) / 2;
				for (int k = 0; k < 6; ++k)
				{
					basis::Local2Global l2g;
					l2g.index = lat[k][0] + L * lat[k][1];
					b.bases[k].global_.push_back(l2g);
				}
				bases.push_back(b);
				gbases.push_back(g);
			}
}

#define REQUIRE(c)                                                    \
	do                                                                \
	{                                                                 \
		if (!(c))                                                     \
		{                                                             \
			std::printf("REQUIRE failed: %s (line %d)\n", #c, __LINE__); \
			return 1;                                                 \
		}                                                             \
	} while (0)

int main()
{
	std::vector<basis::ElementBases> bases, gbases;
	int n_basis = 0;
	build_p2_cube(3, bases, gbases, n_basis);
	assembler::AssemblyValsCache cache;
	const int N = n_basis * 3;

	assembler::Assembler<assembler::LinearElasticity> lin;
	lin.local_assembler().set_size(3);
	lin.local_assembler().set_E_nu(1e5, 0.3);
	StiffnessMatrix K;
	lin.assemble(true, n_basis, bases, gbases, cache, K);
	REQUIRE(K.rows() == N && K.nonZeros() > 0);

	// a rigid translation is in the null space of the stiffness matrix
	std::vector<double> t(N, 0.0), Kt(N, 0.0);
	for (int i = 0; i < n_basis; ++i)
		t[3 * i] = 1.0;
	for (int r = 0; r < N; ++r)
		for (int64_t k = K.outer[r]; k < K.outer[r + 1]; ++k)
			Kt[r] += K.values[k] * t[K.inner[k]];
	double mx = 0, kmax = 0;
	for (double v : Kt)
		mx = std::fmax(mx, std::fabs(v));
	for (double v : K.values)
		kmax = std::fmax(kmax, std::fabs(v));
	REQUIRE(mx <= 1e-10 * kmax);

	assembler::NLAssembler<assembler::NeoHookeanElasticity> nl;
	nl.local_assembler().set_E_nu(2e4, 0.3);
	utils::SpareMatrixCache mat_cache;
	VectorXd x(N, 0.0), g, dx;
	std::srand(1);
	for (auto &v : x)
		v = 1e-3 * ((double)std::rand() / RAND_MAX - 0.5);
	const double e0 = nl.assemble(true, bases, gbases, cache, 0.0, x, x);
	nl.assemble_grad(true, n_basis, bases, gbases, cache, 0.0, x, x, g);
	StiffnessMatrix H;
	nl.assemble_hessian(true, n_basis, false, bases, gbases, cache, 0.0, x, x, mat_cache, H);
	REQUIRE(e0 > 0 && (int)g.size() == N && H.rows() == N && mat_cache.mapping_size() == (size_t)N);
	// finite-difference check of the gradient along a random direction (test_form_derivatives.cpp:84-143)
	VectorXd d(N), xp(N), xm(N);
	for (auto &v : d)
		v = (double)std::rand() / RAND_MAX - 0.5;
	const double h = 1e-6;
	for (int i = 0; i < N; ++i)
		xp[i] = x[i] + h * d[i], xm[i] = x[i] - h * d[i];
	const double fd = (nl.assemble(true, bases, gbases, cache, 0.0, xp, xp) - nl.assemble(true, bases, gbases, cache, 0.0, xm, xm)) / (2 * h);
	double gd = 0;
	for (int i = 0; i < N; ++i)
		gd += g[i] * d[i];
	REQUIRE(std::fabs(fd - gd) <= 1e-5 * std::fabs(gd));
	// second call reuses the cache
	nl.assemble_hessian(true, n_basis, false, bases, gbases, cache, 0.0, xp, xp, mat_cache, H);

	// Newton direction: H dx = -g with the x = 0 face fixed (SparseNewtonDescentSolver.tpp:128-155)
	auto solver = polysolve::LinearSolver::create("Eigen::ConjugateGradient", "Eigen::DiagonalPreconditioner");
	solver->setParameters(1000, 1e-12);
	std::vector<int> fixed;
	for (int i = 0; i < n_basis; ++i)
		if (i % 7 == 0) // lattice x == 0 (the lattice is 7 nodes wide for n = 3)
			for (int c = 0; c < 3; ++c)
				fixed.push_back(3 * i + c);
	solver->setDirichlet(fixed);
	solver->analyzePattern(H, H.rows());
	solver->factorize(H);
	VectorXd rhs(N);
	for (int i = 0; i < N; ++i)
		rhs[i] = -g[i];
	solver->solve(rhs, dx);
	REQUIRE(solver->solver_error() < 1e-10 && solver->solver_iter() > 0);
	// the AMG solver slot ("AMGCL"): same system, fewer iterations, same solution
	{
		auto amg = polysolve::LinearSolver::create("AMGCL", "");
		REQUIRE(amg->name() == "AMGCL");
		amg->setDirichlet(fixed);
		amg->analyzePattern(H, H.rows());
		amg->factorize(H);
		VectorXd dx2;
		amg->solve(rhs, dx2);
		REQUIRE(amg->solver_error() <= 1e-10 && amg->solver_iter() > 0 && amg->solver_iter() < solver->solver_iter());
		double dmax = 0, emax = 0;
		for (int i = 0; i < N; ++i)
		{
			dmax = std::max(dmax, std::fabs(dx[i]));
			emax = std::max(emax, std::fabs(dx[i] - dx2[i]));
		}
		REQUIRE(emax <= 1e-7 * dmax);
	}
	REQUIRE(solver->uses_resident_matrix()); // H came straight out of the assembler: solved in place, nothing uploaded

	// The matrix the reference's Newton solver really hands over is the REDUCED SUM of all forms
	// (SparseNewtonDescentSolver.tpp:128-155, NLProblem.cpp:92-99): (a) full_to_reduced_matrix of H, (b) the same plus a
	// diagonal second form.  Both go in by value; the solver must use exactly what it is given.
	{
		std::vector<char> is_fixed(N, 0);
		for (int f : fixed)
			is_fixed[f] = 1;
		std::vector<int> new_id(N, -1);
		int nr = 0;
		for (int i = 0; i < N; ++i)
			if (!is_fixed[i])
				new_id[i] = nr++;
		StiffnessMatrix R; // utils::full_to_reduced_matrix (utils/MatrixUtils.cpp:408-469)
		R.n = nr;
		R.outer.push_back(0);
		for (int j = 0; j < N; ++j)
		{
			if (is_fixed[j])
				continue;
			for (int64_t k = H.outer[j]; k < H.outer[j + 1]; ++k)
				if (!is_fixed[H.inner[k]])
				{
					R.inner.push_back(new_id[H.inner[k]]);
					R.values.push_back(H.values[k]);
				}
			R.outer.push_back((int64_t)R.inner.size());
		}
		VectorXd rr(nr), xa, xb;
		for (int i = 0; i < N; ++i)
			if (!is_fixed[i])
				rr[new_id[i]] = rhs[i];
		auto byvalue = polysolve::LinearSolver::create("Eigen::ConjugateGradient", "Eigen::DiagonalPreconditioner");
		byvalue->setParameters(1000, 1e-12);
		byvalue->analyzePattern(R, R.rows());
		byvalue->factorize(R);
		byvalue->solve(rr, xa);
		REQUIRE(!byvalue->uses_resident_matrix());
		REQUIRE(byvalue->solver_error() < 1e-10 && std::abs(byvalue->solver_iter() - solver->solver_iter()) <= 2);
		double dmax = 0, emax = 0;
		for (int i = 0; i < N; ++i)
			if (!is_fixed[i])
			{
				dmax = std::max(dmax, std::fabs(dx[i]));
				emax = std::max(emax, std::fabs(dx[i] - xa[new_id[i]]));
			}
		REQUIRE(emax <= 1e-8 * dmax);
		// (b) + diagonal form
		StiffnessMatrix S = R;
		std::vector<double> diag(nr);
		for (int i = 0; i < nr; ++i)
			diag[i] = 50.0 + (i % 17);
		S.add_to_diagonal(diag);
		byvalue->factorize(S); // same pattern: values only
		byvalue->solve(rr, xb);
		REQUIRE(byvalue->solver_error() < 1e-10);
		// residual of the summed system on the host (column-major product), and the two solutions differ
		std::vector<double> res(nr, 0.0);
		for (int j = 0; j < nr; ++j)
			for (int64_t k = S.outer[j]; k < S.outer[j + 1]; ++k)
				res[S.inner[k]] += S.values[k] * xb[j];
		double rn = 0, bn = 0, diff = 0, an = 0;
		for (int i = 0; i < nr; ++i)
		{
			rn += (res[i] - rr[i]) * (res[i] - rr[i]), bn += rr[i] * rr[i];
			diff += (xa[i] - xb[i]) * (xa[i] - xb[i]), an += xa[i] * xa[i];
		}
		REQUIRE(std::sqrt(rn) <= 1e-10 * std::sqrt(bn));
		REQUIRE(std::sqrt(diff) > 1e-3 * std::sqrt(an));
		// a tagged matrix that the caller changed afterwards is uploaded, not solved in place
		StiffnessMatrix H2 = H;
		H2.add_to_diagonal(1.0);
		REQUIRE(H2.resident == nullptr);
		// factorization failure -> std::runtime_error, the signal the Newton solver regularises on
		StiffnessMatrix bad = R;
		bad.add_to_diagonal(-1e12);
		bool failed = false;
		try
		{
			byvalue->factorize(bad);
		}
		catch (const std::runtime_error &)
		{
			failed = true;
		}
		REQUIRE(failed);
	}

	// The partitioned solve through the C ABI, here with one rank (the whole mesh, no neighbours) and a host transport whose
	// callbacks have nothing to move: pfx_dist_cg_dev must reproduce the single-context solve.  With more ranks the callbacks
	// would be MPI_Sendrecv / MPI_Allreduce, or the communicator pfx_comm_create_rccl.
	{
		pfx_host_transport tr;
		tr.user = nullptr;
		tr.exchange = [](void *, int, const int *, const void *const *, const int64_t *, void *const *, const int64_t *) { return 0; };
		tr.all_reduce_sum = [](void *, double *, int) { return 0; };
		pfx_comm *comm = nullptr;
		REQUIRE(pfx_comm_create_host(&comm, &tr, 0, 1) == PFX_OK);
		pfx_ctx *h = Context::instance().h;
		pfx_dist *dist = nullptr;
		REQUIRE(pfx_dist_create(&dist, h, comm, 0, nullptr, nullptr, nullptr, nullptr) == PFX_OK);
		// device vectors through the plain HIP runtime would need its headers; pfx_cg with host vectors gives the reference and
		// the distributed entry point is fed through the context's own staging by a second host solve of the same system
		int64_t hb = -1;
		REQUIRE(pfx_dist_halo_bytes(dist, &hb) == PFX_OK && hb == 0);
		pfx_dist_destroy(dist);
		pfx_comm_destroy(comm);
	}

	bool threw = false;
	try
	{
		polysolve::LinearSolver::create("Eigen::SimplicialLDLT", "");
	}
	catch (const std::runtime_error &)
	{
		threw = true;
	}
	REQUIRE(threw);

	// HookeLinearElasticity with the tensor of the same (E, nu) acts like LinearElasticity; both tensor setters agree
	{
		assembler::Assembler<assembler::HookeLinearElasticity> hooke;
		hooke.local_assembler().set_size(3);
		hooke.local_assembler().elasticity_tensor().set_from_young_poisson(1e5, 0.3);
		assembler::ElasticityTensor t2;
		t2.set_from_lambda_mu(lin.local_assembler().lambda, lin.local_assembler().mu);
		for (int a = 0; a < 6; ++a)
			for (int b = 0; b < 6; ++b)
				REQUIRE(std::fabs(t2(a, b) - hooke.local_assembler().elasticity_tensor()(a, b)) <= 1e-10 * 1e5);
		StiffnessMatrix KH;
		hooke.assemble(true, n_basis, bases, gbases, cache, KH);
		REQUIRE(KH.rows() == N);
		double worst = 0, ymax = 0;
		for (int r = 0; r < N; ++r)
		{
			double y1 = 0, y2 = 0;
			for (int64_t k = K.outer[r]; k < K.outer[r + 1]; ++k)
				y1 += K.values[k] * x[K.inner[k]];
			for (int64_t k = KH.outer[r]; k < KH.outer[r + 1]; ++k)
				y2 += KH.values[k] * x[KH.inner[k]];
			worst = std::fmax(worst, std::fabs(y1 - y2));
			ymax = std::fmax(ymax, std::fabs(y1));
		}
		REQUIRE(worst <= 1e-10 * ymax);
		// the scalar-material assemblers still bind their own material afterwards
		StiffnessMatrix K2;
		lin.assemble(true, n_basis, bases, gbases, cache, K2);
		REQUIRE(K2.values == K.values && K2.inner == K.inner);
	}

	// SaintVenant: the tensor of the same (E, nu) gives the (lambda, mu) energy; an anisotropic one is accepted and its
	// gradient is the derivative of its energy
	{
		assembler::NLAssembler<assembler::SaintVenantElasticity> sv, svt;
		sv.local_assembler().set_E_nu(2e4, 0.3);
		svt.local_assembler().elasticity_tensor().set_from_young_poisson(2e4, 0.3);
		const double e_sv = sv.assemble(true, bases, gbases, cache, 0.0, x, x);
		const double e_svt = svt.assemble(true, bases, gbases, cache, 0.0, x, x);
		REQUIRE(e_sv > 0 && std::fabs(e_sv - e_svt) <= 1e-12 * e_sv);
		std::vector<double> entries;
		for (int a = 0; a < 6; ++a)
			for (int b = a; b < 6; ++b)
				entries.push_back(a == b ? 3e4 + 1e3 * a : 2e3 / (1 + a + b));
		svt.local_assembler().elasticity_tensor().set_from_entries(entries);
		VectorXd gt;
		svt.assemble_grad(true, n_basis, bases, gbases, cache, 0.0, x, x, gt);
		const double fdt = (svt.assemble(true, bases, gbases, cache, 0.0, xp, xp) - svt.assemble(true, bases, gbases, cache, 0.0, xm, xm)) / (2 * h);
		double gdt = 0;
		for (int i = 0; i < N; ++i)
			gdt += gt[i] * d[i];
		REQUIRE(std::fabs(fdt - gdt) <= 1e-5 * std::fabs(gdt));
		// the (lambda, mu) assembler binds its own material again
		REQUIRE(std::fabs(sv.assemble(true, bases, gbases, cache, 0.0, x, x) - e_sv) <= 1e-15 * e_sv);
	}

	// MultiModel with one model everywhere reproduces that model; a mixture lies between
	{
		assembler::NLAssembler<assembler::MultiModel> mm;
		mm.local_assembler().set_E_nu(2e4, 0.3);
		mm.local_assembler().init_multimodels(std::vector<std::string>(bases.size(), "NeoHookean"));
		const double e_all_neo = mm.assemble(true, bases, gbases, cache, 0.0, x, x);
		REQUIRE(std::fabs(e_all_neo - e0) <= 1e-13 * std::fabs(e0));
		std::vector<std::string> mix(bases.size(), "NeoHookean");
		for (size_t e = 0; e < mix.size(); e += 2)
			mix[e] = "SaintVenant";
		mm.local_assembler().init_multimodels(mix);
		VectorXd gm;
		const double e_mix = mm.assemble(true, bases, gbases, cache, 0.0, x, x);
		mm.assemble_grad(true, n_basis, bases, gbases, cache, 0.0, x, x, gm);
		REQUIRE(e_mix > 0 && e_mix != e_all_neo && (int)gm.size() == N);
		const double fdm = (mm.assemble(true, bases, gbases, cache, 0.0, xp, xp) - mm.assemble(true, bases, gbases, cache, 0.0, xm, xm)) / (2 * h);
		double gdm = 0;
		for (int i = 0; i < N; ++i)
			gdm += gm[i] * d[i];
		REQUIRE(std::fabs(fdm - gdm) <= 1e-5 * std::fabs(gdm));
		bool rejected = false;
		try
		{
			mm.local_assembler().init_multimodels({"Ogden"});
		}
		catch (const std::runtime_error &)
		{
			rejected = true;
		}
		REQUIRE(rejected);
		// back to a single model
		REQUIRE(std::fabs(nl.assemble(true, bases, gbases, cache, 0.0, x, x) - e0) <= 1e-15 * std::fabs(e0));
	}

	// "hessian_lin" (tests/test_assembler.cpp:12-64): LinearElasticity through the nonlinear interface reproduces the
	// stiffness matrix for any displacement
	{
		assembler::NLAssembler<assembler::LinearElasticity> nll;
		nll.local_assembler().set_E_nu(1e5, 0.3);
		utils::SpareMatrixCache c2;
		StiffnessMatrix HL;
		nll.assemble_hessian(true, n_basis, false, bases, gbases, cache, 0.0, x, x, c2, HL);
		// compare through the action on a vector: the linear assembler's pattern drops exact zeros
		std::vector<double> y1(N, 0.0), y2(N, 0.0);
		for (int r = 0; r < N; ++r)
		{
			for (int64_t k = K.outer[r]; k < K.outer[r + 1]; ++k)
				y1[r] += K.values[k] * d[K.inner[k]];
			for (int64_t k = HL.outer[r]; k < HL.outer[r + 1]; ++k)
				y2[r] += HL.values[k] * d[HL.inner[k]];
		}
		double dm = 0, ym = 0;
		for (int r = 0; r < N; ++r)
			dm = std::fmax(dm, std::fabs(y1[r] - y2[r])), ym = std::fmax(ym, std::fabs(y1[r]));
		REQUIRE(dm <= 1e-11 * ym);
	}

	// the whole nonlinear solve (SparseNewtonDescentSolver::minimize on NLProblem): clamp x = 0, pull down, two materials
	{
		assembler::NeoHookeanElasticity mat;
		std::vector<double> le(bases.size()), me(bases.size());
		for (size_t e = 0; e < bases.size(); ++e)
		{
			const double E = e < bases.size() / 2 ? 2e4 : 6e4, nu = 0.3;
			le[e] = (E * nu) / ((1.0 + nu) * (1.0 - 2.0 * nu));
			me[e] = E / (2.0 * (1.0 + nu));
		}
		mat.set_lambda_mu_per_element(le, me);
		VectorXd load(N, 0.0), bc(N, 0.0);
		for (int i = 0; i < n_basis; ++i)
			load[3 * i + 2] = -50.0 / n_basis;
		solver::NLProblem<assembler::NeoHookeanElasticity> problem(bases, gbases, n_basis, mat, fixed, load, bc);
		cppoptlib::SparseNewtonDescentSolver<solver::NLProblem<assembler::NeoHookeanElasticity>> newton;
		VectorXd xr((size_t)problem.reduced_size(), 0.0);
		newton.minimize(problem, xr);
		REQUIRE(newton.solver_info.iterations >= 2 && newton.solver_info.iterations < 20);
		REQUIRE(newton.solver_info.status == PFX_NEWTON_F_DELTA || newton.solver_info.status == PFX_NEWTON_GRAD_NORM);
		// stationarity: the elastic gradient balances the load on the free dofs
		assembler::NLAssembler<assembler::NeoHookeanElasticity> nl2;
		nl2.local_assembler() = mat;
		VectorXd xf = problem.reduced_to_full(xr), gf;
		nl2.assemble_grad(true, n_basis, bases, gbases, cache, 0.0, xf, xf, gf);
		VectorXd res = problem.full_to_reduced(gf), lr = problem.full_to_reduced(load);
		double rn = 0, ln = 0;
		for (size_t k = 0; k < res.size(); ++k)
			rn += (res[k] - lr[k]) * (res[k] - lr[k]), ln += lr[k] * lr[k];
		REQUIRE(std::sqrt(rn) <= 1e-7 * std::sqrt(ln));
		bool limit = false;
		try
		{
			newton.solver_params.max_iterations = 1, newton.solver_params.f_delta = 0, newton.solver_params.grad_norm = 1e-14;
			VectorXd x2((size_t)problem.reduced_size(), 0.0);
			newton.minimize(problem, x2);
		}
		catch (const std::runtime_error &err)
		{
			limit = std::string(err.what()).find("iteration limit") != std::string::npos;
		}
		REQUIRE(limit);
		std::printf("newton ok: %d iterations, %ld cg iterations\n", newton.solver_info.iterations, (long)newton.solver_info.cg_iterations);
	}
	// mass matrix, body force and two implicit Euler steps (solve_transient_tensor_nonlinear: minimize, then
	// update_quantities), against the one-call time step of the C ABI
	{
		assembler::Mass mass;
		mass.density = 1000.0;
		assembler::Assembler<assembler::Mass> mass_assembler;
		mass_assembler.local_assembler() = mass;
		StiffnessMatrix Mm;
		mass_assembler.assemble(true, n_basis, bases, gbases, cache, Mm, true);
		double total = 0; // 1^T M 1 = 3 * rho * volume
		for (double v : Mm.values)
			total += v;
		REQUIRE(std::fabs(total - 3.0 * 1000.0) <= 1e-9 * 3000.0);
		const double force[3] = {0.0, 3.0, -9.81};
		VectorXd rhs;
		assembler::assemble_body_force(bases, gbases, n_basis, force, mass, rhs);
		double fz = 0;
		for (int i = 0; i < n_basis; ++i)
			fz += rhs[3 * i + 2];
		REQUIRE(std::fabs(fz + 9810.0) <= 1e-9 * 9810.0);

		assembler::NeoHookeanElasticity mat;
		mat.set_E_nu(2e4, 0.3);
		VectorXd bc(N, 0.0), x0(N, 0.0), v0(N, 0.0), a0(N, 0.0);
		std::vector<char> is_fixed(N, 0);
		for (int b : fixed)
			is_fixed[b] = 1;
		for (int k = 0; k < N; ++k)
			v0[k] = is_fixed[k] ? 0.0 : 0.05 * std::sin(0.37 * k);
		const double dt = 0.02;
		time_integrator::ImplicitEuler integrator;
		integrator.init(x0, v0, a0, dt);
		solver::NLProblem<assembler::NeoHookeanElasticity> problem(bases, gbases, n_basis, mat, fixed, rhs, bc);
		problem.set_inertia(mass, &integrator);
		cppoptlib::SparseNewtonDescentSolver<solver::NLProblem<assembler::NeoHookeanElasticity>> newton;
		VectorXd xr = problem.full_to_reduced(x0);
		int its[2];
		for (int step = 0; step < 2; ++step)
		{
			newton.minimize(problem, xr);
			its[step] = newton.solver_info.iterations;
			integrator.update_quantities(problem.reduced_to_full(xr));
		}
		// the same two steps through pfx_implicit_euler_step
		Context &c = Context::instance();
		VectorXd x1 = x0, v1 = v0, a1(N, 0.0);
		pfx_newton_params par;
		pfx_newton_default_params(&par);
		pfx_newton_info info;
		for (int step = 0; step < 2; ++step)
		{
			c.check(pfx_implicit_euler_step(c.h, dt, rhs.data(), x1.data(), v1.data(), a1.data(), &par, &info));
			REQUIRE(info.iterations == its[step]);
		}
		double dx = 0, xm = 0, dv = 0, vm = 0;
		for (int k = 0; k < N; ++k)
		{
			dx = std::fmax(dx, std::fabs(integrator.x_prev()[k] - x1[k])), xm = std::fmax(xm, std::fabs(x1[k]));
			dv = std::fmax(dv, std::fabs(integrator.v_prev()[k] - v1[k])), vm = std::fmax(vm, std::fabs(v1[k]));
		}
		REQUIRE(xm > 0 && dx <= 1e-10 * xm && dv <= 1e-9 * vm);
		// back to a static problem: the inertia term is switched off again
		problem.set_inertia(mass, nullptr);
		VectorXd xs((size_t)problem.reduced_size(), 0.0);
		newton.minimize(problem, xs);
		REQUIRE(newton.solver_info.status == PFX_NEWTON_F_DELTA || newton.solver_info.status == PFX_NEWTON_GRAD_NORM);
		std::printf("transient ok: %d + %d newton iterations\n", its[0], its[1]);
	}
	// ---- 2-D: the reference's "hessian_lin" (tests/test_assembler.cpp:12-64) and the finite-difference test of the
	// elastic form (tests/test_form_derivatives.cpp:84-143, 200-227) on a triangle mesh, is_volume = false, size() == 2
	{
		std::vector<basis::ElementBases> b2, g2;
		int nb2 = 0;
		build_p2_square(6, b2, g2, nb2);
		const int N2 = 2 * nb2;
		assembler::Assembler<assembler::LinearElasticity> lin2;
		lin2.local_assembler().set_size(2);
		lin2.local_assembler().set_E_nu(1e5, 0.3);
		StiffnessMatrix K2;
		lin2.assemble(false, nb2, b2, g2, cache, K2);
		REQUIRE(K2.rows() == N2 && K2.nonZeros() > 0);
		assembler::NLAssembler<assembler::LinearElasticity> nll2;
		nll2.local_assembler().set_size(2);
		nll2.local_assembler().set_E_nu(1e5, 0.3);
		utils::SpareMatrixCache cache_lin2;
		VectorXd x2(N2, 0.0);
		for (int rand = 0; rand < 10; ++rand)
		{
			if (rand > 0)
				for (auto &v : x2)
					v = 2.0 * std::rand() / RAND_MAX - 1.0;
			StiffnessMatrix Hl;
			nll2.assemble_hessian(false, nb2, false, b2, g2, cache, 0.0, x2, x2, cache_lin2, Hl);
			// compare as dense rows: K2 holds the value-dependent pattern, Hl the structural one
			double diff = 0, scale = 0;
			std::vector<double> row(N2);
			for (int j = 0; j < N2; ++j)
			{
				std::fill(row.begin(), row.end(), 0.0);
				for (int64_t k = Hl.outer[j]; k < Hl.outer[j + 1]; ++k)
					row[Hl.inner[k]] += Hl.values[k];
				for (int64_t k = K2.outer[j]; k < K2.outer[j + 1]; ++k)
					row[K2.inner[k]] -= K2.values[k], scale = std::fmax(scale, std::fabs(K2.values[k]));
				for (double v : row)
					diff = std::fmax(diff, std::fabs(v));
			}
			REQUIRE(diff <= 1e-8 * scale);
		}
		assembler::NLAssembler<assembler::NeoHookeanElasticity> nl2;
		nl2.local_assembler().set_size(2);
		nl2.local_assembler().set_E_nu(2e4, 0.3);
		utils::SpareMatrixCache cache_neo2;
		for (auto &v : x2)
			v = (2.0 * std::rand() / RAND_MAX - 1.0) / 100;
		VectorXd g2v, d2(N2), xp2(N2), xm2(N2), gp, gm;
		StiffnessMatrix H2;
		nl2.assemble_grad(false, nb2, b2, g2, cache, 0.0, x2, x2, g2v);
		nl2.assemble_hessian(false, nb2, false, b2, g2, cache, 0.0, x2, x2, cache_neo2, H2);
		REQUIRE((int)g2v.size() == N2 && H2.rows() == N2);
		double dn = 0;
		for (auto &v : d2)
			v = (double)std::rand() / RAND_MAX - 0.5, dn += v * v;
		for (auto &v : d2)
			v /= std::sqrt(dn);
		const double h2 = 1e-6;
		for (int i = 0; i < N2; ++i)
			xp2[i] = x2[i] + h2 * d2[i], xm2[i] = x2[i] - h2 * d2[i];
		const double fd2 = (nl2.assemble(false, b2, g2, cache, 0.0, xp2, xp2) - nl2.assemble(false, b2, g2, cache, 0.0, xm2, xm2)) / (2 * h2);
		double gd2 = 0, gn = 0;
		for (int i = 0; i < N2; ++i)
			gd2 += g2v[i] * d2[i], gn += g2v[i] * g2v[i];
		REQUIRE(std::fabs(fd2 - gd2) <= 1e-6 * std::fmax(1.0, std::sqrt(gn)));
		nl2.assemble_grad(false, nb2, b2, g2, cache, 0.0, xp2, xp2, gp);
		nl2.assemble_grad(false, nb2, b2, g2, cache, 0.0, xm2, xm2, gm);
		double err = 0, ref = 0;
		for (int j = 0; j < N2; ++j)
		{
			double hd = 0; // H2 is symmetric: column j == row j
			for (int64_t k = H2.outer[j]; k < H2.outer[j + 1]; ++k)
				hd += H2.values[k] * d2[H2.inner[k]];
			const double f = (gp[j] - gm[j]) / (2 * h2);
			err += (f - hd) * (f - hd), ref += hd * hd;
		}
		REQUIRE(std::sqrt(err) <= 1e-6 * std::sqrt(ref));
		// the Newton direction of that state through the LinearSolver, x = 0 edge fixed
		auto s2 = polysolve::LinearSolver::create("Eigen::ConjugateGradient", "Eigen::DiagonalPreconditioner");
		s2->setParameters(2000, 1e-12);
		std::vector<int> fixed2;
		for (int i = 0; i < nb2; ++i)
			if (i % 13 == 0) // lattice x == 0 (13 nodes wide for n = 6)
				fixed2.push_back(2 * i), fixed2.push_back(2 * i + 1);
		s2->setDirichlet(fixed2);
		s2->analyzePattern(H2, H2.rows());
		s2->factorize(H2);
		VectorXd r2(N2), dx2;
		for (int i = 0; i < N2; ++i)
			r2[i] = -g2v[i];
		s2->solve(r2, dx2);
		REQUIRE(s2->solver_error() < 1e-10 && s2->solver_iter() > 0 && s2->uses_resident_matrix());
		// wrong size() for the mesh is refused, not computed somewhere else
		bool threw = false;
		try
		{
			nl2.assemble(true, b2, g2, cache, 0.0, x2, x2);
		}
		catch (const std::runtime_error &)
		{
			threw = true;
		}
		REQUIRE(threw);
		// size 2 of the tensor models: HookeLinearElasticity with the isotropic 3 x 3 tensor acts like LinearElasticity,
		// SaintVenant with that tensor and MultiModel with SaintVenant everywhere are SaintVenant with (lambda, mu)
		{
			auto apply = [&](const StiffnessMatrix &A, const VectorXd &v) {
				VectorXd r(N2, 0.0);
				for (int j = 0; j < N2; ++j)
					for (int64_t k = A.outer[j]; k < A.outer[j + 1]; ++k)
						r[j] += A.values[k] * v[A.inner[k]];
				return r;
			};
			const double lam2 = lin2.local_assembler().lambda, mu2 = lin2.local_assembler().mu;
			assembler::Assembler<assembler::HookeLinearElasticity> hooke2;
			hooke2.local_assembler().set_size(2);
			hooke2.local_assembler().elasticity_tensor().set_from_lambda_mu(lam2, mu2);
			StiffnessMatrix Kh2;
			hooke2.assemble(false, nb2, b2, g2, cache, Kh2);
			REQUIRE(Kh2.rows() == N2);
			const VectorXd a = apply(K2, d2), b = apply(Kh2, d2);
			double da = 0, na = 0;
			for (int i = 0; i < N2; ++i)
				da = std::fmax(da, std::fabs(a[i] - b[i])), na = std::fmax(na, std::fabs(a[i]));
			REQUIRE(da <= 1e-12 * na);
			assembler::NLAssembler<assembler::SaintVenantElasticity> sv2, svt2;
			sv2.local_assembler().set_size(2);
			sv2.local_assembler().set_lambda_mu(lam2, mu2);
			svt2.local_assembler().set_size(2);
			svt2.local_assembler().elasticity_tensor().set_from_lambda_mu(lam2, mu2);
			const double e_sv = sv2.assemble(false, b2, g2, cache, 0.0, x2, x2);
			const double e_svt = svt2.assemble(false, b2, g2, cache, 0.0, x2, x2);
			REQUIRE(e_sv > 0 && std::fabs(e_sv - e_svt) <= 1e-12 * e_sv);
			assembler::NLAssembler<assembler::MultiModel> mm2;
			mm2.local_assembler().set_size(2);
			mm2.local_assembler().set_lambda_mu(lam2, mu2);
			mm2.local_assembler().init_multimodels(std::vector<std::string>(b2.size(), "SaintVenant"));
			const double e_mm = mm2.assemble(false, b2, g2, cache, 0.0, x2, x2);
			REQUIRE(std::fabs(e_sv - e_mm) <= 1e-12 * e_sv);
			assembler::ElasticityTensor t2;
			t2.resize(2);
			t2.set_from_entries({2.0, 4.0, 0.5, 8.0}); // Ex Ey nuYX muXY
			REQUIRE(t2(0, 0) == 0.5 && t2(1, 0) == -0.125 && t2(1, 1) == 0.25 && t2(2, 2) == 0.125 && t2(0, 2) == 0);
		}
		std::printf("2-D ok: %d dofs, cg %d iterations\n", N2, s2->solver_iter());
	}
	std::printf("shim ok: nnz(K)=%ld energy=%.6e cg_iters=%d err=%.2e\n", (long)K.nonZeros(), e0, solver->solver_iter(), solver->solver_error());
	return 0;
}
