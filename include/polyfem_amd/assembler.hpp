// polyfem_amd/assembler.hpp -- C++ host shim with the reference's own signatures on top of the C ABI
// (include/polyfem_amd.h).  A PolyFEM build would include this instead of
//   src/polyfem/assembler/Assembler.hpp            (Assembler<L>::assemble :32-39, NLAssembler<L> :80-113)
//   src/polyfem/utils/MatrixUtils.hpp              (SpareMatrixCache :39-100)
//   src/polyfem/solver/SparseNewtonDescentSolver.hpp + NLProblem.hpp (minimize on the reduced unknowns)
//   polysolve/LinearSolver.hpp                     (create / setParameters / analyzePattern / factorize / solve / getInfo;
//                                                   call sites src/polyfem/solver/SparseNewtonDescentSolver.tpp:128-155)
// and keep its call sites unchanged.  Eigen is not available in this repository, so the few Eigen types that
// appear in those signatures are replaced by the minimal stand-ins below (same storage conventions: column
// vectors are contiguous doubles, StiffnessMatrix is compressed storage with int indices); in a real PolyFEM build
// they are typedefs of the Eigen types and the bodies stay the same.
//
// Error behaviour follows the reference: failures surface as std::runtime_error (the Newton solver catches exactly
// that around factorize/solve, SparseNewtonDescentSolver.tpp:135-150).  Element sets the engine does not implement
// (non-conforming bases with global().size() != 1, mixed element types) are rejected with std::runtime_error
// instead of being computed somewhere else: there is no CPU fallback.
#pragma once

#include "../polyfem_amd.h"

#include <cmath>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

namespace polyfem_amd
{
	// ---- stand-ins for the Eigen / basis types that appear in the signatures -------------------------------------
	using VectorXd = std::vector<double>; // Eigen::MatrixXd column, x[node*dim + c]

	struct StiffnessMatrix // Eigen::SparseMatrix<double, ColMajor, int> in compressed form (utils/Types.hpp:16-20)
	{
		int n = 0;
		std::vector<int64_t> outer; // column offsets (n + 1); the wide index type of POLYSOLVE_LARGE_INDEX builds
		std::vector<int32_t> inner; // row of every entry, ascending inside a column
		std::vector<double> values;
		int rows() const { return n; }
		int cols() const { return n; }
		int64_t nonZeros() const { return (int64_t)inner.size(); }

		// Tag of a matrix that an assembler of this shim has just produced: the context that holds the same matrix on the
		// device and that context's matrix generation at the time.  polysolve::LinearSolver solves a tagged, unchanged
		// matrix in place (no values cross PCIe); everything else is uploaded.  Whoever writes to `values` / changes the
		// pattern (hessian += other forms, regularisation ...) must drop the tag: the helpers below do, direct writes call
		// untag().  In a PolyFEM build, where this type is the Eigen matrix itself, the tag travels beside it.
		pfx_ctx *resident = nullptr;
		int64_t resident_generation = -1;
		void untag() { resident = nullptr, resident_generation = -1; }

		// hessian += w * I or a diagonal form (SparseNewtonDescentSolver.tpp:116-121; lumped inertia / penalty terms).
		// Every diagonal entry must be part of the pattern, as it is for every matrix the assemblers produce.
		void add_to_diagonal(const std::vector<double> &d)
		{
			if ((int)d.size() != n)
				throw std::runtime_error("polyfem_amd: add_to_diagonal needs one value per row");
			for (int j = 0; j < n; ++j)
			{
				int64_t k = outer[(size_t)j];
				while (k < outer[(size_t)j + 1] && inner[(size_t)k] != j)
					++k;
				if (k == outer[(size_t)j + 1])
					throw std::runtime_error("polyfem_amd: add_to_diagonal: diagonal entry missing from the pattern");
				values[(size_t)k] += d[(size_t)j];
			}
			untag();
		}
		void add_to_diagonal(double w) { add_to_diagonal(std::vector<double>((size_t)n, w)); }
	};

	namespace basis
	{
		struct Local2Global // basis/Basis.hpp:21-38
		{
			int index = 0;
			double val = 1.0;
			double node[3] = {0, 0, 0};
		};
		class Basis // basis/Basis.hpp:43-118 (only what the assembler reads)
		{
		public:
			std::vector<Local2Global> global_;
			const std::vector<Local2Global> &global() const { return global_; }
		};
		class ElementBases // basis/ElementBases.hpp:16-104
		{
		public:
			std::vector<Basis> bases;
			bool has_parameterization = true;
		};
	} // namespace basis

	namespace utils
	{
		// The caller-owned pattern cache of NLAssembler::assemble_hessian.  The pattern itself lives on the device;
		// this object keeps the host copy of the index arrays that are handed back inside the StiffnessMatrix.
		class SpareMatrixCache
		{
		public:
			std::vector<int64_t> outer_index_;
			std::vector<int32_t> inner_index_;
			size_t non_zeros() const { return inner_index_.size(); }
			size_t mapping_size() const { return outer_index_.empty() ? 0 : outer_index_.size() - 1; }
		};
	} // namespace utils

	// ---- one device context per process ----------------------------------------------------------------------------
	class Context
	{
	public:
		static Context &instance(int device = 0)
		{
			static Context c(device);
			return c;
		}
		pfx_ctx *h = nullptr;
		const void *bound_bases = nullptr;
		size_t bound_count = 0;
		int bound_model = -1, bound_bs = 0;
		long pattern_generation = 0; // bumped whenever the device pattern is rebuilt (per-pattern device state must follow)
		double bound_lambda = 0, bound_mu = 0;
		const double *bound_lambda_e = nullptr;
		const void *bound_models = nullptr; // MultiModel: the per-element model list that is bound ...
		long bound_models_version = 0;      // ... and its edit counter (the list may be refilled in place)

		void check(pfx_status rc) const
		{
			if (rc != PFX_OK)
				throw std::runtime_error(std::string("polyfem_amd: ") + pfx_last_error(h));
		}

		// Flattens bases/gbases the way State::sending_data_to_GPU walks them (state/StateSolveNonlinear.cpp:224-240)
		// when the mesh changed, passes the material when it changed and (re)builds the device pattern when needed.
		// lambda_e / mu_e: one value per element (multi-material bodies) or null.
		void bind(const std::vector<basis::ElementBases> &bases, const std::vector<basis::ElementBases> &gbases, int n_basis, int model, int dim, double lambda, double mu,
				  const std::vector<double> *lambda_e = nullptr, const std::vector<double> *mu_e = nullptr, const std::vector<int32_t> *models = nullptr,
				  long models_version = 0)
		{
			const bool per_element = lambda_e && !lambda_e->empty();
			const bool same_mesh = bound_bases == (const void *)bases.data() && bound_count == bases.size();
			const bool same_material = same_mesh && bound_model == model && bound_lambda == lambda && bound_mu == mu && bound_lambda_e == (per_element ? lambda_e->data() : nullptr)
									   && bound_models == (models ? (const void *)models->data() : nullptr) && bound_models_version == models_version;
			if (same_material)
				return;
			if (!same_mesh)
				set_mesh(bases, gbases, n_basis);
			if (per_element)
			{
				if (!mu_e || lambda_e->size() != bases.size() || mu_e->size() != bases.size())
					throw std::runtime_error("polyfem_amd: per-element materials need one (lambda, mu) per element");
				check(pfx_set_material_per_element(h, model, dim, lambda_e->data(), mu_e->data(), (int64_t)bases.size()));
			}
			else
				check(pfx_set_material(h, model, dim, lambda, mu));
			if (models)
				check(pfx_set_element_models(h, models->data(), (int64_t)models->size()));
			const int bs = model == PFX_LAPLACIAN ? 1 : dim;
			if (!same_mesh || bs != bound_bs)
			{
				int64_t nnz = 0;
				check(pfx_build_pattern(h, &nnz)); // once per mesh and block size; reused by every assembly afterwards
				++pattern_generation;
			}
			bound_bases = bases.data();
			bound_count = bases.size();
			bound_model = model;
			bound_bs = bs;
			bound_lambda = lambda;
			bound_mu = mu;
			bound_lambda_e = per_element ? lambda_e->data() : nullptr;
			bound_models = models ? (const void *)models->data() : nullptr;
			bound_models_version = models_version;
		}

	private:
		void set_mesh(const std::vector<basis::ElementBases> &bases, const std::vector<basis::ElementBases> &gbases, int n_basis)
		{
			if (bases.empty() || bases.size() != gbases.size())
				throw std::runtime_error("polyfem_amd: empty or inconsistent bases");
			const int n_loc = (int)bases[0].bases.size(), n_gloc = (int)gbases[0].bases.size();
			int kind = -1;
			if (n_loc == 4 && n_gloc == 4)
				kind = PFX_P1TET;
			else if (n_loc == 10 && n_gloc == 4)
				kind = PFX_P2TET;
			else if (n_loc == 8 && n_gloc == 8)
				kind = PFX_Q1HEX;
			else if (n_loc == 27 && n_gloc == 8)
				kind = PFX_Q2HEX;
			else if (n_loc == 3 && n_gloc == 3)
				kind = PFX_P1TRI;
			else if (n_loc == 6 && n_gloc == 3)
				kind = PFX_P2TRI;
			else
				throw std::runtime_error("polyfem_amd: unsupported element (P1/P2 tets, Q1/Q2 hexes and P1/P2 triangles on linear geometry only)");
			const int gdim = (n_gloc == 3) ? 2 : 3; // coordinates per geometry node
			const int64_t n_el = (int64_t)bases.size();
			std::vector<int32_t> conn((size_t)n_el * n_loc), gconn((size_t)n_el * n_gloc);
			int64_t n_gnodes = 0;
			for (int64_t e = 0; e < n_el; ++e)
			{
				if ((int)bases[e].bases.size() != n_loc || (int)gbases[e].bases.size() != n_gloc || !gbases[e].has_parameterization)
					throw std::runtime_error("polyfem_amd: mixed element types are not supported");
				for (int j = 0; j < n_loc; ++j)
				{
					const auto &g = bases[e].bases[j].global();
					if (g.size() != 1 || g[0].val != 1.0)
						throw std::runtime_error("polyfem_amd: non-conforming bases (global().size() != 1) are not supported");
					conn[e * n_loc + j] = g[0].index;
				}
				for (int j = 0; j < n_gloc; ++j)
				{
					const auto &g = gbases[e].bases[j].global();
					if (g.size() != 1 || g[0].val != 1.0)
						throw std::runtime_error("polyfem_amd: non-conforming geometry bases are not supported");
					gconn[e * n_gloc + j] = g[0].index;
					if (g[0].index + 1 > n_gnodes)
						n_gnodes = g[0].index + 1;
				}
			}
			std::vector<double> xyz((size_t)n_gnodes * gdim, 0.0);
			for (int64_t e = 0; e < n_el; ++e)
				for (int j = 0; j < n_gloc; ++j)
				{
					const auto &g = gbases[e].bases[j].global()[0];
					for (int c = 0; c < gdim; ++c)
						xyz[(size_t)g.index * gdim + c] = g.node[c];
				}
			check(pfx_set_mesh(h, kind, n_el, n_basis, n_gnodes, conn.data(), gconn.data(), xyz.data(), -1));
			bound_bs = 0;
		}

		explicit Context(int device)
		{
			if (pfx_create(&h, device) != PFX_OK)
				throw std::runtime_error("polyfem_amd: no usable HIP device (the engine has no CPU fallback)");
		}
		~Context() { pfx_destroy(h); }
	};

	namespace assembler
	{
		class AssemblyValsCache // assembler/AssemblyValsCache.hpp: the engine recomputes per-element values on the fly
		{
		};

		// local assemblers: only the parameters the reference objects carry (set_size / add_multimaterial)
		struct Laplacian
		{
			static constexpr int model = PFX_LAPLACIAN;
			int size() const { return 1; }
			double lambda = 1, mu = 1;
			std::vector<double> lambda_e, mu_e; // unused
		};
		struct ElasticBase
		{
			int size_ = 3;
			double lambda = 1, mu = 1;
			int size() const { return size_; }
			void set_size(int s)
			{
				if (s != 3 && s != 2)
					throw std::runtime_error("polyfem_amd: elasticity is implemented for size 2 (triangles) and 3");
				size_ = s;
			}
			// LameParameters::add_multimaterial for uniform (E, nu): utils/ElasticityUtils.cpp:203-214 (convert_to_lambda with
			// is_volume = size() == 3); call after set_size
			void set_E_nu(double E, double nu)
			{
				lambda = size_ == 3 ? (E * nu) / ((1.0 + nu) * (1.0 - 2.0 * nu)) : (nu * E) / (1.0 - nu * nu);
				mu = E / (2.0 * (1.0 + nu));
			}
			void set_lambda_mu(double l, double m) { lambda = l, mu = m; }
			// LameParameters with one entry per element id, what add_multimaterial builds for multi-body meshes
			// (utils/ElasticityUtils.cpp:490-560)
			std::vector<double> lambda_e, mu_e;
			void set_lambda_mu_per_element(const std::vector<double> &l, const std::vector<double> &m) { lambda_e = l, mu_e = m; }
		};
		struct LinearElasticity : ElasticBase
		{
			static constexpr int model = PFX_LINEAR_ELASTICITY;
		};
		struct NeoHookeanElasticity : ElasticBase
		{
			static constexpr int model = PFX_NEOHOOKEAN;
		};
		// utils/ElasticityUtils.hpp ElasticityTensor: 6 x 6 for size 3, 3 x 3 (xx yy xy) for size 2 (resize,
		// ElasticityUtils.cpp:264-274); only the upper triangle is ever written or read (operator() swaps its indices,
		// :276-296), so full() mirrors it.
		class ElasticityTensor
		{
		public:
			void resize(const int size)
			{
				if (size != 2 && size != 3)
					throw std::runtime_error("polyfem_amd: the elasticity tensor is defined for size 2 and 3");
				size_ = size;
				for (double &v : upper_)
					v = 0;
			}
			int size() const { return size_; }
			double &operator()(int i, int j) { return i <= j ? upper_[i * 6 + j] : upper_[j * 6 + i]; }
			double operator()(int i, int j) const { return i <= j ? upper_[i * 6 + j] : upper_[j * 6 + i]; }
			void set_from_entries(const std::vector<double> &entries) // ElasticityUtils.cpp:298-370
			{
				if (size_ == 2)
				{
					if (entries.size() == 4)
					{
						set_orthotropic(entries[0], entries[1], entries[2], entries[3]);
						return;
					}
					if (entries.size() < 6)
						throw std::runtime_error("polyfem_amd: the 2-D elasticity tensor takes 6 (or 4 orthotropic) entries");
					int k = 0;
					for (int i = 0; i < 3; ++i)
						for (int j = i; j < 3; ++j)
							(*this)(i, j) = entries[k++];
					return;
				}
				if (entries.size() == 9)
				{
					set_orthotropic(entries[0], entries[1], entries[2], entries[3], entries[4], entries[5], entries[6], entries[7], entries[8]);
					return;
				}
				if (entries.size() < 21)
					throw std::runtime_error("polyfem_amd: the elasticity tensor takes 21 (or 9 orthotropic) entries");
				int k = 0;
				for (int i = 0; i < 6; ++i)
					for (int j = i; j < 6; ++j)
						(*this)(i, j) = entries[k++];
			}
			void set_from_lambda_mu(const double lambda, const double mu) // ElasticityUtils.cpp:372-413
			{
				for (double &v : upper_)
					v = 0;
				if (size_ == 2)
				{
					(*this)(0, 0) = (*this)(1, 1) = 2 * mu + lambda;
					(*this)(0, 1) = lambda;
					(*this)(2, 2) = mu;
					return;
				}
				for (int i = 0; i < 3; ++i)
				{
					(*this)(i, i) = 2 * mu + lambda;
					for (int j = i + 1; j < 3; ++j)
						(*this)(i, j) = lambda;
					(*this)(3 + i, 3 + i) = mu;
				}
			}
			void set_from_young_poisson(const double young, const double nu) // ElasticityUtils.cpp:416-437
			{
				for (double &x : upper_)
					x = 0;
				if (size_ == 2)
				{
					const double f2 = young / (1.0 - nu * nu);
					(*this)(0, 0) = (*this)(1, 1) = 1.0 * f2;
					(*this)(0, 1) = nu * f2;
					(*this)(2, 2) = ((1.0 - nu) / 2.0) * f2;
					return;
				}
				const double v = nu, f = young / ((1. + v) * (1. - 2. * v));
				for (int i = 0; i < 3; ++i)
				{
					(*this)(i, i) = (1. - v) * f;
					for (int j = i + 1; j < 3; ++j)
						(*this)(i, j) = v * f;
					(*this)(3 + i, 3 + i) = ((1. - 2. * v) / 2.) * f;
				}
			}
			// ElasticityUtils.cpp:439-455: the reference stores these reciprocals as they are (no inversion follows)
			void set_orthotropic(double Ex, double Ey, double Ez, double nuYX, double nuZX, double nuZY, double muYZ, double muZX, double muXY)
			{
				for (double &x : upper_)
					x = 0;
				(*this)(0, 0) = 1.0 / Ex, (*this)(0, 1) = -nuYX / Ey, (*this)(0, 2) = -nuZX / Ez;
				(*this)(1, 1) = 1.0 / Ey, (*this)(1, 2) = -nuZY / Ez;
				(*this)(2, 2) = 1.0 / Ez;
				(*this)(3, 3) = 1.0 / muYZ, (*this)(4, 4) = 1.0 / muZX, (*this)(5, 5) = 1.0 / muXY;
			}
			// ElasticityUtils.cpp:457-466 (size 2)
			void set_orthotropic(double Ex, double Ey, double nuYX, double muXY)
			{
				for (double &x : upper_)
					x = 0;
				(*this)(0, 0) = 1.0 / Ex, (*this)(0, 1) = -nuYX / Ey;
				(*this)(1, 1) = 1.0 / Ey;
				(*this)(2, 2) = 1.0 / muXY;
			}
			// the symmetric n x n matrix, row-major: n = 6 for size 3, 3 for size 2
			void full(double (&out)[36]) const
			{
				const int n = size_ == 2 ? 3 : 6;
				for (int i = 0; i < n; ++i)
					for (int j = 0; j < n; ++j)
						out[i * n + j] = (*this)(i, j);
			}

		private:
			int size_ = 3;
			double upper_[36] = {};
		};
		// assembler/SaintVenantElasticity.hpp: (lambda, mu) / (young, nu), or the material's "elasticity_tensor" entries
		// (add_multimaterial, SaintVenantElasticity.cpp:42-69): calling elasticity_tensor() selects the tensor form
		struct SaintVenantElasticity : ElasticBase
		{
			static constexpr int model = PFX_SAINT_VENANT;
			void set_size(int s) // SaintVenantElasticity.cpp:71-75
			{
				ElasticBase::set_size(s);
				tensor_.resize(s);
			}
			ElasticityTensor &elasticity_tensor()
			{
				has_tensor = true;
				return tensor_;
			}
			const ElasticityTensor &elasticity_tensor() const { return tensor_; }
			bool has_tensor = false;

		private:
			ElasticityTensor tensor_;
		};

		// assembler/HookeLinearElasticity.hpp: linear assembler with a general elasticity tensor ("elasticity_tensor"
		// entries, else young/nu, else lambda/mu: HookeLinearElasticity.cpp:24-49)
		struct HookeLinearElasticity
		{
			static constexpr int model = PFX_HOOKE_LINEAR;
			int size() const { return tensor_.size(); }
			void set_size(int s) { tensor_.resize(s); } // HookeLinearElasticity.cpp:69-73
			ElasticityTensor &elasticity_tensor() { return tensor_; }
			const ElasticityTensor &elasticity_tensor() const { return tensor_; }
			double lambda = 0, mu = 0;
			std::vector<double> lambda_e, mu_e; // unused

		private:
			ElasticityTensor tensor_;
		};

		// assembler/MultiModel.hpp: one of the three elastic models per element ("multi_material_models_[el_id]",
		// MultiModel.cpp:42-96); the Lame parameters are shared by the member assemblers (add_multimaterial, :22-29)
		struct MultiModel : ElasticBase
		{
			static constexpr int model = PFX_NEOHOOKEAN; // what pfx_set_material receives before pfx_set_element_models
			void init_multimodels(const std::vector<std::string> &mats)
			{
				models.clear();
				++version;
				for (const std::string &m : mats)
				{
					if (m == "LinearElasticity")
						models.push_back(PFX_LINEAR_ELASTICITY);
					else if (m == "NeoHookean")
						models.push_back(PFX_NEOHOOKEAN);
					else if (m == "SaintVenant")
						models.push_back(PFX_SAINT_VENANT);
					else
						throw std::runtime_error("polyfem_amd: MultiModel knows LinearElasticity, NeoHookean and SaintVenant, not " + m);
				}
			}
			std::vector<int32_t> models;
			long version = 0;
		};

		// assembler/Mass.hpp: rho phi_i phi_j on the diagonal of every size x size block; density uniform or one value
		// per element (Density with one entry per body id, utils/ElasticityUtils.cpp:561-570)
		struct Mass
		{
			int size_ = 3;
			double density = 1.0;
			std::vector<double> density_e;
			int size() const { return size_; }
			void set_size(int s)
			{
				if (s != 1 && s != 2 && s != 3)
					throw std::runtime_error("polyfem_amd: the mass matrix is implemented for size 1, 2 and 3");
				size_ = s;
			}
			// the block size decides the layout of the device pattern: bound like the stiffness of the same size
			int model_for_binding() const { return size_ == 1 ? PFX_LAPLACIAN : PFX_LINEAR_ELASTICITY; }
			double lambda = 1, mu = 1;
			std::vector<double> lambda_e, mu_e; // unused
		};

		template <class LocalAssembler>
		class Assembler // assembler/Assembler.hpp:26-47
		{
		public:
			void assemble(const bool is_volume, const int n_basis, const std::vector<basis::ElementBases> &bases,
						  const std::vector<basis::ElementBases> &gbases, const AssemblyValsCache &, StiffnessMatrix &stiffness,
						  const bool is_mass = false) const
			{
				// is_volume = the mesh is 3-D (State::assemble_stiffness_mat, State.cpp:1091); otherwise a triangle mesh, size() == 2
				const int dim = is_volume ? 3 : 2;
				if (local_assembler_.size() != 1 && local_assembler_.size() != dim)
					throw std::runtime_error("polyfem_amd: the local assembler's size() does not match is_volume");
				Context &c = Context::instance();
				if constexpr (std::is_same<LocalAssembler, Mass>::value)
				{
					if (!is_mass)
						throw std::runtime_error("polyfem_amd: Assembler<Mass>::assemble is called with is_mass = true");
					if (!local_assembler_.density_e.empty() && local_assembler_.density_e.size() != bases.size())
						throw std::runtime_error("polyfem_amd: one density per element expected");
					c.bind(bases, gbases, n_basis, local_assembler_.model_for_binding(), dim, 1.0, 1.0);
					c.check(pfx_assemble_mass(c.h, local_assembler_.density, local_assembler_.density_e.empty() ? nullptr : local_assembler_.density_e.data()));
				}
				else if constexpr (std::is_same<LocalAssembler, HookeLinearElasticity>::value)
				{
					if (is_mass)
						throw std::runtime_error("polyfem_amd: is_mass = true belongs to Assembler<Mass>");
					c.bind(bases, gbases, n_basis, PFX_LINEAR_ELASTICITY, dim, 0.0, 0.0); // mesh and pattern of block size dim
					double t[36];
					local_assembler_.elasticity_tensor().full(t);
					c.check(pfx_set_elasticity_tensor(c.h, dim, t));
					c.bound_model = -1; // the next bind passes its own material again
					c.check(pfx_assemble_linear(c.h));
				}
				else
				{
					if (is_mass)
						throw std::runtime_error("polyfem_amd: is_mass = true belongs to Assembler<Mass>");
					c.bind(bases, gbases, n_basis, LocalAssembler::model, dim, local_assembler_.lambda, local_assembler_.mu, &local_assembler_.lambda_e, &local_assembler_.mu_e);
					c.check(pfx_assemble_linear(c.h));
				}
				int64_t nnz = 0;
				c.check(pfx_get_linear_csr(c.h, &nnz, nullptr, nullptr, nullptr));
				stiffness.n = n_basis * local_assembler_.size();
				stiffness.outer.assign((size_t)stiffness.n + 1, 0);
				stiffness.inner.assign((size_t)nnz, 0);
				stiffness.values.assign((size_t)nnz, 0.0);
				c.check(pfx_get_linear_csr(c.h, &nnz, stiffness.outer.data(), stiffness.inner.data(), stiffness.values.data()));
				stiffness.resident = c.h;
				c.check(pfx_matrix_generation(c.h, &stiffness.resident_generation));
			}
			LocalAssembler &local_assembler() { return local_assembler_; }
			const LocalAssembler &local_assembler() const { return local_assembler_; }

		private:
			LocalAssembler local_assembler_;
		};

		// RhsAssembler::assemble (assembler/RhsAssembler.cpp:55-98) for a constant body force: rhs[node*3 + c] =
		// sum_e sum_q force_c rho det w phi_i(q).  The mesh must be bound (any assembler call binds it).
		inline void assemble_body_force(const std::vector<basis::ElementBases> &bases, const std::vector<basis::ElementBases> &gbases, const int n_basis,
										const double (&force)[3], const Mass &density, VectorXd &rhs)
		{
			Context &c = Context::instance();
			c.bind(bases, gbases, n_basis, PFX_LINEAR_ELASTICITY, 3, 1.0, 1.0);
			rhs.assign((size_t)n_basis * 3, 0.0);
			c.check(pfx_assemble_rhs(c.h, force, density.density, density.density_e.empty() ? nullptr : density.density_e.data(), rhs.data()));
		}

		// the same on a triangle mesh (two force components, rhs[node*2 + c])
		inline void assemble_body_force(const std::vector<basis::ElementBases> &bases, const std::vector<basis::ElementBases> &gbases, const int n_basis,
										const double (&force)[2], const Mass &density, VectorXd &rhs)
		{
			Context &c = Context::instance();
			c.bind(bases, gbases, n_basis, PFX_LINEAR_ELASTICITY, 2, 1.0, 1.0);
			rhs.assign((size_t)n_basis * 2, 0.0);
			c.check(pfx_assemble_rhs(c.h, force, density.density, density.density_e.empty() ? nullptr : density.density_e.data(), rhs.data()));
		}

		template <class LocalAssembler>
		class NLAssembler // assembler/Assembler.hpp:70-118
		{
		public:
			double assemble(const bool is_volume, const std::vector<basis::ElementBases> &bases, const std::vector<basis::ElementBases> &gbases,
							const AssemblyValsCache &, const double /*dt*/, const VectorXd &displacement, const VectorXd & /*displacement_prev*/) const
			{
				Context &c = bound(is_volume, bases, gbases, (int)displacement.size() / local_assembler_.size());
				double e = 0;
				c.check(pfx_energy(c.h, displacement.data(), &e));
				return e;
			}
			void assemble_grad(const bool is_volume, const int n_basis, const std::vector<basis::ElementBases> &bases,
							   const std::vector<basis::ElementBases> &gbases, const AssemblyValsCache &, const double /*dt*/,
							   const VectorXd &displacement, const VectorXd & /*displacement_prev*/, VectorXd &rhs) const
			{
				Context &c = bound(is_volume, bases, gbases, n_basis);
				rhs.assign((size_t)n_basis * local_assembler_.size(), 0.0);
				c.check(pfx_gradient(c.h, displacement.data(), rhs.data()));
			}
			void assemble_hessian(const bool is_volume, const int n_basis, const bool project_to_psd, const std::vector<basis::ElementBases> &bases,
								  const std::vector<basis::ElementBases> &gbases, const AssemblyValsCache &, const double /*dt*/,
								  const VectorXd &displacement, const VectorXd & /*displacement_prev*/, utils::SpareMatrixCache &mat_cache,
								  StiffnessMatrix &grad) const
			{
				Context &c = bound(is_volume, bases, gbases, n_basis);
				const int n = n_basis * local_assembler_.size();
				if (mat_cache.mapping_size() != (size_t)n)
				{
					// first call: the pattern built on the device is copied once into the caller's cache
					// (the reference fills it during its first CPU assembly, utils/MatrixUtils.cpp:171-241)
					int64_t nnz = 0;
					pfx_info info;
					c.check(pfx_get_info(c.h, &info));
					nnz = info.nnz;
					mat_cache.outer_index_.assign((size_t)n + 1, 0);
					mat_cache.inner_index_.assign((size_t)nnz, 0);
					c.check(pfx_get_pattern(c.h, mat_cache.outer_index_.data(), mat_cache.inner_index_.data()));
				}
				grad.n = n;
				grad.outer = mat_cache.outer_index_;
				grad.inner = mat_cache.inner_index_;
				grad.values.assign(mat_cache.inner_index_.size(), 0.0);
				c.check(pfx_hessian(c.h, displacement.data(), project_to_psd ? 1 : 0, grad.values.data()));
				grad.resident = c.h;
				c.check(pfx_matrix_generation(c.h, &grad.resident_generation));
			}
			LocalAssembler &local_assembler() { return local_assembler_; }

		private:
			Context &bound(bool is_volume, const std::vector<basis::ElementBases> &bases, const std::vector<basis::ElementBases> &gbases, int n_basis) const
			{
				const int dim = is_volume ? 3 : 2;
				if (local_assembler_.size() != dim)
					throw std::runtime_error("polyfem_amd: the local assembler's size() does not match is_volume");
				Context &c = Context::instance();
				if constexpr (std::is_same<LocalAssembler, MultiModel>::value)
				{
					if (local_assembler_.models.size() != bases.size())
						throw std::runtime_error("polyfem_amd: MultiModel needs one model per element (init_multimodels)");
					c.bind(bases, gbases, n_basis, LocalAssembler::model, dim, local_assembler_.lambda, local_assembler_.mu, &local_assembler_.lambda_e, &local_assembler_.mu_e,
						   &local_assembler_.models, local_assembler_.version);
				}
				else
					c.bind(bases, gbases, n_basis, LocalAssembler::model, dim, local_assembler_.lambda, local_assembler_.mu, &local_assembler_.lambda_e, &local_assembler_.mu_e);
				if constexpr (std::is_same<LocalAssembler, SaintVenantElasticity>::value)
					if (local_assembler_.has_tensor)
					{
						double t[36];
						local_assembler_.elasticity_tensor().full(t);
						c.check(pfx_set_saint_venant_tensor(c.h, dim, t));
						c.bound_model = -1; // the next bind passes its own material again
					}
				return c;
			}
			LocalAssembler local_assembler_;
		};
	} // namespace assembler

	namespace polysolve
	{
		// polysolve::LinearSolver for "Eigen::ConjugateGradient" (and, in the slot of "AMGCL" / "Hypre", CG with this
		// library's AMG).  analyzePattern / factorize / solve work on the matrix they are HANDED -- the reduced sum of all
		// forms' Hessians at solver/SparseNewtonDescentSolver.tpp:128-155 (built at solver/FullNLProblem.cpp:115-135 and
		// solver/NLProblem.cpp:92-99), the stiffness matrix at state/StateSolveLinear.cpp:27-29 -- through a solver context
		// of its own (pfx_solver_analyze64 / pfx_solver_factorize).  Only a matrix that still carries the tag of the
		// assembling context and whose device copy is unchanged (StiffnessMatrix::resident) is solved in place.
		class LinearSolver
		{
		public:
			static std::unique_ptr<LinearSolver> create(const std::string &solver, const std::string &precond)
			{
				const bool amg = solver == "AMGCL" || solver == "Hypre";
				if (!solver.empty() && solver != "Eigen::ConjugateGradient" && !amg)
					throw std::runtime_error("Unrecognized solver type: " + solver);
				auto s = std::unique_ptr<LinearSolver>(new LinearSolver());
				s->name_ = solver.empty() ? "Eigen::ConjugateGradient" : solver;
				pfx_amg_default_params(&s->amg_params_);
				if (amg)
				{
					// the slot of polysolve's AMG solvers (CMakeLists.txt:101-102): CG preconditioned by this library's own
					// aggregation AMG, with their defaults tol 1e-10 / 1000 iterations (input-spec.json:1381-1396, 1423-1434)
					s->precond_ = PFX_PRECOND_AMG;
					s->tolerance_ = 1e-10;
				}
				else
					s->precond_ = (precond.empty() || precond == "Eigen::DiagonalPreconditioner") ? PFX_PRECOND_JACOBI : PFX_PRECOND_IDENTITY;
				return s;
			}
			~LinearSolver()
			{
				if (own_)
					pfx_destroy(own_);
			}
			LinearSolver(const LinearSolver &) = delete;
			LinearSolver &operator=(const LinearSolver &) = delete;
			std::string name() const { return name_; }
			// AMGCL/precond/{ncycle, max_levels, relax/degree} (input-spec.json:1447-1496)
			void setAmgParameters(int ncycle, int max_levels, int relax_degree)
			{
				amg_params_.ncycle = ncycle, amg_params_.max_levels = max_levels, amg_params_.relax_degree = relax_degree;
				amg_params_set_ = true;
				Context &c = Context::instance();
				c.check(pfx_amg_set_params(c.h, &amg_params_));
				if (own_)
					check_own(pfx_amg_set_params(own_, &amg_params_));
			}
			// setParameters(json): params[name()]["max_iter"|"tolerance"], defaults input-spec.json:1326-1336
			void setParameters(int max_iter, double tolerance)
			{
				max_iter_ = max_iter;
				tolerance_ = tolerance;
			}
			// 0: 3 x 3 node blocks where the pattern allows (default), 1: scalar format, 3: blocks
			void setBlockSize(int bs) { block_size_ = bs; }
			// positions of the nodes (rows / 3) of a handed-in matrix: the AMG preconditioner's rigid-body modes need them
			void setNodePositions(const std::vector<double> &xyz) { positions_ = xyz; }

			void analyzePattern(const StiffnessMatrix &A, const int /*precond_num*/)
			{
				if (A.rows() <= 0 || (int64_t)A.outer.size() != (int64_t)A.rows() + 1)
					throw std::runtime_error("polyfem_amd: analyzePattern needs a compressed matrix");
				n_ = A.rows();
				nnz_ = A.nonZeros();
				if (is_resident(A))
				{
					active_ = A.resident;
					resident_ = true;
					return;
				}
				resident_ = false;
				if (!own_)
				{
					if (pfx_create(&own_, 0) != PFX_OK)
						throw std::runtime_error("polyfem_amd: no usable HIP device (the engine has no CPU fallback)");
					if (amg_params_set_)
						check_own(pfx_amg_set_params(own_, &amg_params_));
				}
				check_own(pfx_solver_analyze64(own_, A.rows(), A.outer.data(), A.inner.data(), block_size_));
				if (!positions_.empty())
					check_own(pfx_solver_set_node_positions(own_, positions_.data(), (int64_t)positions_.size() / 3));
				active_ = own_;
			}
			// throws std::runtime_error when the matrix cannot be solved with (non-finite entries, a diagonal entry that is not
			// positive): the Newton solver catches exactly that and regularises (SparseNewtonDescentSolver.tpp:135-150)
			void factorize(const StiffnessMatrix &A)
			{
				if (!active_)
					throw std::runtime_error("polyfem_amd: factorize called before analyzePattern");
				if (resident_)
				{
					if (is_resident(A) && A.rows() == n_)
						return;
					analyzePattern(A, 0); // the caller changed the matrix since: it is no longer the device copy
				}
				if (A.rows() != n_ || A.nonZeros() != nnz_)
					throw std::runtime_error("polyfem_amd: factorize called with a matrix of another pattern than analyzePattern");
				check_own(pfx_solver_factorize(own_, A.values.data()));
			}
			// full-size solves on the resident matrix: the constrained dofs are given once and the solver works on the masked
			// system (the reduced matrix is never formed)
			void setDirichlet(const std::vector<int> &sorted_dofs)
			{
				pfx_ctx *h = active_ ? active_ : Context::instance().h;
				check(h, pfx_set_dirichlet(h, sorted_dofs.data(), (int64_t)sorted_dofs.size()));
			}
			void solve(const VectorXd &b, VectorXd &x)
			{
				if (!active_)
					throw std::runtime_error("polyfem_amd: solve called before analyzePattern / factorize");
				if ((int)b.size() != n_)
					throw std::runtime_error("polyfem_amd: right-hand side of the wrong size");
				if (x.size() != b.size())
					x.assign(b.size(), 0.0);
				check(active_, pfx_cg(active_, b.data(), x.data(), precond_, max_iter_, tolerance_, &iters_, &error_));
				if (!std::isfinite(error_))
					throw std::runtime_error("polyfem_amd: CG broke down (non-finite residual)");
			}
			// getInfo(json&): "solver_iter", "solver_error"
			int solver_iter() const { return iters_; }
			double solver_error() const { return error_; }
			bool uses_resident_matrix() const { return resident_; }

		private:
			LinearSolver() = default;
			static bool is_resident(const StiffnessMatrix &A)
			{
				if (!A.resident || A.resident != Context::instance().h)
					return false;
				int64_t gen = -1;
				return pfx_matrix_generation(A.resident, &gen) == PFX_OK && gen == A.resident_generation;
			}
			static void check(pfx_ctx *h, pfx_status rc)
			{
				if (rc != PFX_OK)
					throw std::runtime_error(std::string("polyfem_amd: ") + pfx_last_error(h));
			}
			void check_own(pfx_status rc) const { check(own_, rc); }
			int precond_ = PFX_PRECOND_JACOBI, max_iter_ = 1000, iters_ = 0, n_ = 0, block_size_ = 0;
			int64_t nnz_ = 0;
			double tolerance_ = 1e-12, error_ = 0;
			std::string name_ = "Eigen::ConjugateGradient";
			pfx_ctx *own_ = nullptr;    // solver context for matrices handed in by value
			pfx_ctx *active_ = nullptr; // what the next solve runs on
			bool resident_ = false;
			pfx_amg_params amg_params_;
			bool amg_params_set_ = false;
			std::vector<double> positions_;
		};
	} // namespace polysolve

	namespace time_integrator
	{
		// time_integrator::ImplicitEuler (time_integrator/ImplicitEuler.cpp:5-33, ImplicitTimeIntegrator.cpp:16-30): host
		// vectors, as in the reference
		class ImplicitEuler
		{
		public:
			void init(const VectorXd &x_prev, const VectorXd &v_prev, const VectorXd &a_prev, double dt)
			{
				if (!(dt > 0))
					throw std::runtime_error("polyfem_amd: dt must be positive");
				x_prev_ = x_prev, v_prev_ = v_prev, a_prev_ = a_prev, dt_ = dt;
			}
			double dt() const { return dt_; }
			const VectorXd &x_prev() const { return x_prev_; }
			const VectorXd &v_prev() const { return v_prev_; }
			const VectorXd &a_prev() const { return a_prev_; }
			VectorXd x_tilde() const
			{
				VectorXd t(x_prev_.size());
				for (size_t k = 0; k < t.size(); ++k)
					t[k] = x_prev_[k] + dt_ * v_prev_[k];
				return t;
			}
			double acceleration_scaling() const { return dt_ * dt_; }
			void update_quantities(const VectorXd &x)
			{
				for (size_t k = 0; k < x.size(); ++k)
				{
					const double v = (x[k] - x_prev_[k]) / dt_;
					a_prev_[k] = (v - v_prev_[k]) / dt_;
					v_prev_[k] = v;
					x_prev_[k] = x[k];
				}
			}

		private:
			VectorXd x_prev_, v_prev_, a_prev_;
			double dt_ = 0;
		};
	} // namespace time_integrator

	namespace solver
	{
		// solver::NLProblem (solver/NLProblem.hpp, NLProblem.cpp:79-211) for one ElasticForm plus one BodyForm: the
		// unknowns the nonlinear solver sees are the free dofs, boundary_nodes (sorted) keep the values of
		// `boundary_values` (full size).
		template <class LocalAssembler>
		class NLProblem
		{
		public:
			NLProblem(const std::vector<basis::ElementBases> &bases, const std::vector<basis::ElementBases> &gbases, int n_basis,
					  const LocalAssembler &local_assembler, const std::vector<int> &boundary_nodes, const VectorXd &rhs, const VectorXd &boundary_values)
				: bases_(bases), gbases_(gbases), n_basis_(n_basis), local_assembler_(local_assembler), boundary_nodes_(boundary_nodes), rhs_(rhs), boundary_values_(boundary_values)
			{
				is_fixed_.assign((size_t)full_size(), 0);
				for (int b : boundary_nodes_)
					is_fixed_[(size_t)b] = 1;
			}
			int full_size() const { return n_basis_ * 3; }
			int reduced_size() const { return full_size() - (int)boundary_nodes_.size(); }
			VectorXd full_to_reduced(const VectorXd &full) const
			{
				VectorXd r;
				r.reserve((size_t)reduced_size());
				for (int k = 0; k < full_size(); ++k)
					if (!is_fixed_[(size_t)k])
						r.push_back(full[(size_t)k]);
				return r;
			}
			VectorXd reduced_to_full(const VectorXd &reduced) const
			{
				VectorXd f((size_t)full_size());
				size_t w = 0;
				for (int k = 0; k < full_size(); ++k)
					f[(size_t)k] = is_fixed_[(size_t)k] ? (boundary_values_.empty() ? 0.0 : boundary_values_[(size_t)k]) : reduced[w++];
				return f;
			}
			// InertiaForm (solver/forms/InertiaForm.cpp) with the mass matrix of `mass` and the predictor of `integrator`;
			// the elastic and body forms are then weighted with its acceleration_scaling (state/StateSolveNonlinear.cpp:
			// 78-84).  nullptr: static problem.
			void set_inertia(const assembler::Mass &mass, const time_integrator::ImplicitEuler *integrator)
			{
				mass_ = mass;
				integrator_ = integrator;
				mass_generation_ = -1;
			}
			void bind()
			{
				Context &c = Context::instance();
				c.bind(bases_, gbases_, n_basis_, LocalAssembler::model, 3, local_assembler_.lambda, local_assembler_.mu, &local_assembler_.lambda_e, &local_assembler_.mu_e);
				c.check(pfx_set_dirichlet(c.h, boundary_nodes_.data(), (int64_t)boundary_nodes_.size()));
				if (!integrator_)
				{
					c.check(pfx_inertia_enable(c.h, 0.0, nullptr));
					return;
				}
				if (mass_generation_ != c.pattern_generation)
				{
					// the mass matrix lives with the device pattern: assembled once per pattern
					c.check(pfx_inertia_set_mass(c.h, mass_.density, mass_.density_e.empty() ? nullptr : mass_.density_e.data()));
					mass_generation_ = c.pattern_generation;
				}
				const VectorXd xt = integrator_->x_tilde();
				c.check(pfx_inertia_enable_host(c.h, integrator_->acceleration_scaling(), xt.data()));
			}
			const VectorXd &rhs() const { return rhs_; }

		private:
			const std::vector<basis::ElementBases> &bases_, &gbases_;
			int n_basis_;
			LocalAssembler local_assembler_;
			std::vector<int> boundary_nodes_;
			VectorXd rhs_, boundary_values_;
			std::vector<char> is_fixed_;
			assembler::Mass mass_;
			const time_integrator::ImplicitEuler *integrator_ = nullptr;
			long mass_generation_ = -1;
		};
	} // namespace solver
} // namespace polyfem_amd

namespace cppoptlib
{
	// cppoptlib::SparseNewtonDescentSolver<ProblemType> (solver/SparseNewtonDescentSolver.hpp, NonlinearSolver.tpp:33-275):
	// the whole minimisation runs on the device; solver_params carry solver/nonlinear/* of the input spec.
	template <class ProblemType>
	class SparseNewtonDescentSolver
	{
	public:
		pfx_newton_params solver_params;
		pfx_newton_info solver_info{};
		SparseNewtonDescentSolver() { pfx_newton_default_params(&solver_params); }
		std::string name() const { return "Newton"; }
		// x: the reduced unknowns, starting point on input, minimiser on output.  Throws std::runtime_error where the
		// reference does ("Reached iteration limit", "f(x) is nan or inf; stopping", "Line search failed on gradient
		// descent", ...).
		void minimize(ProblemType &objFunc, polyfem_amd::VectorXd &x)
		{
			polyfem_amd::Context &c = polyfem_amd::Context::instance();
			objFunc.bind();
			polyfem_amd::VectorXd full = objFunc.reduced_to_full(x);
			c.check(pfx_newton_solve(c.h, objFunc.rhs().empty() ? nullptr : objFunc.rhs().data(), full.data(), &solver_params, &solver_info));
			x = objFunc.full_to_reduced(full);
		}
	};
} // namespace cppoptlib
