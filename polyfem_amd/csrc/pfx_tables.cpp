// pfx_tables.cpp -- reference-element data: quadrature rules and shape-function gradients at the quadrature
// points, evaluated once on the host and uploaded (they are the same for every element of a mesh).
//
// Built with -ffp-contract=off: the linear assembler's sparsity pattern depends on exact zeros, so every
// expression is written in the operation order of the reference's generated code
//   tets : src/polyfem/autogen/auto_p_bases.cpp:1000-1078 (P1), :1152-1326 (P2)
//   hexes: src/polyfem/autogen/auto_q_bases.cpp:1115-1260 (Q1), :1277-1871 (Q2, factor order recorded in pf_q2_tables.h)
//   triangles: src/polyfem/autogen/auto_p_bases.cpp:60-136 (P1), :146-273 (P2); third components are zero
// and of the quadrature set-up
//   src/polyfem/quadrature/TetQuadrature.cpp:31-43, LineQuadrature.cpp:4707-4734, HexQuadrature.cpp:17-45,
//   TriQuadrature.cpp:30-43, order rule src/polyfem/basis/FEBasis3d.cpp:1467,1506, basis/FEBasis2d.cpp:797-798.
#include "pfx_internal.h"
#include "pf_quad_tables.h"
#include "pf_q2_tables.h"

#include <cmath>
#include <cstring>

namespace
{
	struct rule
	{
		int n = 0;
		double p[PFX_MAX_PTS][3];
		double w[PFX_MAX_PTS];
	};

	bool tet_rule(int order, rule &r)
	{
		const double *p = nullptr, *w = nullptr;
		switch (order)
		{
		case 1:
			p = pf_tet_pts_1, w = pf_tet_wts_1, r.n = PF_TET_NPTS_1;
			break;
		case 2:
			p = pf_tet_pts_2, w = pf_tet_wts_2, r.n = PF_TET_NPTS_2;
			break;
		case 3:
			p = pf_tet_pts_3, w = pf_tet_wts_3, r.n = PF_TET_NPTS_3;
			break;
		case 4:
			p = pf_tet_pts_4, w = pf_tet_wts_4, r.n = PF_TET_NPTS_4;
			break;
		case 5:
			p = pf_tet_pts_5, w = pf_tet_wts_5, r.n = PF_TET_NPTS_5;
			break;
		default:
			return false;
		}
		for (int k = 0; k < r.n; ++k)
		{
			for (int c = 0; c < 3; ++c)
				r.p[k][c] = p[3 * k + c];
			r.w[k] = w[k] / 6;
		}
		return true;
	}

	bool tri_rule(int order, rule &r)
	{
		const double *p = nullptr, *w = nullptr;
		switch (order)
		{
		case 1:
			p = pf_tri_pts_1, w = pf_tri_wts_1, r.n = PF_TRI_NPTS_1;
			break;
		case 2:
			p = pf_tri_pts_2, w = pf_tri_wts_2, r.n = PF_TRI_NPTS_2;
			break;
		case 3:
			p = pf_tri_pts_3, w = pf_tri_wts_3, r.n = PF_TRI_NPTS_3;
			break;
		case 4:
			p = pf_tri_pts_4, w = pf_tri_wts_4, r.n = PF_TRI_NPTS_4;
			break;
		case 5:
			p = pf_tri_pts_5, w = pf_tri_wts_5, r.n = PF_TRI_NPTS_5;
			break;
		default:
			return false;
		}
		for (int k = 0; k < r.n; ++k)
		{
			r.p[k][0] = p[2 * k], r.p[k][1] = p[2 * k + 1], r.p[k][2] = 0;
			r.w[k] = w[k] / 2;
		}
		return true;
	}

	bool line_rule(int n, double *x, double *w)
	{
		const double *p = nullptr, *ww = nullptr;
		switch (n)
		{
		case 1:
			p = pf_line_pts_1, ww = pf_line_wts_1;
			break;
		case 2:
			p = pf_line_pts_2, ww = pf_line_wts_2;
			break;
		case 3:
			p = pf_line_pts_3, ww = pf_line_wts_3;
			break;
		case 4:
			p = pf_line_pts_4, ww = pf_line_wts_4;
			break;
		case 5:
			p = pf_line_pts_5, ww = pf_line_wts_5;
			break;
		default:
			return false;
		}
		double sum = 0;
		for (int k = 0; k < n; ++k)
		{
			w[k] = ww[k] / 2;
			x[k] = p[k] / 2 + 1.0 * 0.5;
		}
		for (int k = 0; k < n; ++k)
			sum += w[k];
		for (int k = 0; k < n; ++k)
			w[k] /= sum;
		return true;
	}

	bool hex_rule(int order, rule &r)
	{
		double x[8], w[8];
		if (!line_rule(order, x, w))
			return false;
		const int n = order;
		r.n = n * n * n;
		for (int i = 0; i < n; ++i)
			for (int j = 0; j < n; ++j)
				for (int k = 0; k < n; ++k)
				{
					const int idx = (i * n + j) * n + k;
					r.p[idx][0] = x[k];
					r.p[idx][1] = x[j];
					r.p[idx][2] = x[i];
					r.w[idx] = w[i] * w[j] * w[k];
				}
		return true;
	}

	void p1_grad(int li, double *g)
	{
		static const double t[4][3] = {{-1, -1, -1}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
		g[0] = t[li][0], g[1] = t[li][1], g[2] = t[li][2];
	}

	void p2_grad(int li, const double *pt, double *g)
	{
		const double x = pt[0], y = pt[1], z = pt[2];
		switch (li)
		{
		case 0:
			g[0] = g[1] = g[2] = 4 * x + 4 * y + 4 * z - 3;
			break;
		case 1:
			g[0] = 4 * x - 1, g[1] = 0, g[2] = 0;
			break;
		case 2:
			g[0] = 0, g[1] = 4 * y - 1, g[2] = 0;
			break;
		case 3:
			g[0] = 0, g[1] = 0, g[2] = 4 * z - 1;
			break;
		case 4:
			g[0] = 4 * (-2 * x - y - z + 1), g[1] = -4 * x, g[2] = -4 * x;
			break;
		case 5:
			g[0] = 4 * y, g[1] = 4 * x, g[2] = 0;
			break;
		case 6:
			g[0] = -4 * y, g[1] = 4 * (-x - 2 * y - z + 1), g[2] = -4 * y;
			break;
		case 7:
			g[0] = -4 * z, g[1] = -4 * z, g[2] = 4 * (-x - y - 2 * z + 1);
			break;
		case 8:
			g[0] = 4 * z, g[1] = 0, g[2] = 4 * x;
			break;
		case 9:
			g[0] = 0, g[1] = 4 * z, g[2] = 4 * y;
			break;
		}
	}

	void p1tri_grad(int li, double *g)
	{
		static const double t[3][2] = {{-1, -1}, {1, 0}, {0, 1}};
		g[0] = t[li][0], g[1] = t[li][1], g[2] = 0;
	}

	void p2tri_grad(int li, const double *pt, double *g)
	{
		const double x = pt[0], y = pt[1];
		g[2] = 0;
		switch (li)
		{
		case 0:
			g[0] = g[1] = 4 * x + 4 * y - 3;
			break;
		case 1:
			g[0] = 4 * x - 1, g[1] = 0;
			break;
		case 2:
			g[0] = 0, g[1] = 4 * y - 1;
			break;
		case 3:
			g[0] = 4 * (-2 * x - y + 1), g[1] = -4 * x;
			break;
		case 4:
			g[0] = 4 * y, g[1] = 4 * x;
			break;
		case 5:
			g[0] = -4 * y, g[1] = 4 * (-x - 2 * y + 1);
			break;
		}
	}

	// Q1: corner c has reference position q1c[c]; N_c = prod_axis (s ? t : -(t-1)); printed factor order differs
	// per corner in the reference, which only reorders exact-sign factors of two-term products.
	const int q1c[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};

	void q1_grad(int li, const double *pt, double *g)
	{
		// d/d(axis a) of prod_b f_b = sign_a * prod_{b != a} f_b, f_b = t_b (corner at 1) or (t_b - 1) with an
		// overall sign (-1)^(number of corner coordinates at 0)
		double f[3];
		int nzero = 0;
		for (int b = 0; b < 3; ++b)
		{
			f[b] = q1c[li][b] ? pt[b] : (pt[b] - 1);
			nzero += q1c[li][b] ? 0 : 1;
		}
		const double sign = (nzero % 2) ? -1.0 : 1.0;
		// auto_q_bases.cpp:1115-1260 prints "+-1.0 * a * b": the +-1.0 factor is exact and a two-factor product
		// is commutative in IEEE arithmetic, so the factor order does not change the result.
		for (int a = 0; a < 3; ++a)
		{
			const int b0 = (a == 0) ? 1 : 0;
			const int b1 = 3 - a - b0;
			g[a] = sign * f[b0] * f[b1];
		}
	}

	// one factor of a generated Q expression (codes of pf_q2_tables.h)
	double q_factor(int code, double t)
	{
		switch (code)
		{
		case 0:
			return t;
		case 1:
			return t - 1;
		case 2:
			return 2.0 * t - 1.0;
		case 3:
			return 4.0 * t - 3.0;
		case 4:
			return 4.0 * t - 1.0;
		case 5:
			return 16.0 * t - 12.0;
		case 6:
			return 16.0 * t - 4.0;
		case 7:
			return 64.0 * t - 48.0;
		default:
			return 64.0 * t - 16.0;
		}
	}

	// coef * f0 * f1 * ... from left to right, the way the generated C++ expression is evaluated
	double q_eval(const pf_qexpr &e, const double *pt)
	{
		double r = e.coef;
		for (int k = 0; k < 8 && e.f[k] != 0xff; ++k)
		{
			const double v = q_factor(e.f[k] & 15, pt[e.f[k] >> 4]);
			r = (k == 0) ? (e.coef * v) : (r * v);
		}
		return r;
	}

	void q2_grad(int li, const double *pt, double *g)
	{
		for (int c = 0; c < 3; ++c)
			g[c] = q_eval(pf_q2_grad[li][c], pt);
	}

	void basis_grad(int kind, int li, const double *pt, double *g)
	{
		switch (kind)
		{
		case PFX_P1TET:
			p1_grad(li, g);
			break;
		case PFX_P2TET:
			p2_grad(li, pt, g);
			break;
		case PFX_Q1HEX:
			q1_grad(li, pt, g);
			break;
		case PFX_Q2HEX:
			q2_grad(li, pt, g);
			break;
		case PFX_P1TRI:
			p1tri_grad(li, g);
			break;
		case PFX_P2TRI:
			p2tri_grad(li, pt, g);
			break;
		}
	}
	// values: auto_p_bases.cpp:967-999 (P1), :1089-1151 (P2); auto_q_bases.cpp:1067-1110 (Q1, printed factor order),
	// :1277-1418 (Q2, printed factor order)
	double basis_value(int kind, int li, const double *pt)
	{
		const double x = pt[0], y = pt[1], z = pt[2];
		switch (kind)
		{
		case PFX_P1TET:
			return li == 0 ? -x - y - z + 1 : pt[li - 1];
		case PFX_P1TRI:
			return li == 0 ? -x - y + 1 : pt[li - 1];
		case PFX_P2TRI:
			switch (li)
			{
			case 0:
				return (x + y - 1) * (2 * x + 2 * y - 1);
			case 1:
				return x * (2 * x - 1);
			case 2:
				return y * (2 * y - 1);
			case 3:
				return -4 * x * (x + y - 1);
			case 4:
				return 4 * x * y;
			default:
				return -4 * y * (x + y - 1);
			}
		case PFX_P2TET:
			switch (li)
			{
			case 0:
				return (x + y + z - 1) * (2 * x + 2 * y + 2 * z - 1);
			case 1:
				return x * (2 * x - 1);
			case 2:
				return y * (2 * y - 1);
			case 3:
				return z * (2 * z - 1);
			case 4:
				return -4 * x * (x + y + z - 1);
			case 5:
				return 4 * x * y;
			case 6:
				return -4 * y * (x + y + z - 1);
			case 7:
				return -4 * z * (x + y + z - 1);
			case 8:
				return 4 * x * z;
			default:
				return 4 * y * z;
			}
		case PFX_Q1HEX:
		{
			// coefficient and the order in which the generated code multiplies the three axis factors
			static const double coef[8] = {-1, 1, -1, 1, 1, -1, 1, -1};
			static const int order[8][3] = {{0, 1, 2}, {0, 1, 2}, {0, 1, 2}, {1, 0, 2}, {2, 0, 1}, {0, 2, 1}, {0, 1, 2}, {1, 2, 0}};
			double r = 0;
			for (int k = 0; k < 3; ++k)
			{
				const int a = order[li][k];
				const double f = q1c[li][a] ? pt[a] : (pt[a] - 1);
				r = k == 0 ? coef[li] * f : r * f;
			}
			return r;
		}
		default:
			return q_eval(pf_q2_val[li], pt);
		}
	}

	void fill(int kind, int real_order, pfx_tables *t);
} // namespace

void pfx_fill_tables(int kind, int quad_order, pfx_tables *t)
{
	const int p = (kind == PFX_P1TET || kind == PFX_Q1HEX || kind == PFX_P1TRI) ? 1 : 2;
	int real_order = (p - 1) * 2 + 1;
	if (quad_order > real_order)
		real_order = quad_order;
	fill(kind, real_order, t);
}

// mass quadrature: order max(mass_quadrature_order = -1, 2p+1)  (basis/FEBasis3d.cpp:1468, 1507)
void pfx_fill_tables_mass(int kind, pfx_tables *t)
{
	const int p = (kind == PFX_P1TET || kind == PFX_Q1HEX || kind == PFX_P1TRI) ? 1 : 2;
	fill(kind, 2 * p + 1, t);
}

namespace
{
void fill(int kind, int real_order, pfx_tables *t)
{
	std::memset(t, 0, sizeof(*t));
	static const int nloc[6] = {4, 10, 8, 27, 3, 6};
	const bool hex = (kind == PFX_Q1HEX || kind == PFX_Q2HEX), tri = (kind == PFX_P1TRI || kind == PFX_P2TRI);
	const int gkind = hex ? PFX_Q1HEX : tri ? PFX_P1TRI : PFX_P1TET;
	rule r;
	const bool ok = hex ? hex_rule(real_order, r) : tri ? tri_rule(real_order, r) : tet_rule(real_order, r);
	if (!ok)
	{
		t->n_pts = 0;
		return;
	}
	t->n_loc = nloc[kind];
	t->n_gloc = nloc[gkind];
	t->n_pts = r.n;
	for (int k = 0; k < r.n; ++k)
	{
		t->w[k] = r.w[k];
		for (int j = 0; j < t->n_loc; ++j)
		{
			basis_grad(kind, j, r.p[k], t->grad[k][j]);
			t->val[k][j] = basis_value(kind, j, r.p[k]);
		}
		for (int j = 0; j < t->n_gloc; ++j)
		{
			basis_grad(gkind, j, r.p[k], t->ggrad[k][j]);
			t->gval[k][j] = basis_value(gkind, j, r.p[k]);
		}
	}
}
} // namespace

// Reference position of every local node of an element kind and the geometry-basis values there: the node of basis j
// is the lattice point of {0, 1/2, 1}^3 where basis j equals one (the Kronecker property the reference's own tests
// check, tests/test_bases.cpp:716-776).  Used to place the basis nodes in space (aggregation-AMG rigid-body modes).
void pfx_local_node_positions(int kind, int *n_loc, int *n_gloc, double (*pos)[3], double (*gval)[PFX_MAX_GLOC])
{
	static const int nloc[6] = {4, 10, 8, 27, 3, 6};
	const bool hex = (kind == PFX_Q1HEX || kind == PFX_Q2HEX), tri = (kind == PFX_P1TRI || kind == PFX_P2TRI);
	const int gkind = hex ? PFX_Q1HEX : tri ? PFX_P1TRI : PFX_P1TET;
	*n_loc = nloc[kind];
	*n_gloc = nloc[gkind];
	for (int j = 0; j < *n_loc; ++j)
	{
		bool found = false;
		for (int a = 0; a < 3 && !found; ++a)
			for (int b = 0; b < 3 && !found; ++b)
				for (int c = 0; c < 3 && !found; ++c)
				{
					const double pt[3] = {0.5 * a, 0.5 * b, 0.5 * c};
					if (!hex && pt[0] + pt[1] + pt[2] > 1.0)
						continue;
					if (std::fabs(basis_value(kind, j, pt) - 1.0) < 1e-12)
					{
						pos[j][0] = pt[0], pos[j][1] = pt[1], pos[j][2] = pt[2];
						found = true;
					}
				}
		for (int g = 0; g < *n_gloc; ++g)
			gval[j][g] = basis_value(gkind, g, pos[j]);
	}
}
