/* pf_oracle.c -- TEST INFRASTRUCTURE ONLY (see pf_oracle.h).
 *
 * Plain-C restatement of the reference's assembly + CG path.  Compile with -ffp-contract=off so that the
 * floating-point operation order written here is the one executed (the linear assembler's sparsity pattern
 * depends on which local values are exactly < 1e-30 in magnitude, Assembler.cpp:170-173).
 */
#include "pf_oracle.h"
#include "pf_quad_tables.h"
#include "pf_q2_tables.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------ */
/* small helpers                                                                                    */
/* ------------------------------------------------------------------------------------------------ */

void pfo_csc_free(pfo_csc *m)
{
	if (!m)
		return;
	free(m->outer);
	free(m->inner);
	free(m->values);
	m->outer = NULL;
	m->inner = NULL;
	m->values = NULL;
	m->nnz = 0;
}

int pfo_n_loc(int kind)
{
	switch (kind)
	{
	case PFO_P1TET:
		return 4;
	case PFO_P2TET:
		return 10;
	case PFO_Q1HEX:
		return 8;
	case PFO_Q2HEX:
		return 27;
	case PFO_P1TRI:
		return 3;
	case PFO_P2TRI:
		return 6;
	}
	return -1;
}

int pfo_order_of(int kind)
{
	return (kind == PFO_P1TET || kind == PFO_Q1HEX || kind == PFO_P1TRI) ? 1 : 2;
}

static int is_hex(int kind) { return kind == PFO_Q1HEX || kind == PFO_Q2HEX; }
static int is_tri(int kind) { return kind == PFO_P1TRI || kind == PFO_P2TRI; }
int pfo_kind_dim(int kind) { return is_tri(kind) ? 2 : 3; }

/* utils/ElasticityUtils.cpp:203-214 */
double pfo_lambda_from_E_nu(double E, double nu) { return (E * nu) / ((1.0 + nu) * (1.0 - 2.0 * nu)); }
double pfo_mu_from_E_nu(double E, double nu) { return E / (2.0 * (1.0 + nu)); }
double pfo_lambda_from_E_nu_2d(double E, double nu) { return (nu * E) / (1.0 - nu * nu); } /* :208, is_volume == false */

/* ------------------------------------------------------------------------------------------------ */
/* quadrature                                                                                       */
/* ------------------------------------------------------------------------------------------------ */

/* quadrature/TetQuadrature.cpp:31-43 : table lookup, then weights /= 6 */
int pfo_tet_quadrature(int order, pfo_quad *q)
{
	const double *p = NULL, *w = NULL;
	int n = 0;
	switch (order)
	{
	case 1:
		p = pf_tet_pts_1, w = pf_tet_wts_1, n = PF_TET_NPTS_1;
		break;
	case 2:
		p = pf_tet_pts_2, w = pf_tet_wts_2, n = PF_TET_NPTS_2;
		break;
	case 3:
		p = pf_tet_pts_3, w = pf_tet_wts_3, n = PF_TET_NPTS_3;
		break;
	case 4:
		p = pf_tet_pts_4, w = pf_tet_wts_4, n = PF_TET_NPTS_4;
		break;
	case 5:
		p = pf_tet_pts_5, w = pf_tet_wts_5, n = PF_TET_NPTS_5;
		break;
	default:
		return -1;
	}
	q->n_pts = n;
	for (int k = 0; k < n; ++k)
	{
		q->pts[k][0] = p[3 * k + 0];
		q->pts[k][1] = p[3 * k + 1];
		q->pts[k][2] = p[3 * k + 2];
		q->w[k] = w[k] / 6;
	}
	return 0;
}

/* quadrature/TriQuadrature.cpp:30-43 : table lookup (autogen/auto_triangle.ipp), then weights /= 2 */
int pfo_tri_quadrature(int order, pfo_quad *q)
{
	const double *p = NULL, *w = NULL;
	int n = 0;
	switch (order)
	{
	case 1:
		p = pf_tri_pts_1, w = pf_tri_wts_1, n = PF_TRI_NPTS_1;
		break;
	case 2:
		p = pf_tri_pts_2, w = pf_tri_wts_2, n = PF_TRI_NPTS_2;
		break;
	case 3:
		p = pf_tri_pts_3, w = pf_tri_wts_3, n = PF_TRI_NPTS_3;
		break;
	case 4:
		p = pf_tri_pts_4, w = pf_tri_wts_4, n = PF_TRI_NPTS_4;
		break;
	case 5:
		p = pf_tri_pts_5, w = pf_tri_wts_5, n = PF_TRI_NPTS_5;
		break;
	default:
		return -1;
	}
	q->n_pts = n;
	for (int k = 0; k < n; ++k)
	{
		q->pts[k][0] = p[2 * k + 0];
		q->pts[k][1] = p[2 * k + 1];
		q->pts[k][2] = 0;
		q->w[k] = w[k] / 2;
	}
	return 0;
}

/* quadrature/LineQuadrature.cpp:4707-4734.  The two std::sort calls there compare doubles that were
 * converted to int indices (all 0), so the comparator is always false and the table order is kept.
 * weights /= 2 ; points = points/2 + 1*0.5 ; weights /= weights.sum() */
int pfo_line_quadrature(int order, pfo_quad *q)
{
	const double *p = NULL, *w = NULL;
	switch (order)
	{
	case 1:
		p = pf_line_pts_1, w = pf_line_wts_1;
		break;
	case 2:
		p = pf_line_pts_2, w = pf_line_wts_2;
		break;
	case 3:
		p = pf_line_pts_3, w = pf_line_wts_3;
		break;
	case 4:
		p = pf_line_pts_4, w = pf_line_wts_4;
		break;
	case 5:
		p = pf_line_pts_5, w = pf_line_wts_5;
		break;
	default:
		return -1;
	}
	q->n_pts = order;
	double sum = 0;
	for (int k = 0; k < order; ++k)
	{
		q->w[k] = w[k] / 2;
		q->pts[k][0] = p[k] / 2 + 1.0 * 0.5;
		q->pts[k][1] = 0;
		q->pts[k][2] = 0;
	}
	/* Eigen's sum() of a dynamic vector is a plain left-to-right redux for these sizes (<= 5 entries,
	 * packets of 2 doubles would pair (w0+w2)+(w1+w3)...; with so few entries the difference is at most
	 * 1 ulp of a number equal to 1 -- parity unpinned) */
	for (int k = 0; k < order; ++k)
		sum += q->w[k];
	for (int k = 0; k < order; ++k)
		q->w[k] /= sum;
	return 0;
}

/* quadrature/HexQuadrature.cpp:17-45 : tensor product, index (i*n+j)*n+k, point (x_k,y_j,z_i) */
int pfo_hex_quadrature(int order, pfo_quad *q)
{
	pfo_quad l;
	if (pfo_line_quadrature(order, &l))
		return -1;
	const int n = l.n_pts;
	if (n * n * n > PFO_MAX_PTS)
		return -1;
	q->n_pts = n * n * n;
	for (int i = 0; i < n; ++i)
		for (int j = 0; j < n; ++j)
			for (int k = 0; k < n; ++k)
			{
				const int index = (i * n + j) * n + k;
				q->pts[index][0] = l.pts[k][0];
				q->pts[index][1] = l.pts[j][0];
				q->pts[index][2] = l.pts[i][0];
				q->w[index] = l.w[i] * l.w[j] * l.w[k];
			}
	return 0;
}

/* basis/FEBasis3d.cpp:1467 (hex), :1506 (tet), basis/FEBasis2d.cpp:797 (triangle): real_order = max(quadrature_order, 2(p-1)+1) */
int pfo_element_quadrature(int kind, int quad_order, pfo_quad *q)
{
	const int p = pfo_order_of(kind);
	int real_order = (p - 1) * 2 + 1;
	if (quad_order > real_order)
		real_order = quad_order;
	if (is_tri(kind))
		return pfo_tri_quadrature(real_order, q);
	return is_hex(kind) ? pfo_hex_quadrature(real_order, q) : pfo_tet_quadrature(real_order, q);
}

/* basis/FEBasis3d.cpp:1468 (hex), :1507 (tet): real_mass_order = max(mass_quadrature_order, 2p+1), with the default
 * mass_quadrature_order = -1 */
int pfo_mass_quadrature(int kind, pfo_quad *q)
{
	const int p = pfo_order_of(kind);
	const int real_mass_order = p * 2 + 1;
	if (is_tri(kind)) /* basis/FEBasis2d.cpp:798 */
		return pfo_tri_quadrature(real_mass_order, q);
	return is_hex(kind) ? pfo_hex_quadrature(real_mass_order, q) : pfo_tet_quadrature(real_mass_order, q);
}

/* ------------------------------------------------------------------------------------------------ */
/* shape functions                                                                                  */
/* ------------------------------------------------------------------------------------------------ */

/* One factor of a Q basis expression as the reference's generator printed it; evaluated in place. */
enum
{
	F_X,     /* t            */
	F_XM1,   /* (t - 1)      */
	F_2XM1,  /* (2.0*t - 1.0)*/
	F_4XM3,  /* (4.0*t - 3.0)*/
	F_4XM1,  /* (4.0*t - 1.0)*/
	F_16XM12,
	F_16XM4,
	F_64XM48,
	F_64XM16,
	F_END
};

static double qfac(int code, double t)
{
	switch (code)
	{
	case F_X:
		return t;
	case F_XM1:
		return t - 1;
	case F_2XM1:
		return 2.0 * t - 1.0;
	case F_4XM3:
		return 4.0 * t - 3.0;
	case F_4XM1:
		return 4.0 * t - 1.0;
	case F_16XM12:
		return 16.0 * t - 12.0;
	case F_16XM4:
		return 16.0 * t - 4.0;
	case F_64XM48:
		return 64.0 * t - 48.0;
	case F_64XM16:
		return 64.0 * t - 16.0;
	}
	return 0;
}

/* A Q expression: coef * f0 * f1 * ... evaluated left to right.  Each factor = (axis<<4)|code.
 * The factor order of every expression is the one printed in autogen/auto_q_bases.cpp:1067-1871. */
typedef struct
{
	double coef;
	unsigned char f[7];
} qexpr;
#define X(c) ((0 << 4) | (c))
#define Y(c) ((1 << 4) | (c))
#define Z(c) ((2 << 4) | (c))
#define E_ 0xff

static double qeval(const qexpr *e, const double *pt)
{
	double r = e->coef;
	int first = 1;
	for (int k = 0; k < 7 && e->f[k] != E_; ++k)
	{
		const double v = qfac(e->f[k] & 15, pt[e->f[k] >> 4]);
		/* coef is +-1, +-4, +-16, +-64: "c * f0" ; a leading "-x" is coef=-1 times x (exact) */
		r = first ? (e->coef * v) : (r * v);
		first = 0;
	}
	return r;
}

/* same for the generated Q2 tables */
static double qeval2(const pf_qexpr *e, const double *pt)
{
	double r = e->coef;
	for (int k = 0; k < 8 && e->f[k] != 0xff; ++k)
	{
		const double v = qfac(e->f[k] & 15, pt[e->f[k] >> 4]);
		r = (k == 0) ? (e->coef * v) : (r * v);
	}
	return r;
}

/* autogen/auto_q_bases.cpp:1067-1110 (values), :1115-1260 (gradients) */
static const qexpr q1_val[8] = {
	{-1.0, {X(F_XM1), Y(F_XM1), Z(F_XM1), E_}},
	{1.0, {X(F_X), Y(F_XM1), Z(F_XM1), E_}},
	{-1.0, {X(F_X), Y(F_X), Z(F_XM1), E_}},
	{1.0, {Y(F_X), X(F_XM1), Z(F_XM1), E_}},
	{1.0, {Z(F_X), X(F_XM1), Y(F_XM1), E_}},
	{-1.0, {X(F_X), Z(F_X), Y(F_XM1), E_}},
	{1.0, {X(F_X), Y(F_X), Z(F_X), E_}},
	{-1.0, {Y(F_X), Z(F_X), X(F_XM1), E_}},
};
static const qexpr q1_grad[8][3] = {
	{{-1.0, {Y(F_XM1), Z(F_XM1), E_}}, {-1.0, {X(F_XM1), Z(F_XM1), E_}}, {-1.0, {X(F_XM1), Y(F_XM1), E_}}},
	{{1.0, {Y(F_XM1), Z(F_XM1), E_}}, {1.0, {X(F_X), Z(F_XM1), E_}}, {1.0, {X(F_X), Y(F_XM1), E_}}},
	{{-1.0, {Y(F_X), Z(F_XM1), E_}}, {-1.0, {X(F_X), Z(F_XM1), E_}}, {-1.0, {X(F_X), Y(F_X), E_}}},
	{{1.0, {Y(F_X), Z(F_XM1), E_}}, {1.0, {X(F_XM1), Z(F_XM1), E_}}, {1.0, {Y(F_X), X(F_XM1), E_}}},
	{{1.0, {Z(F_X), Y(F_XM1), E_}}, {1.0, {Z(F_X), X(F_XM1), E_}}, {1.0, {X(F_XM1), Y(F_XM1), E_}}},
	{{-1.0, {Z(F_X), Y(F_XM1), E_}}, {-1.0, {X(F_X), Z(F_X), E_}}, {-1.0, {X(F_X), Y(F_XM1), E_}}},
	{{1.0, {Y(F_X), Z(F_X), E_}}, {1.0, {X(F_X), Z(F_X), E_}}, {1.0, {X(F_X), Y(F_X), E_}}},
	{{-1.0, {Y(F_X), Z(F_X), E_}}, {-1.0, {Z(F_X), X(F_XM1), E_}}, {-1.0, {Y(F_X), X(F_XM1), E_}}},
};

/* Q2: local node order from the node table auto_q_bases.cpp:1873-1903.  Values and gradients
 * (auto_q_bases.cpp:1277-1871) are evaluated from pf_q2_tables.h, which records the factor order of every generated
 * expression; q2_l / q2_dl below (the 1-D quadratic Lagrange factors) are kept as an independent cross-check of those
 * tables (tests compare the two to a few ulp). */
static const signed char q2_node[27][3] = {
	/* lattice position of each local node in units of 1/2 (auto_q_bases.cpp:1873-1903) */
	{0, 0, 0}, {2, 0, 0}, {2, 2, 0}, {0, 2, 0}, {0, 0, 2}, {2, 0, 2}, {2, 2, 2}, {0, 2, 2},
	{1, 0, 0}, {2, 1, 0}, {1, 2, 0}, {0, 1, 0}, {0, 0, 1}, {2, 0, 1}, {2, 2, 1}, {0, 2, 1},
	{1, 0, 2}, {2, 1, 2}, {1, 2, 2}, {0, 1, 2}, {0, 1, 1}, {2, 1, 1}, {1, 0, 1}, {1, 2, 1},
	{1, 1, 0}, {1, 1, 2}, {1, 1, 1}};

/* 1-D factors: l0(t) = (t-1)(2t-1), l2(t) = t(2t-1), l1(t) = -4 t (t-1);
 *              l0' = 4t-3,          l2' = 4t-1,      l1' = -4(2t-1)                      */
static double q2_l(int a, double t)
{
	switch (a)
	{
	case 0:
		return (t - 1) * (2.0 * t - 1.0);
	case 2:
		return t * (2.0 * t - 1.0);
	default:
		return -4.0 * t * (t - 1);
	}
}
static double q2_dl(int a, double t)
{
	switch (a)
	{
	case 0:
		return 4.0 * t - 3.0;
	case 2:
		return 4.0 * t - 1.0;
	default:
		return -4.0 * (2.0 * t - 1.0);
	}
}

void pfo_basis_nodes(int kind, double *nodes)
{
	static const double p1[4][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
	static const double p2[10][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {.5, 0, 0}, {.5, .5, 0}, {0, .5, 0}, {0, 0, .5}, {.5, 0, .5}, {0, .5, .5}};
	static const double q1[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
	switch (kind)
	{
	case PFO_P1TET: /* auto_p_bases.cpp:1080-1087 */
		memcpy(nodes, p1, sizeof(p1));
		break;
	case PFO_P2TET: /* auto_p_bases.cpp:1328-1341 */
		memcpy(nodes, p2, sizeof(p2));
		break;
	case PFO_Q1HEX: /* auto_q_bases.cpp:1262-1275 */
		memcpy(nodes, q1, sizeof(q1));
		break;
	case PFO_Q2HEX:
		for (int i = 0; i < 27; ++i)
			for (int c = 0; c < 3; ++c)
				nodes[3 * i + c] = q2_node[i][c] / 2.0;
		break;
	case PFO_P1TRI: /* auto_p_bases.cpp:138-144 (third coordinate 0) */
	{
		static const double t1[3][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}};
		memcpy(nodes, t1, sizeof(t1));
		break;
	}
	case PFO_P2TRI: /* auto_p_bases.cpp:275-284 */
	{
		static const double t2[6][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {1.0 / 2.0, 0, 0}, {1.0 / 2.0, 1.0 / 2.0, 0}, {0, 1.0 / 2.0, 0}};
		memcpy(nodes, t2, sizeof(t2));
		break;
	}
	}
}

/* Q2 basis and gradient as tensor products of the 1-D factors: cross-check of the generated tables */
void pfo_q2_tensor_basis(int li, const double *pt, double *val, double *g)
{
	const int a = q2_node[li][0], b = q2_node[li][1], c = q2_node[li][2];
	*val = q2_l(a, pt[0]) * q2_l(b, pt[1]) * q2_l(c, pt[2]);
	g[0] = q2_dl(a, pt[0]) * q2_l(b, pt[1]) * q2_l(c, pt[2]);
	g[1] = q2_l(a, pt[0]) * q2_dl(b, pt[1]) * q2_l(c, pt[2]);
	g[2] = q2_l(a, pt[0]) * q2_l(b, pt[1]) * q2_dl(c, pt[2]);
}

void pfo_basis_value(int kind, int li, const double *pt, double *val)
{
	const double x = pt[0], y = pt[1], z = pt[2];
	switch (kind)
	{
	case PFO_P1TET: /* auto_p_bases.cpp:967-999 */
		switch (li)
		{
		case 0:
			*val = -x - y - z + 1;
			break;
		case 1:
			*val = x;
			break;
		case 2:
			*val = y;
			break;
		case 3:
			*val = z;
			break;
		}
		break;
	case PFO_P2TET: /* auto_p_bases.cpp:1089-1151 */
		switch (li)
		{
		case 0:
			*val = (x + y + z - 1) * (2 * x + 2 * y + 2 * z - 1);
			break;
		case 1:
			*val = x * (2 * x - 1);
			break;
		case 2:
			*val = y * (2 * y - 1);
			break;
		case 3:
			*val = z * (2 * z - 1);
			break;
		case 4:
			*val = -4 * x * (x + y + z - 1);
			break;
		case 5:
			*val = 4 * x * y;
			break;
		case 6:
			*val = -4 * y * (x + y + z - 1);
			break;
		case 7:
			*val = -4 * z * (x + y + z - 1);
			break;
		case 8:
			*val = 4 * x * z;
			break;
		case 9:
			*val = 4 * y * z;
			break;
		}
		break;
	case PFO_Q1HEX:
		*val = qeval(&q1_val[li], pt);
		break;
	case PFO_Q2HEX: /* auto_q_bases.cpp:1277-1418, factors multiplied in the printed order (pf_q2_tables.h) */
		*val = qeval2(&pf_q2_val[li], pt);
		break;
	case PFO_P1TRI: /* auto_p_bases.cpp:60-86 */
		switch (li)
		{
		case 0:
			*val = -x - y + 1;
			break;
		case 1:
			*val = x;
			break;
		case 2:
			*val = y;
			break;
		}
		break;
	case PFO_P2TRI: /* auto_p_bases.cpp:146-187 */
		switch (li)
		{
		case 0:
			*val = (x + y - 1) * (2 * x + 2 * y - 1);
			break;
		case 1:
			*val = x * (2 * x - 1);
			break;
		case 2:
			*val = y * (2 * y - 1);
			break;
		case 3:
			*val = -4 * x * (x + y - 1);
			break;
		case 4:
			*val = 4 * x * y;
			break;
		case 5:
			*val = -4 * y * (x + y - 1);
			break;
		}
		break;
	}
}

void pfo_basis_grad(int kind, int li, const double *pt, double *g)
{
	const double x = pt[0], y = pt[1], z = pt[2];
	switch (kind)
	{
	case PFO_P1TET: /* auto_p_bases.cpp:1000-1078 */
	{
		static const double t[4][3] = {{-1, -1, -1}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
		g[0] = t[li][0], g[1] = t[li][1], g[2] = t[li][2];
		break;
	}
	case PFO_P2TET: /* auto_p_bases.cpp:1152-1326 */
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
		break;
	case PFO_Q1HEX:
		for (int c = 0; c < 3; ++c)
			g[c] = qeval(&q1_grad[li][c], pt);
		break;
	case PFO_Q2HEX: /* auto_q_bases.cpp:1420-1871, factors multiplied in the printed order (pf_q2_tables.h) */
		for (int c = 0; c < 3; ++c)
			g[c] = qeval2(&pf_q2_grad[li][c], pt);
		break;
	case PFO_P1TRI: /* auto_p_bases.cpp:87-136 */
	{
		static const double t[3][2] = {{-1, -1}, {1, 0}, {0, 1}};
		g[0] = t[li][0], g[1] = t[li][1], g[2] = 0;
		break;
	}
	case PFO_P2TRI: /* auto_p_bases.cpp:188-273 */
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
		break;
	}
}

/* ------------------------------------------------------------------------------------------------ */
/* 3x3 dense pieces done by Eigen 3.4.0 in the reference (un-vendored; restated; parity unpinned)    */
/* ------------------------------------------------------------------------------------------------ */

#define M(m, r, c) ((m)[3 * (r) + (c)])

/* Eigen/src/LU/Determinant.h: bruteforce_det3_helper(m,0,1,2) - helper(m,1,0,2) + helper(m,2,0,1),
 * helper(m,a,b,c) = m(0,a) * (m(1,b)*m(2,c) - m(1,c)*m(2,b)) */
static double det3_eigen(const double *m)
{
	const double h0 = M(m, 0, 0) * (M(m, 1, 1) * M(m, 2, 2) - M(m, 1, 2) * M(m, 2, 1));
	const double h1 = M(m, 0, 1) * (M(m, 1, 0) * M(m, 2, 2) - M(m, 1, 2) * M(m, 2, 0));
	const double h2 = M(m, 0, 2) * (M(m, 1, 0) * M(m, 2, 1) - M(m, 1, 1) * M(m, 2, 0));
	return h0 - h1 + h2;
}

static double cof3(const double *m, int i, int j)
{
	const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
	return M(m, i1, j1) * M(m, i2, j2) - M(m, i1, j2) * M(m, i2, j1);
}

/* Eigen/src/LU/InverseImpl.h compute_inverse<.,.,3>: cofactors of column 0, det = sum(cof .* col0),
 * invdet = 1/det, inverse(r,c) = cofactor<c,r> * invdet */
static void inv3_eigen(const double *m, double *inv)
{
	const double c0 = cof3(m, 0, 0), c1 = cof3(m, 1, 0), c2 = cof3(m, 2, 0);
	const double det = (c0 * M(m, 0, 0) + c1 * M(m, 1, 0)) + c2 * M(m, 2, 0);
	const double invdet = 1.0 / det;
	M(inv, 0, 0) = c0 * invdet;
	M(inv, 0, 1) = c1 * invdet;
	M(inv, 0, 2) = c2 * invdet;
	M(inv, 1, 0) = cof3(m, 0, 1) * invdet;
	M(inv, 1, 1) = cof3(m, 1, 1) * invdet;
	M(inv, 1, 2) = cof3(m, 2, 1) * invdet;
	M(inv, 2, 0) = cof3(m, 0, 2) * invdet;
	M(inv, 2, 1) = cof3(m, 1, 2) * invdet;
	M(inv, 2, 2) = cof3(m, 2, 2) * invdet;
}

/* utils/MatrixUtils.hpp:16-30 */
double pfo_determinant3(const double *m)
{
	return M(m, 0, 0) * (M(m, 1, 1) * M(m, 2, 2) - M(m, 1, 2) * M(m, 2, 1)) - M(m, 0, 1) * (M(m, 1, 0) * M(m, 2, 2) - M(m, 1, 2) * M(m, 2, 0)) + M(m, 0, 2) * (M(m, 1, 0) * M(m, 2, 1) - M(m, 1, 1) * M(m, 2, 0));
}

/* ------------------------------------------------------------------------------------------------ */
/* ElementAssemblyValues::compute + finalize3d                                                      */
/* ------------------------------------------------------------------------------------------------ */

/* ElementAssemblyValues::finalize2d (assembler/ElementAssemblyValues.cpp:161-197): Jacobian rows summed over the
 * geometry bases, det = Eigen's 2x2 determinant (Eigen/src/LU/Determinant.h: m(0,0) m(1,1) - m(1,0) m(0,1)), jac_it =
 * transposed 2x2 inverse (Eigen/src/LU/InverseImpl.h compute_inverse_size2_helper: invdet = 1/det; (0,0) = m(1,1) invdet,
 * (1,0) = -m(1,0) invdet, (0,1) = -m(0,1) invdet, (1,1) = m(0,0) invdet), physical gradients grad.row(k) * jac_it.  The 2x2
 * matrices sit in the upper left of the 3x3 slots, everything else is zero.  Eigen un-vendored: parity unpinned. */
static void finalize2d(const pfo_mesh *mesh, int64_t e, const pfo_quad *quad, pfo_vals *v)
{
	const int n_g = pfo_n_loc(mesh->gkind);
	for (int k = 0; k < quad->n_pts; ++k)
	{
		double t00 = 0, t01 = 0, t10 = 0, t11 = 0;
		for (int j = 0; j < n_g; ++j)
		{
			double gg[3];
			pfo_basis_grad(mesh->gkind, j, quad->pts[k], gg);
			const double *node = mesh->xyz + 2 * (int64_t)mesh->gconn[e * n_g + j];
			const double gval = 1.0; /* Local2Global.val */
			t00 += gg[0] * node[0] * gval, t01 += gg[0] * node[1] * gval;
			t10 += gg[1] * node[0] * gval, t11 += gg[1] * node[1] * gval;
		}
		const double det = t00 * t11 - t10 * t01;
		v->det[k] = det;
		const double invdet = 1.0 / det;
		const double i00 = t11 * invdet, i10 = -t10 * invdet, i01 = -t01 * invdet, i11 = t00 * invdet;
		double *jt = v->jac_it[k];
		for (int q = 0; q < 9; ++q)
			jt[q] = 0;
		M(jt, 0, 0) = i00, M(jt, 0, 1) = i10, M(jt, 1, 0) = i01, M(jt, 1, 1) = i11;
		for (int j = 0; j < v->n_loc; ++j)
		{
			const double *g = v->grad[j][k];
			v->grad_t_m[j][k][0] = g[0] * M(jt, 0, 0) + g[1] * M(jt, 1, 0);
			v->grad_t_m[j][k][1] = g[0] * M(jt, 0, 1) + g[1] * M(jt, 1, 1);
			v->grad_t_m[j][k][2] = 0;
		}
	}
}

/* assembler/ElementAssemblyValues.cpp:205-351 and :94-133.  Local2Global lists have size 1, val 1.0. */
void pfo_compute_vals(const pfo_mesh *mesh, int64_t e, const pfo_quad *quad, pfo_vals *v)
{
	const int n_loc = pfo_n_loc(mesh->kind);
	const int n_g = pfo_n_loc(mesh->gkind);
	const int m = quad->n_pts;
	v->n_pts = m;
	v->n_loc = n_loc;
	v->dim = pfo_kind_dim(mesh->kind);
	v->quad = *quad;

	for (int j = 0; j < n_loc; ++j)
		for (int k = 0; k < m; ++k)
		{
			pfo_basis_value(mesh->kind, j, quad->pts[k], &v->val[j][k]);
			pfo_basis_grad(mesh->kind, j, quad->pts[k], v->grad[j][k]);
		}

	if (v->dim == 2)
	{
		finalize2d(mesh, e, quad, v);
		return;
	}
	for (int k = 0; k < m; ++k)
	{
		double tmp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
		for (int j = 0; j < n_g; ++j)
		{
			double gg[3];
			pfo_basis_grad(mesh->gkind, j, quad->pts[k], gg);
			const double *node = mesh->xyz + 3 * (int64_t)mesh->gconn[e * n_g + j];
			const double gval = 1.0; /* Local2Global.val */
			for (int r = 0; r < 3; ++r)
				for (int c = 0; c < 3; ++c)
					M(tmp, r, c) += gg[r] * node[c] * gval;
		}
		v->det[k] = det3_eigen(tmp);
		double inv[9];
		inv3_eigen(tmp, inv);
		for (int r = 0; r < 3; ++r)
			for (int c = 0; c < 3; ++c)
				M(v->jac_it[k], r, c) = M(inv, c, r);
		for (int j = 0; j < n_loc; ++j)
			for (int c = 0; c < 3; ++c)
			{
				const double *g = v->grad[j][k];
				const double *jt = v->jac_it[k];
				v->grad_t_m[j][k][c] = (g[0] * M(jt, 0, c) + g[1] * M(jt, 1, c)) + g[2] * M(jt, 2, c);
			}
	}
}

/* ------------------------------------------------------------------------------------------------ */
/* local assemblers                                                                                 */
/* ------------------------------------------------------------------------------------------------ */

/* assembler/Laplacian.cpp:7-18 */
void pfo_local_laplacian(const pfo_vals *v, int i, int j, const double *da, double *res1)
{
	double res = 0;
	for (int k = 0; k < v->n_pts; ++k)
	{
		const double *gi = v->grad_t_m[i][k], *gj = v->grad_t_m[j][k];
		const double dot = (gi[0] * gj[0] + gi[1] * gj[1]) + gi[2] * gj[2];
		res += dot * da[k];
	}
	*res1 = res;
}

/* assembler/LinearElasticity.cpp:21-54 ; res index jj*size+ii */
void pfo_local_linear_elasticity(const pfo_vals *v, int i, int j, const double *da, double lambda, double mu, double *res)
{
	for (int c = 0; c < 9; ++c)
		res[c] = 0;
	for (int k = 0; k < v->n_pts; ++k)
	{
		const double *gi = v->grad_t_m[i][k], *gj = v->grad_t_m[j][k];
		double outer[9]; /* column-major 3x3: outer[col*3+row] = gi[row]*gj[col] */
		for (int col = 0; col < 3; ++col)
			for (int row = 0; row < 3; ++row)
				outer[col * 3 + row] = gi[row] * gj[col];
		const double dot = (gi[0] * gj[0] + gi[1] * gj[1]) + gi[2] * gj[2];
		double res_k[9];
		for (int ii = 0; ii < 3; ++ii)
			for (int jj = 0; jj < 3; ++jj)
			{
				res_k[jj * 3 + ii] = outer[ii * 3 + jj] * mu + outer[jj * 3 + ii] * lambda;
				if (ii == jj)
					res_k[jj * 3 + ii] += mu * dot;
			}
		for (int c = 0; c < 9; ++c)
			res[c] += res_k[c] * da[k];
	}
}

/* assembler/HookeLinearElasticity.cpp:75-190 (3-D branch) ; res index jj*3+ii.  strain<3>(grad, jac_it, k, coo) puts
 * the physical gradient into row coo of a zero matrix and symmetrises it; the Voigt strain (engineering shears) goes
 * through ElasticityTensor::compute_stress (utils/ElasticityUtils.cpp:469-477) and the entry is trace(sigma_i * eps_j). */
static void hooke_strain(const double *g, int coo, double *eps)
{
	double J[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
	for (int c = 0; c < 3; ++c)
		M(J, coo, c) = g[c];
	for (int r = 0; r < 3; ++r)
		for (int c = 0; c < 3; ++c)
			M(eps, r, c) = (M(J, r, c) + M(J, c, r)) * 0.5;
}

void pfo_local_hooke(const pfo_vals *v, int i, int j, const double *da, const double *C, double *res)
{
	for (int c = 0; c < 9; ++c)
		res[c] = 0;
	for (int k = 0; k < v->n_pts; ++k)
	{
		const double *gi = v->grad_t_m[i][k], *gj = v->grad_t_m[j][k];
		double sigma[3][9], epsj[3][9];
		for (int coo = 0; coo < 3; ++coo)
		{
			double ei[9];
			hooke_strain(gi, coo, ei);
			hooke_strain(gj, coo, epsj[coo]);
			const double e6[6] = {M(ei, 0, 0), M(ei, 1, 1), M(ei, 2, 2), 2 * M(ei, 1, 2), 2 * M(ei, 0, 2), 2 * M(ei, 0, 1)};
			double s6[6];
			for (int a = 0; a < 6; ++a)
			{
				double r = 0;
				for (int b = 0; b < 6; ++b)
					r += C[a * 6 + b] * e6[b];
				s6[a] = r;
			}
			double *sg = sigma[coo];
			M(sg, 0, 0) = s6[0], M(sg, 0, 1) = s6[5], M(sg, 0, 2) = s6[4];
			M(sg, 1, 0) = s6[5], M(sg, 1, 1) = s6[1], M(sg, 1, 2) = s6[3];
			M(sg, 2, 0) = s6[4], M(sg, 2, 1) = s6[3], M(sg, 2, 2) = s6[2];
		}
		double res_k[9];
		for (int ci = 0; ci < 3; ++ci)
			for (int cj = 0; cj < 3; ++cj)
			{
				const double *sg = sigma[ci], *ej = epsj[cj];
				double tr = 0;
				for (int r = 0; r < 3; ++r)
				{
					const double dr = (M(sg, r, 0) * M(ej, 0, r) + M(sg, r, 1) * M(ej, 1, r)) + M(sg, r, 2) * M(ej, 2, r);
					tr = r == 0 ? dr : tr + dr;
				}
				res_k[cj * 3 + ci] = tr;
			}
		for (int c = 0; c < 9; ++c)
			res[c] += res_k[c] * da[k];
	}
}

/* assembler/HookeLinearElasticity.cpp:92-122 (size_ == 2): strain<2>() as above with 2 x 2 matrices, the Voigt strain
 * (xx, yy, 2 xy) through the 3 x 3 tensor (compute_stress<3>, utils/ElasticityUtils.cpp:469-477), sigma = [[s0, s2], [s2, s1]],
 * entry jj*2+ii = trace(sigma_i * eps_j).  C: symmetric 3 x 3, row-major. */
static void hooke_strain_2d(const double *g, int coo, double *eps)
{
	double J[4] = {0, 0, 0, 0};
	J[2 * coo] = g[0], J[2 * coo + 1] = g[1];
	for (int r = 0; r < 2; ++r)
		for (int c = 0; c < 2; ++c)
			eps[2 * r + c] = (J[2 * r + c] + J[2 * c + r]) * 0.5;
}

void pfo_local_hooke_2d(const pfo_vals *v, int i, int j, const double *da, const double *C, double *res)
{
	for (int c = 0; c < 4; ++c)
		res[c] = 0;
	for (int k = 0; k < v->n_pts; ++k)
	{
		const double *gi = v->grad_t_m[i][k], *gj = v->grad_t_m[j][k];
		double sigma[2][4], epsj[2][4];
		for (int coo = 0; coo < 2; ++coo)
		{
			double ei[4];
			hooke_strain_2d(gi, coo, ei);
			hooke_strain_2d(gj, coo, epsj[coo]);
			const double e3[3] = {ei[0], ei[3], 2 * ei[1]};
			double s3[3];
			for (int a = 0; a < 3; ++a)
			{
				double r = 0;
				for (int b = 0; b < 3; ++b)
					r += C[a * 3 + b] * e3[b];
				s3[a] = r;
			}
			double *sg = sigma[coo];
			sg[0] = s3[0], sg[1] = s3[2], sg[2] = s3[2], sg[3] = s3[1];
		}
		double res_k[4];
		for (int ci = 0; ci < 2; ++ci)
			for (int cj = 0; cj < 2; ++cj)
			{
				const double *sg = sigma[ci], *ej = epsj[cj];
				const double d0 = sg[0] * ej[0] + sg[1] * ej[2];
				const double d1 = sg[2] * ej[1] + sg[3] * ej[3];
				res_k[cj * 2 + ci] = d0 + d1;
			}
		for (int c = 0; c < 4; ++c)
			res[c] += res_k[c] * da[k];
	}
}

/* F = local_disp^T * grad(p) * jac_it(p) + I ; local_disp is n_loc x 3 (row i = displacement of basis i).
 * NeoHookeanElasticity.cpp:402 / :491 (dense products).  F row-major. */
static void def_grad_at_id(const pfo_vals *v, int p, const double *local_disp, double *F, double id)
{
	double A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; /* local_disp^T * grad : A(d,c) = sum_i u_i[d] grad_i[c] */
	for (int d = 0; d < 3; ++d)
		for (int c = 0; c < 3; ++c)
		{
			double s = 0;
			for (int i = 0; i < v->n_loc; ++i)
				s += local_disp[3 * i + d] * v->grad[i][p][c];
			M(A, d, c) = s;
		}
	const double *jt = v->jac_it[p];
	for (int d = 0; d < 3; ++d)
		for (int c = 0; c < 3; ++c)
		{
			double s = 0;
			for (int a = 0; a < 3; ++a)
				s += M(A, d, a) * M(jt, a, c);
			M(F, d, c) = (id != 0.0) ? s + (d == c ? id : 0.0) : s;
		}
}

static void def_grad_at(const pfo_vals *v, int p, const double *local_disp, double *F)
{
	def_grad_at_id(v, p, local_disp, F, 1.0);
}

/* NeoHookeanElasticity.cpp:250-331 : sum_p da_p [ mu/2 (tr(F^T F) - 3 - 2 ln J) + lambda/2 ln^2 J ] */
double pfo_local_neohookean_energy(const pfo_vals *v, const double *da, const double *local_disp, double lambda, double mu)
{
	double energy = 0;
	for (int p = 0; p < v->n_pts; ++p)
	{
		double F[9];
		def_grad_at(v, p, local_disp, F);
		const double log_det_j = log(pfo_determinant3(F));
		double tr = 0;
		for (int k = 0; k < 9; ++k)
			tr += F[k] * F[k];
		const double val = mu / 2 * (tr - 3 - 2 * log_det_j) + lambda / 2 * log_det_j * log_det_j;
		energy += val * da[p];
	}
	return energy;
}

/* LinearElasticity.cpp:96-169 : sum_p da_p [ mu tr(eps^T eps) + lambda/2 tr(eps)^2 ], eps = sym(grad u) */
double pfo_local_linear_elasticity_energy(const pfo_vals *v, const double *da, const double *local_disp, double lambda, double mu)
{
	double energy = 0;
	for (int p = 0; p < v->n_pts; ++p)
	{
		double F[9];
		def_grad_at_id(v, p, local_disp, F, 0.0); /* grad u (LinearElasticity.cpp:139-157) */
		double s = 0, tr = 0;
		for (int a = 0; a < 3; ++a)
		{
			tr += (M(F, a, a) + M(F, a, a)) / 2;
			for (int b = 0; b < 3; ++b)
			{
				const double eab = (M(F, a, b) + M(F, b, a)) / 2;
				s += eab * eab;
			}
		}
		energy += (mu * s + lambda / 2 * tr * tr) * da[p];
	}
	return energy;
}

/* physical gradient of basis i at point p: grad(p) * jac_it(p) */
static void phys_grad(const pfo_vals *v, int i, int p, double *G)
{
	const double *jt = v->jac_it[p];
	for (int c = 0; c < 3; ++c)
	{
		double s = 0;
		for (int a = 0; a < 3; ++a)
			s += v->grad[i][p][a] * M(jt, a, c);
		G[c] = s;
	}
}

/* LinearElasticity.cpp:61-77 : gradient of compute_energy_aux w.r.t. the local displacements.  The reference obtains
 * it by forward autodiff (DScalar1); the derivative written out is  d/du_{i,d} = sum_p da_p (sigma G_i)[d],
 * sigma = 2 mu eps + lambda tr(eps) I.  G is n_loc*3, entry i*3+d. */
void pfo_local_linear_elasticity_grad(const pfo_vals *v, const double *da, const double *local_disp, double lambda, double mu, double *G)
{
	for (int k = 0; k < 3 * v->n_loc; ++k)
		G[k] = 0;
	for (int p = 0; p < v->n_pts; ++p)
	{
		double D[9], sg[9];
		def_grad_at_id(v, p, local_disp, D, 0.0);
		const double tr = M(D, 0, 0) + M(D, 1, 1) + M(D, 2, 2);
		for (int a = 0; a < 3; ++a)
			for (int b = 0; b < 3; ++b)
				M(sg, a, b) = 2 * mu * ((M(D, a, b) + M(D, b, a)) / 2) + (a == b ? lambda * tr : 0.0);
		for (int i = 0; i < v->n_loc; ++i)
		{
			double Gi[3];
			phys_grad(v, i, p, Gi);
			for (int d = 0; d < 3; ++d)
				G[3 * i + d] += (M(sg, d, 0) * Gi[0] + M(sg, d, 1) * Gi[1] + M(sg, d, 2) * Gi[2]) * da[p];
		}
	}
}

/* LinearElasticity.cpp:79-94 : Hessian of compute_energy_aux (DScalar2 autodiff in the reference), written out:
 * d2/du_{i,a} du_{j,b} = sum_p da_p [ mu (G_i.G_j) delta_ab + mu G_i[b] G_j[a] + lambda G_i[a] G_j[b] ].
 * H is (n_loc*3)^2 row-major, row i*3+a, column j*3+b. */
void pfo_local_linear_elasticity_hessian(const pfo_vals *v, const double *da, double lambda, double mu, double *H)
{
	const int N = 3 * v->n_loc;
	for (int k = 0; k < N * N; ++k)
		H[k] = 0;
	for (int p = 0; p < v->n_pts; ++p)
		for (int i = 0; i < v->n_loc; ++i)
		{
			double Gi[3];
			phys_grad(v, i, p, Gi);
			for (int j = 0; j < v->n_loc; ++j)
			{
				double Gj[3];
				phys_grad(v, j, p, Gj);
				const double dot = Gi[0] * Gj[0] + Gi[1] * Gj[1] + Gi[2] * Gj[2];
				for (int a = 0; a < 3; ++a)
					for (int b = 0; b < 3; ++b)
						H[(3 * i + a) * N + 3 * j + b] += (mu * Gi[b] * Gj[a] + lambda * Gi[a] * Gj[b] + (a == b ? mu * dot : 0.0)) * da[p];
			}
		}
}

/* ---- SaintVenantElasticity (assembler/SaintVenantElasticity.cpp): isotropic tensor from (lambda, mu), or the tensor
 * of the material's "elasticity_tensor" entries (add_multimaterial :42-69) ---- */

/* ElasticityTensor::set_from_lambda_mu (utils/ElasticityUtils.cpp:372-413), 3-D: symmetric 6 x 6, Voigt order
 * xx, yy, zz, yz, xz, xy with engineering shear strains */
static void sv_tensor(double lambda, double mu, double C[6][6])
{
	for (int a = 0; a < 6; ++a)
		for (int b = 0; b < 6; ++b)
			C[a][b] = 0;
	for (int a = 0; a < 3; ++a)
	{
		for (int b = 0; b < 3; ++b)
			C[a][b] = lambda;
		C[a][a] = 2 * mu + lambda;
		C[3 + a][3 + a] = mu;
	}
}

/* strain_from_disp_grad (:15-24) on G = F - I, then the Voigt strain / stress of compute_energy_aux (:276-302) */
static void sv_strain_stress(const double *F, double C[6][6], double *E, double *S)
{
	double G[9];
	for (int a = 0; a < 3; ++a)
		for (int b = 0; b < 3; ++b)
			M(G, a, b) = M(F, a, b) - (a == b ? 1.0 : 0.0);
	for (int a = 0; a < 3; ++a)
		for (int b = 0; b < 3; ++b)
		{
			double gtg = 0;
			for (int k = 0; k < 3; ++k)
				gtg += M(G, k, a) * M(G, k, b);
			M(E, a, b) = (gtg + M(G, a, b) + M(G, b, a)) * 0.5;
		}
	const double eps[6] = {M(E, 0, 0), M(E, 1, 1), M(E, 2, 2), 2 * M(E, 1, 2), 2 * M(E, 0, 2), 2 * M(E, 0, 1)};
	double sg[6];
	for (int j = 0; j < 6; ++j)
	{
		double res = C[j][0] * eps[0];
		for (int k = 1; k < 6; ++k)
			res += C[j][k] * eps[k];
		sg[j] = res;
	}
	M(S, 0, 0) = sg[0], M(S, 0, 1) = sg[5], M(S, 0, 2) = sg[4];
	M(S, 1, 0) = sg[5], M(S, 1, 1) = sg[1], M(S, 1, 2) = sg[3];
	M(S, 2, 0) = sg[4], M(S, 2, 1) = sg[3], M(S, 2, 2) = sg[2];
}

/* the caller's symmetric 6 x 6 tensor (row-major) if there is one, the isotropic one otherwise */
static void sv_pick_tensor(const double *tensor36, double lambda, double mu, double C[6][6])
{
	if (!tensor36)
	{
		sv_tensor(lambda, mu, C);
		return;
	}
	for (int a = 0; a < 6; ++a)
		for (int b = 0; b < 6; ++b)
			C[a][b] = tensor36[6 * a + b];
}

/* SaintVenantElasticity::compute_energy_aux<double> (:215-312): 1/2 sum_p trace(stress * strain) da_p */
double pfo_local_saint_venant_energy_t(const pfo_vals *v, const double *da, const double *local_disp, double lambda, double mu, const double *tensor36)
{
	double C[6][6], energy = 0;
	sv_pick_tensor(tensor36, lambda, mu, C);
	for (int p = 0; p < v->n_pts; ++p)
	{
		double F[9], E[9], S[9];
		def_grad_at(v, p, local_disp, F);
		sv_strain_stress(F, C, E, S);
		double tr = 0;
		for (int a = 0; a < 3; ++a)
			for (int k = 0; k < 3; ++k)
				tr += M(S, a, k) * M(E, k, a);
		energy += tr * da[p];
	}
	return energy * 0.5;
}

/* Voigt strain (engineering shears) of sym(f (x) g) */
static void sv_voigt_sym(const double *f, const double *g, double *e)
{
	e[0] = f[0] * g[0], e[1] = f[1] * g[1], e[2] = f[2] * g[2];
	e[3] = f[1] * g[2] + g[1] * f[2];
	e[4] = f[0] * g[2] + g[0] * f[2];
	e[5] = f[0] * g[1] + g[0] * f[1];
}

/* SaintVenantElasticity::assemble_grad (:104-122) differentiates compute_energy_aux with forward autodiff; written
 * out: d/du_{i,d} = sum_p da_p (F S G_i)[d]  (first Piola-Kirchhoff stress F S, S = C : E symmetric) */
void pfo_local_saint_venant_grad_t(const pfo_vals *v, const double *da, const double *local_disp, double lambda, double mu, const double *tensor36, double *G)
{
	double C[6][6];
	sv_pick_tensor(tensor36, lambda, mu, C);
	for (int k = 0; k < 3 * v->n_loc; ++k)
		G[k] = 0;
	for (int p = 0; p < v->n_pts; ++p)
	{
		double F[9], E[9], S[9], P[9];
		def_grad_at(v, p, local_disp, F);
		sv_strain_stress(F, C, E, S);
		for (int a = 0; a < 3; ++a)
			for (int b = 0; b < 3; ++b)
				M(P, a, b) = M(F, a, 0) * M(S, 0, b) + M(F, a, 1) * M(S, 1, b) + M(F, a, 2) * M(S, 2, b);
		for (int i = 0; i < v->n_loc; ++i)
		{
			double Gi[3];
			phys_grad(v, i, p, Gi);
			for (int d = 0; d < 3; ++d)
				G[3 * i + d] += (M(P, d, 0) * Gi[0] + M(P, d, 1) * Gi[1] + M(P, d, 2) * Gi[2]) * da[p];
		}
	}
}

/* SaintVenantElasticity::assemble_hessian (:124-143, second-order autodiff in the reference), written out: the
 * variation of E along dof (j,b) is sym(f_b (x) G_j) with f_b = row b of F, so
 * d2/du_{i,a} du_{j,b} = sum_p da_p [ delta_ab G_i^T S G_j + voigt(f_a, G_i)^T C voigt(f_b, G_j) ] */
void pfo_local_saint_venant_hessian_t(const pfo_vals *v, const double *da, const double *local_disp, double lambda, double mu, const double *tensor36, double *H)
{
	const int N = 3 * v->n_loc;
	double C[6][6];
	sv_pick_tensor(tensor36, lambda, mu, C);
	for (int k = 0; k < N * N; ++k)
		H[k] = 0;
	for (int p = 0; p < v->n_pts; ++p)
	{
		double F[9], E[9], S[9];
		def_grad_at(v, p, local_disp, F);
		sv_strain_stress(F, C, E, S);
		for (int i = 0; i < v->n_loc; ++i)
		{
			double Gi[3];
			phys_grad(v, i, p, Gi);
			for (int j = 0; j < v->n_loc; ++j)
			{
				double Gj[3], SGj[3];
				phys_grad(v, j, p, Gj);
				for (int a = 0; a < 3; ++a)
					SGj[a] = M(S, a, 0) * Gj[0] + M(S, a, 1) * Gj[1] + M(S, a, 2) * Gj[2];
				const double gsg = Gi[0] * SGj[0] + Gi[1] * SGj[1] + Gi[2] * SGj[2];
				for (int a = 0; a < 3; ++a)
				{
					double va[6];
					sv_voigt_sym(&M(F, a, 0), Gi, va);
					for (int b = 0; b < 3; ++b)
					{
						double vb[6], acc = (a == b) ? gsg : 0.0;
						sv_voigt_sym(&M(F, b, 0), Gj, vb);
						for (int r = 0; r < 6; ++r)
						{
							double cv = 0;
							for (int c = 0; c < 6; ++c)
								cv += C[r][c] * vb[c];
							acc += va[r] * cv;
						}
						H[(3 * i + a) * N + 3 * j + b] += acc * da[p];
					}
				}
			}
		}
	}
}

double pfo_local_saint_venant_energy(const pfo_vals *v, const double *da, const double *local_disp, double lambda, double mu)
{
	return pfo_local_saint_venant_energy_t(v, da, local_disp, lambda, mu, NULL);
}
void pfo_local_saint_venant_grad(const pfo_vals *v, const double *da, const double *local_disp, double lambda, double mu, double *G)
{
	pfo_local_saint_venant_grad_t(v, da, local_disp, lambda, mu, NULL, G);
}
void pfo_local_saint_venant_hessian(const pfo_vals *v, const double *da, const double *local_disp, double lambda, double mu, double *H)
{
	pfo_local_saint_venant_hessian_t(v, da, local_disp, lambda, mu, NULL, H);
}

/* the three nonlinear local assemblers behind one switch */
static int is_nl_model(int model) { return model == PFO_NEOHOOKEAN || model == PFO_LINEAR_ELASTICITY || model == PFO_SAINT_VENANT || model == PFO_MULTI_MODEL; }
/* MultiModel (assembler/MultiModel.cpp:42-96) picks the local assembler by the element id */
static int mat_model(const pfo_material *mat, int64_t e) { return (mat->model == PFO_MULTI_MODEL && mat->model_e) ? mat->model_e[e] : mat->model; }
static double local_linear_elasticity_energy_2d(const pfo_vals *v, const double *da, const double *ld, double lambda, double mu);
static void local_linear_elasticity_grad_2d(const pfo_vals *v, const double *da, const double *ld, double lambda, double mu, double *G);
static void local_linear_elasticity_hessian_2d(const pfo_vals *v, const double *da, double lambda, double mu, double *H);
double pfo_local_saint_venant_energy_2d(const pfo_vals *v, const double *da, const double *local_disp, double lambda, double mu, const double *tensor9);
static double nl_local_energy(int model, const pfo_vals *v, const double *da, const double *ld, double lambda, double mu)
{
	if (v->dim == 2) /* triangles */
	{
		if (model == PFO_SAINT_VENANT)
			return pfo_local_saint_venant_energy_2d(v, da, ld, lambda, mu, NULL);
		return model == PFO_NEOHOOKEAN ? pfo_local_neohookean_energy_2d(v, da, ld, lambda, mu) : local_linear_elasticity_energy_2d(v, da, ld, lambda, mu);
	}
	if (model == PFO_NEOHOOKEAN)
		return pfo_local_neohookean_energy(v, da, ld, lambda, mu);
	if (model == PFO_SAINT_VENANT)
		return pfo_local_saint_venant_energy(v, da, ld, lambda, mu);
	return pfo_local_linear_elasticity_energy(v, da, ld, lambda, mu);
}

static void cross3(const double *x, const double *y, double *z)
{
	z[0] = x[1] * y[2] - x[2] * y[1];
	z[1] = x[2] * y[0] - x[0] * y[2];
	z[2] = x[0] * y[1] - x[1] * y[0];
}

/* delJ_delF: columns cross(v,w), cross(w,u), cross(u,v) with u,v,w the columns of F
 * (NeoHookeanElasticity.cpp:422-432). Row-major output. */
static void delJ_delF3(const double *F, double *dJ)
{
	double u[3], vv[3], w[3], c0[3], c1[3], c2[3];
	for (int r = 0; r < 3; ++r)
		u[r] = M(F, r, 0), vv[r] = M(F, r, 1), w[r] = M(F, r, 2);
	cross3(vv, w, c0);
	cross3(w, u, c1);
	cross3(u, vv, c2);
	for (int r = 0; r < 3; ++r)
		M(dJ, r, 0) = c0[r], M(dJ, r, 1) = c1[r], M(dJ, r, 2) = c2[r];
}

/* NeoHookeanElasticity.cpp:364-453 ; G is n_loc x 3 flattened node-major (G_T.data()) */
void pfo_local_neohookean_grad(const pfo_vals *v, const double *da, const double *local_disp, double lambda, double mu, double *G)
{
	const int n = v->n_loc;
	for (int k = 0; k < 3 * n; ++k)
		G[k] = 0;
	for (int p = 0; p < v->n_pts; ++p)
	{
		double F[9], dJ[9], T[9];
		def_grad_at(v, p, local_disp, F);
		const double J = det3_eigen(F);
		const double log_det_j = log(J);
		delJ_delF3(F, dJ);
		/* gradient_temp = mu F - mu (1/J) dJ + lambda lnJ (1/J) dJ */
		for (int k = 0; k < 9; ++k)
			T[k] = mu * F[k] - mu * (1 / J) * dJ[k] + lambda * log_det_j * (1 / J) * dJ[k];
		const double *jt = v->jac_it[p];
		for (int i = 0; i < n; ++i)
		{
			double dFdU[3]; /* (grad * jac_it).row(i) */
			for (int c = 0; c < 3; ++c)
			{
				double s = 0;
				for (int a = 0; a < 3; ++a)
					s += v->grad[i][p][a] * M(jt, a, c);
				dFdU[c] = s;
			}
			/* gradient(i,d) = sum_c dFdU(i,c) * T^T(c,d) = sum_c dFdU[c] * T(d,c) */
			for (int d = 0; d < 3; ++d)
			{
				double s = 0;
				for (int c = 0; c < 3; ++c)
					s += dFdU[c] * M(T, d, c);
				G[3 * i + d] += s * da[p];
			}
		}
	}
}

/* hat(x), NeoHookeanElasticity.cpp:333-348 */
static void hat3(const double *x, double *h)
{
	for (int k = 0; k < 9; ++k)
		h[k] = 0;
	M(h, 0, 1) = -x[2];
	M(h, 0, 2) = x[1];
	M(h, 1, 0) = x[2];
	M(h, 1, 2) = -x[0];
	M(h, 2, 0) = -x[1];
	M(h, 2, 1) = x[0];
}

/* NeoHookeanElasticity.cpp:455-565.  vec() is Eigen's column-major flattening: index(r,c) = c*3 + r.
 * H is (3n x 3n) row-major, row/col index = i*3 + d. */
void pfo_local_neohookean_hessian(const pfo_vals *v, const double *da, const double *local_disp, double lambda, double mu, double *H)
{
	const int n = v->n_loc;
	const int N = 3 * n;
	for (int k = 0; k < N * N; ++k)
		H[k] = 0;
	double *B = (double *)malloc(sizeof(double) * 9 * N);  /* delF_delU_tensor 9 x N */
	double *HB = (double *)malloc(sizeof(double) * 9 * N); /* hessian_temp * B */
	for (int p = 0; p < v->n_pts; ++p)
	{
		double F[9], dJ[9];
		def_grad_at(v, p, local_disp, F);
		const double J = det3_eigen(F);
		const double log_det_j = log(J);
		delJ_delF3(F, dJ);
		double u[3], vv[3], w[3];
		for (int r = 0; r < 3; ++r)
			u[r] = M(F, r, 0), vv[r] = M(F, r, 1), w[r] = M(F, r, 2);

		double d2[81];
		for (int k = 0; k < 81; ++k)
			d2[k] = 0;
		double hu[9], hv[9], hw[9];
		hat3(u, hu), hat3(vv, hv), hat3(w, hw);
#define D2(r, c) d2[(r)*9 + (c)]
		for (int r = 0; r < 3; ++r)
			for (int c = 0; c < 3; ++c)
			{
				D2(0 + r, 6 + c) = M(hv, r, c);
				D2(6 + r, 0 + c) = -M(hv, r, c);
				D2(0 + r, 3 + c) = -M(hw, r, c);
				D2(3 + r, 0 + c) = M(hw, r, c);
				D2(3 + r, 6 + c) = -M(hu, r, c);
				D2(6 + r, 3 + c) = M(hu, r, c);
			}
		double gj[9]; /* vec(delJ_delF), column-major */
		for (int r = 0; r < 3; ++r)
			for (int c = 0; c < 3; ++c)
				gj[c * 3 + r] = M(dJ, r, c);
		double ht[81];
		const double c1 = (mu + lambda * (1 - log_det_j)) / (J * J);
		const double c2 = (lambda * log_det_j - mu) / (J);
		for (int r = 0; r < 9; ++r)
			for (int c = 0; c < 9; ++c)
				ht[r * 9 + c] = (r == c ? mu : 0.0) + c1 * (gj[r] * gj[c]) + c2 * D2(r, c);
#undef D2
		/* column (i*3+j) of B = vec( e_j (x) (grad.row(i) * jac_it) ) : temp(j, :) = dFdU_i */
		const double *jt = v->jac_it[p];
		for (int k = 0; k < 9 * N; ++k)
			B[k] = 0;
		for (int i = 0; i < n; ++i)
		{
			double dFdU[3];
			for (int c = 0; c < 3; ++c)
			{
				double s = 0;
				for (int a = 0; a < 3; ++a)
					s += v->grad[i][p][a] * M(jt, a, c);
				dFdU[c] = s;
			}
			for (int j = 0; j < 3; ++j)
				for (int c = 0; c < 3; ++c)
					B[(c * 3 + j) * N + (i * 3 + j)] = dFdU[c];
		}
		for (int r = 0; r < 9; ++r)
			for (int c = 0; c < N; ++c)
			{
				double s = 0;
				for (int k = 0; k < 9; ++k)
					s += ht[r * 9 + k] * B[k * N + c];
				HB[r * N + c] = s;
			}
		for (int r = 0; r < N; ++r)
			for (int c = 0; c < N; ++c)
			{
				double s = 0;
				for (int k = 0; k < 9; ++k)
					s += B[k * N + r] * HB[k * N + c];
				H[r * N + c] += s * da[p];
			}
	}
	free(B);
	free(HB);
}

/* ------------------------------------------------------------------------------------------------ */
/* 2-D (triangles): the dim == 2 branches of the local assemblers                                   */
/* ------------------------------------------------------------------------------------------------ */

/* F = local_disp^T * grad(p) * jac_it(p) + id * I, 2 x 2 row-major; local_disp is n_loc x 2 */
static void def_grad_at_2d(const pfo_vals *v, int p, const double *local_disp, double *F, double id)
{
	double A[4];
	for (int d = 0; d < 2; ++d)
		for (int c = 0; c < 2; ++c)
		{
			double s = 0;
			for (int i = 0; i < v->n_loc; ++i)
				s += local_disp[2 * i + d] * v->grad[i][p][c];
			A[2 * d + c] = s;
		}
	const double *jt = v->jac_it[p];
	for (int d = 0; d < 2; ++d)
		for (int c = 0; c < 2; ++c)
		{
			const double s = A[2 * d] * M(jt, 0, c) + A[2 * d + 1] * M(jt, 1, c);
			F[2 * d + c] = (id != 0.0) ? s + (d == c ? id : 0.0) : s;
		}
}

/* NeoHookeanElasticity.cpp:250-331 with size() == 2; determinant of utils/MatrixUtils.hpp:16-30 (2 x 2 branch) */
double pfo_local_neohookean_energy_2d(const pfo_vals *v, const double *da, const double *local_disp, double lambda, double mu)
{
	double energy = 0;
	for (int p = 0; p < v->n_pts; ++p)
	{
		double F[4];
		def_grad_at_2d(v, p, local_disp, F, 1.0);
		const double log_det_j = log(F[0] * F[3] - F[1] * F[2]);
		double tr = 0;
		for (int k = 0; k < 4; ++k)
			tr += F[k] * F[k];
		const double val = mu / 2 * (tr - 2 - 2 * log_det_j) + lambda / 2 * log_det_j * log_det_j;
		energy += val * da[p];
	}
	return energy;
}

/* NeoHookeanElasticity.cpp:364-453, dim == 2 branch (:410-417): delJ_delF = [[F11, -F10], [-F01, F00]].
 * G is n_loc x 2 flattened node-major. */
void pfo_local_neohookean_grad_2d(const pfo_vals *v, const double *da, const double *local_disp, double lambda, double mu, double *G)
{
	const int n = v->n_loc;
	for (int k = 0; k < 2 * n; ++k)
		G[k] = 0;
	for (int p = 0; p < v->n_pts; ++p)
	{
		double F[4], dJ[4], T[4];
		def_grad_at_2d(v, p, local_disp, F, 1.0);
		const double J = F[0] * F[3] - F[2] * F[1]; /* Eigen 2x2 determinant */
		const double log_det_j = log(J);
		dJ[0] = F[3], dJ[1] = -F[2], dJ[2] = -F[1], dJ[3] = F[0];
		for (int k = 0; k < 4; ++k)
			T[k] = mu * F[k] - mu * (1 / J) * dJ[k] + lambda * log_det_j * (1 / J) * dJ[k];
		const double *jt = v->jac_it[p];
		for (int i = 0; i < n; ++i)
		{
			double dFdU[2];
			for (int c = 0; c < 2; ++c)
				dFdU[c] = v->grad[i][p][0] * M(jt, 0, c) + v->grad[i][p][1] * M(jt, 1, c);
			for (int d = 0; d < 2; ++d)
				G[2 * i + d] += (dFdU[0] * T[2 * d] + dFdU[1] * T[2 * d + 1]) * da[p];
		}
	}
}

/* NeoHookeanElasticity.cpp:455-565, dim == 2 branch (:501-512).  vec() is column-major: index(r,c) = c*2 + r;
 * del2J_delF2(0,3) = del2J_delF2(3,0) = 1, (1,2) = (2,1) = -1.  H is (2n x 2n) row-major, index i*2 + d. */
void pfo_local_neohookean_hessian_2d(const pfo_vals *v, const double *da, const double *local_disp, double lambda, double mu, double *H)
{
	const int n = v->n_loc, N = 2 * n;
	for (int k = 0; k < N * N; ++k)
		H[k] = 0;
	double *B = (double *)malloc(sizeof(double) * 4 * N);
	double *HB = (double *)malloc(sizeof(double) * 4 * N);
	for (int p = 0; p < v->n_pts; ++p)
	{
		double F[4];
		def_grad_at_2d(v, p, local_disp, F, 1.0);
		const double J = F[0] * F[3] - F[2] * F[1];
		const double log_det_j = log(J);
		/* delJ_delF row-major [[F11, -F10], [-F01, F00]], flattened column-major */
		const double gj[4] = {F[3], -F[1], -F[2], F[0]};
		double d2[16];
		for (int k = 0; k < 16; ++k)
			d2[k] = 0;
		d2[0 * 4 + 3] = 1, d2[1 * 4 + 2] = -1, d2[2 * 4 + 1] = -1, d2[3 * 4 + 0] = 1;
		const double c1 = (mu + lambda * (1 - log_det_j)) / (J * J);
		const double c2 = (lambda * log_det_j - mu) / (J);
		double ht[16];
		for (int r = 0; r < 4; ++r)
			for (int c = 0; c < 4; ++c)
				ht[r * 4 + c] = (r == c ? mu : 0.0) + c1 * (gj[r] * gj[c]) + c2 * d2[r * 4 + c];
		const double *jt = v->jac_it[p];
		for (int k = 0; k < 4 * N; ++k)
			B[k] = 0;
		for (int i = 0; i < n; ++i)
		{
			double dFdU[2];
			for (int c = 0; c < 2; ++c)
				dFdU[c] = v->grad[i][p][0] * M(jt, 0, c) + v->grad[i][p][1] * M(jt, 1, c);
			/* column (i*2+j) = vec(temp), temp.row(j) = dFdU */
			for (int j = 0; j < 2; ++j)
				for (int c = 0; c < 2; ++c)
					B[(c * 2 + j) * N + (i * 2 + j)] = dFdU[c];
		}
		for (int r = 0; r < 4; ++r)
			for (int c = 0; c < N; ++c)
			{
				double s = 0;
				for (int k = 0; k < 4; ++k)
					s += ht[r * 4 + k] * B[k * N + c];
				HB[r * N + c] = s;
			}
		for (int r = 0; r < N; ++r)
			for (int c = 0; c < N; ++c)
			{
				double s = 0;
				for (int k = 0; k < 4; ++k)
					s += B[k * N + r] * HB[k * N + c];
				H[r * N + c] += s * da[p];
			}
	}
	free(B);
	free(HB);
}

/* LinearElasticity through the nonlinear interface in 2-D (LinearElasticity.cpp:56-169 is written for any size()): the
 * 3-D routines applied to the displacements padded with a zero third component (jac_it and the gradients carry zeros
 * there), in-plane entries taken */
static void pad_disp_2d(const pfo_vals *v, const double *ld2, double *ld3)
{
	for (int i = 0; i < v->n_loc; ++i)
		ld3[3 * i] = ld2[2 * i], ld3[3 * i + 1] = ld2[2 * i + 1], ld3[3 * i + 2] = 0;
}
static double local_linear_elasticity_energy_2d(const pfo_vals *v, const double *da, const double *ld, double lambda, double mu)
{
	double ld3[3 * PFO_MAX_LOC];
	pad_disp_2d(v, ld, ld3);
	return pfo_local_linear_elasticity_energy(v, da, ld3, lambda, mu);
}
static void local_linear_elasticity_grad_2d(const pfo_vals *v, const double *da, const double *ld, double lambda, double mu, double *G)
{
	double ld3[3 * PFO_MAX_LOC], G3[3 * PFO_MAX_LOC];
	pad_disp_2d(v, ld, ld3);
	pfo_local_linear_elasticity_grad(v, da, ld3, lambda, mu, G3);
	for (int i = 0; i < v->n_loc; ++i)
		G[2 * i] = G3[3 * i], G[2 * i + 1] = G3[3 * i + 1];
}
static void local_linear_elasticity_hessian_2d(const pfo_vals *v, const double *da, double lambda, double mu, double *H)
{
	const int n = v->n_loc, N3 = 3 * n, N = 2 * n;
	double *H3 = (double *)malloc(sizeof(double) * (size_t)(N3 * N3));
	pfo_local_linear_elasticity_hessian(v, da, lambda, mu, H3);
	for (int i = 0; i < n; ++i)
		for (int a = 0; a < 2; ++a)
			for (int j = 0; j < n; ++j)
				for (int b = 0; b < 2; ++b)
					H[(2 * i + a) * N + 2 * j + b] = H3[(3 * i + a) * N3 + 3 * j + b];
	free(H3);
}

/* SaintVenantElasticity with size() == 2 (assembler/SaintVenantElasticity.cpp:215-312, the branch at :284-293): the
 * 2 x 2 Green strain, the Voigt strain (xx, yy, 2 xy), the 3 x 3 tensor of ElasticityTensor::resize(2)
 * (utils/ElasticityUtils.cpp:264-274; set_from_lambda_mu :374-384), stress [[s0, s2], [s2, s1]]. */
static void sv_tensor_2d(double lambda, double mu, double C[3][3])
{
	C[0][0] = 2 * mu + lambda, C[0][1] = lambda, C[0][2] = 0;
	C[1][0] = lambda, C[1][1] = 2 * mu + lambda, C[1][2] = 0;
	C[2][0] = 0, C[2][1] = 0, C[2][2] = mu;
}
static void sv_pick_tensor_2d(const double *tensor9, double lambda, double mu, double C[3][3])
{
	if (!tensor9)
	{
		sv_tensor_2d(lambda, mu, C);
		return;
	}
	for (int a = 0; a < 3; ++a)
		for (int b = 0; b < 3; ++b)
			C[a][b] = tensor9[3 * a + b];
}
/* G: the 2 x 2 displacement gradient; strain_from_disp_grad (:15-24) */
static void sv_strain_stress_2d(const double *G, double C[3][3], double *E, double *S)
{
	for (int a = 0; a < 2; ++a)
		for (int b = 0; b < 2; ++b)
		{
			double gtg = 0;
			for (int k = 0; k < 2; ++k)
				gtg += G[2 * k + a] * G[2 * k + b];
			E[2 * a + b] = (gtg + G[2 * a + b] + G[2 * b + a]) * 0.5;
		}
	const double eps[3] = {E[0], E[3], 2 * E[1]};
	double sg[3];
	for (int j = 0; j < 3; ++j)
	{
		double res = C[j][0] * eps[0];
		for (int k = 1; k < 3; ++k)
			res += C[j][k] * eps[k];
		sg[j] = res;
	}
	S[0] = sg[0], S[1] = sg[2], S[2] = sg[2], S[3] = sg[1];
}
double pfo_local_saint_venant_energy_2d(const pfo_vals *v, const double *da, const double *local_disp, double lambda, double mu, const double *tensor9)
{
	double C[3][3], energy = 0;
	sv_pick_tensor_2d(tensor9, lambda, mu, C);
	for (int p = 0; p < v->n_pts; ++p)
	{
		double G[4], E[4], S[4];
		def_grad_at_2d(v, p, local_disp, G, 0.0);
		sv_strain_stress_2d(G, C, E, S);
		double tr = 0;
		for (int a = 0; a < 2; ++a)
			for (int k = 0; k < 2; ++k)
				tr += S[2 * a + k] * E[2 * k + a];
		energy += tr * da[p];
	}
	return energy * 0.5;
}
/* the derivatives the reference takes with autodiff (:104-143), written out as for the 3-D case: the gradient
 * contracts the first Piola-Kirchhoff stress F S with the physical basis gradients, the Hessian is
 * delta_ab G_i^T S G_j + voigt(f_a, G_i)^T C voigt(f_b, G_j) with voigt(f, g) = (f0 g0, f1 g1, f0 g1 + g0 f1) */
void pfo_local_saint_venant_grad_2d(const pfo_vals *v, const double *da, const double *local_disp, double lambda, double mu, const double *tensor9, double *Gout)
{
	double C[3][3];
	sv_pick_tensor_2d(tensor9, lambda, mu, C);
	for (int k = 0; k < 2 * v->n_loc; ++k)
		Gout[k] = 0;
	for (int p = 0; p < v->n_pts; ++p)
	{
		double G[4], F[4], E[4], S[4], P[4];
		def_grad_at_2d(v, p, local_disp, G, 0.0);
		sv_strain_stress_2d(G, C, E, S);
		F[0] = G[0] + 1, F[1] = G[1], F[2] = G[2], F[3] = G[3] + 1;
		for (int a = 0; a < 2; ++a)
			for (int b = 0; b < 2; ++b)
				P[2 * a + b] = F[2 * a] * S[b] + F[2 * a + 1] * S[2 + b];
		for (int i = 0; i < v->n_loc; ++i)
		{
			double Gi[3];
			phys_grad(v, i, p, Gi);
			for (int d = 0; d < 2; ++d)
				Gout[2 * i + d] += (P[2 * d] * Gi[0] + P[2 * d + 1] * Gi[1]) * da[p];
		}
	}
}
void pfo_local_saint_venant_hessian_2d(const pfo_vals *v, const double *da, const double *local_disp, double lambda, double mu, const double *tensor9, double *H)
{
	const int N = 2 * v->n_loc;
	double C[3][3];
	sv_pick_tensor_2d(tensor9, lambda, mu, C);
	for (int k = 0; k < N * N; ++k)
		H[k] = 0;
	for (int p = 0; p < v->n_pts; ++p)
	{
		double G[4], F[4], E[4], S[4];
		def_grad_at_2d(v, p, local_disp, G, 0.0);
		sv_strain_stress_2d(G, C, E, S);
		F[0] = G[0] + 1, F[1] = G[1], F[2] = G[2], F[3] = G[3] + 1;
		for (int i = 0; i < v->n_loc; ++i)
		{
			double Gi[3];
			phys_grad(v, i, p, Gi);
			for (int j = 0; j < v->n_loc; ++j)
			{
				double Gj[3];
				phys_grad(v, j, p, Gj);
				const double gsg = Gi[0] * (S[0] * Gj[0] + S[1] * Gj[1]) + Gi[1] * (S[2] * Gj[0] + S[3] * Gj[1]);
				for (int a = 0; a < 2; ++a)
				{
					const double *fa = F + 2 * a;
					const double va[3] = {fa[0] * Gi[0], fa[1] * Gi[1], fa[0] * Gi[1] + Gi[0] * fa[1]};
					for (int b = 0; b < 2; ++b)
					{
						const double *fb = F + 2 * b;
						const double vb[3] = {fb[0] * Gj[0], fb[1] * Gj[1], fb[0] * Gj[1] + Gj[0] * fb[1]};
						double acc = (a == b) ? gsg : 0.0;
						for (int r = 0; r < 3; ++r)
						{
							double cv = 0;
							for (int c = 0; c < 3; ++c)
								cv += C[r][c] * vb[c];
							acc += va[r] * cv;
						}
						H[(2 * i + a) * N + 2 * j + b] += acc * da[p];
					}
				}
			}
		}
	}
}

/* ------------------------------------------------------------------------------------------------ */
/* triplets -> CSC  (Eigen::SparseMatrix::setFromTriplets: duplicates summed in insertion order,     */
/* explicit zeros kept, inner indices ascending)                                                     */
/* ------------------------------------------------------------------------------------------------ */

typedef struct
{
	int32_t r, c;
	double v;
} trip;

typedef struct
{
	trip *t;
	int64_t n, cap;
} tripvec;

static void tv_push(tripvec *tv, int32_t r, int32_t c, double v)
{
	if (tv->n == tv->cap)
	{
		tv->cap = tv->cap ? tv->cap * 2 : (1 << 16);
		tv->t = (trip *)realloc(tv->t, sizeof(trip) * tv->cap);
	}
	tv->t[tv->n].r = r;
	tv->t[tv->n].c = c;
	tv->t[tv->n].v = v;
	tv->n++;
}

/* Eigen/src/SparseCore/SparseMatrix.h set_from_triplets: pass 1 builds the transposed (row-major) matrix
 * by counting + insertion in triplet order, collapses duplicates per row in insertion order, pass 2
 * transposes back, which sorts inner indices. */
static int trips_to_csc(int32_t n, const trip *t, int64_t nt, pfo_csc *out)
{
	int64_t *rptr = (int64_t *)calloc((size_t)n + 1, sizeof(int64_t));
	for (int64_t k = 0; k < nt; ++k)
		rptr[t[k].r + 1]++;
	for (int32_t r = 0; r < n; ++r)
		rptr[r + 1] += rptr[r];
	int32_t *rc = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nt ? nt : 1));
	double *rv = (double *)malloc(sizeof(double) * (size_t)(nt ? nt : 1));
	int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * ((size_t)n + 1));
	memcpy(cur, rptr, sizeof(int64_t) * ((size_t)n + 1));
	for (int64_t k = 0; k < nt; ++k)
	{
		const int64_t pos = cur[t[k].r]++;
		rc[pos] = t[k].c;
		rv[pos] = t[k].v;
	}
	/* collapse duplicates row by row, first occurrence keeps its place */
	int64_t *wi = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
	for (int32_t c = 0; c < n; ++c)
		wi[c] = -1;
	int64_t *ccount = (int64_t *)calloc((size_t)n + 1, sizeof(int64_t));
	int64_t count = 0;
	int64_t *rstart = (int64_t *)malloc(sizeof(int64_t) * ((size_t)n + 1));
	for (int32_t r = 0; r < n; ++r)
	{
		const int64_t start = count;
		rstart[r] = start;
		for (int64_t k = rptr[r]; k < rptr[r + 1]; ++k)
		{
			const int32_t c = rc[k];
			if (wi[c] >= start)
				rv[wi[c]] += rv[k];
			else
			{
				rv[count] = rv[k];
				rc[count] = c;
				wi[c] = count;
				++count;
				ccount[c + 1]++;
			}
		}
	}
	rstart[n] = count;
	out->n = n;
	out->nnz = count;
	out->outer = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n + 1));
	out->inner = (int32_t *)malloc(sizeof(int32_t) * (size_t)(count ? count : 1));
	out->values = (double *)malloc(sizeof(double) * (size_t)(count ? count : 1));
	for (int32_t c = 0; c < n; ++c)
		ccount[c + 1] += ccount[c];
	for (int32_t c = 0; c <= n; ++c)
		out->outer[c] = (int32_t)ccount[c];
	memcpy(cur, ccount, sizeof(int64_t) * ((size_t)n + 1));
	for (int32_t r = 0; r < n; ++r)
		for (int64_t k = rstart[r]; k < rstart[r + 1]; ++k)
		{
			const int64_t pos = cur[rc[k]]++;
			out->inner[pos] = r;
			out->values[pos] = rv[k];
		}
	free(rptr), free(rc), free(rv), free(cur), free(wi), free(ccount), free(rstart);
	return 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* global assemblers                                                                                */
/* ------------------------------------------------------------------------------------------------ */

/* LameParameters::lambda_mu (utils/ElasticityUtils.cpp:490-517): one value, or one per element id */
static double mat_lambda(const pfo_material *mat, int64_t e) { return mat->lambda_e ? mat->lambda_e[e] : mat->lambda; }
static double mat_mu(const pfo_material *mat, int64_t e) { return mat->mu_e ? mat->mu_e[e] : mat->mu; }

/* Materials that vary from quadrature point to quadrature point.  Every local quantity is a sum over the points of
 * terms weighted with da[q], so it is evaluated point by point -- all other weights set to zero, which adds exact zeros
 * -- with that point's (lambda, mu), and the per-point results are added in point order: the additions the reference's
 * fused loop performs with lambda_mu() called inside it. */
static int mat_points(const pfo_material *mat) { return (mat->per_point > 0 && mat->lambda_e && mat->mu_e) ? mat->per_point : 0; }
static void one_point(const double *da, int n, int q, double *out)
{
	for (int k = 0; k < n; ++k)
		out[k] = k == q ? da[k] : 0.0;
}
static void local_linear_elasticity_mat(const pfo_vals *v, int i, int j, const double *da, const pfo_material *mat, int64_t e, double *res)
{
	const int np = mat_points(mat);
	if (!np)
	{
		pfo_local_linear_elasticity(v, i, j, da, mat_lambda(mat, e), mat_mu(mat, e), res);
		return;
	}
	double daq[PFO_MAX_PTS], part[9];
	for (int c = 0; c < 9; ++c)
		res[c] = 0;
	for (int q = 0; q < v->n_pts; ++q)
	{
		one_point(da, v->n_pts, q, daq);
		pfo_local_linear_elasticity(v, i, j, daq, mat->lambda_e[e * np + q], mat->mu_e[e * np + q], part);
		for (int c = 0; c < 9; ++c)
			res[c] += part[c];
	}
}
static double nl_local_energy(int model, const pfo_vals *v, const double *da, const double *ld, double lambda, double mu);
static double nl_local_energy_mat(int model, const pfo_vals *v, const double *da, const double *ld, const pfo_material *mat, int64_t e)
{
	if (model == PFO_SAINT_VENANT && mat->tensor)
		return v->dim == 2 ? pfo_local_saint_venant_energy_2d(v, da, ld, 0, 0, mat->tensor) : pfo_local_saint_venant_energy_t(v, da, ld, 0, 0, mat->tensor);
	const int np = mat_points(mat);
	if (!np)
		return nl_local_energy(model, v, da, ld, mat_lambda(mat, e), mat_mu(mat, e));
	double daq[PFO_MAX_PTS], sum = 0;
	for (int q = 0; q < v->n_pts; ++q)
	{
		one_point(da, v->n_pts, q, daq);
		sum += nl_local_energy(model, v, daq, ld, mat->lambda_e[e * np + q], mat->mu_e[e * np + q]);
	}
	return sum;
}
static void nl_local_grad(int model, const pfo_vals *v, const double *da, const double *ld, double lambda, double mu, double *G)
{
	if (v->dim == 2)
	{
		if (model == PFO_NEOHOOKEAN)
			pfo_local_neohookean_grad_2d(v, da, ld, lambda, mu, G);
		else if (model == PFO_SAINT_VENANT)
			pfo_local_saint_venant_grad_2d(v, da, ld, lambda, mu, NULL, G);
		else
			local_linear_elasticity_grad_2d(v, da, ld, lambda, mu, G);
		return;
	}
	if (model == PFO_NEOHOOKEAN)
		pfo_local_neohookean_grad(v, da, ld, lambda, mu, G);
	else if (model == PFO_SAINT_VENANT)
		pfo_local_saint_venant_grad(v, da, ld, lambda, mu, G);
	else
		pfo_local_linear_elasticity_grad(v, da, ld, lambda, mu, G);
}
static void nl_local_hessian(int model, const pfo_vals *v, const double *da, const double *ld, double lambda, double mu, double *H)
{
	if (v->dim == 2)
	{
		if (model == PFO_NEOHOOKEAN)
			pfo_local_neohookean_hessian_2d(v, da, ld, lambda, mu, H);
		else if (model == PFO_SAINT_VENANT)
			pfo_local_saint_venant_hessian_2d(v, da, ld, lambda, mu, NULL, H);
		else
			local_linear_elasticity_hessian_2d(v, da, lambda, mu, H);
		return;
	}
	if (model == PFO_NEOHOOKEAN)
		pfo_local_neohookean_hessian(v, da, ld, lambda, mu, H);
	else if (model == PFO_SAINT_VENANT)
		pfo_local_saint_venant_hessian(v, da, ld, lambda, mu, H);
	else
		pfo_local_linear_elasticity_hessian(v, da, lambda, mu, H);
}
static void nl_local_grad_mat(int model, const pfo_vals *v, const double *da, const double *ld, const pfo_material *mat, int64_t e, double *G)
{
	if (model == PFO_SAINT_VENANT && mat->tensor)
	{
		if (v->dim == 2)
			pfo_local_saint_venant_grad_2d(v, da, ld, 0, 0, mat->tensor, G);
		else
			pfo_local_saint_venant_grad_t(v, da, ld, 0, 0, mat->tensor, G);
		return;
	}
	const int np = mat_points(mat), N = v->n_loc * v->dim;
	if (!np)
	{
		nl_local_grad(model, v, da, ld, mat_lambda(mat, e), mat_mu(mat, e), G);
		return;
	}
	double daq[PFO_MAX_PTS], part[3 * PFO_MAX_LOC];
	for (int c = 0; c < N; ++c)
		G[c] = 0;
	for (int q = 0; q < v->n_pts; ++q)
	{
		one_point(da, v->n_pts, q, daq);
		nl_local_grad(model, v, daq, ld, mat->lambda_e[e * np + q], mat->mu_e[e * np + q], part);
		for (int c = 0; c < N; ++c)
			G[c] += part[c];
	}
}
static void nl_local_hessian_mat(int model, const pfo_vals *v, const double *da, const double *ld, const pfo_material *mat, int64_t e, double *H)
{
	if (model == PFO_SAINT_VENANT && mat->tensor)
	{
		if (v->dim == 2)
			pfo_local_saint_venant_hessian_2d(v, da, ld, 0, 0, mat->tensor, H);
		else
			pfo_local_saint_venant_hessian_t(v, da, ld, 0, 0, mat->tensor, H);
		return;
	}
	const int np = mat_points(mat), N = v->n_loc * v->dim;
	if (!np)
	{
		nl_local_hessian(model, v, da, ld, mat_lambda(mat, e), mat_mu(mat, e), H);
		return;
	}
	double daq[PFO_MAX_PTS];
	double *part = (double *)malloc(sizeof(double) * (size_t)(N * N));
	for (int c = 0; c < N * N; ++c)
		H[c] = 0;
	for (int q = 0; q < v->n_pts; ++q)
	{
		one_point(da, v->n_pts, q, daq);
		nl_local_hessian(model, v, daq, ld, mat->lambda_e[e * np + q], mat->mu_e[e * np + q], part);
		for (int c = 0; c < N * N; ++c)
			H[c] += part[c];
	}
	free(part);
}

static int block_size(const pfo_material *mat) { return mat->model == PFO_LAPLACIAN ? 1 : mat->dim; }
/* triangles: dim == 2 (a 2-D elasticity tensor is 3 x 3); every other mesh: dim == 3 */
static int model_fits_mesh(const pfo_mesh *mesh, const pfo_material *mat)
{
	if (!is_tri(mesh->kind))
		return mat->model == PFO_LAPLACIAN || mat->dim == 3;
	if (mat->model == PFO_LAPLACIAN)
		return 1;
	return mat->dim == 2;
}

typedef struct
{
	const pfo_mesh *mesh;
	const pfo_material *mat;
	const double *x;
	int64_t e0, e1;
	tripvec tv;
	double scalar;
	double *vec;
	int mode; /* 0 linear, 1 energy, 2 grad, 3 hessian, 4 hessian projected to PSD */
} job;

void pfo_project_to_psd(double *a, int n)
{
	double *w = (double *)malloc(sizeof(double) * (size_t)n * n * 2);
	double *A = w, *V = w + (size_t)n * n;
	double norm2 = 0;
	for (int i = 0; i < n; ++i)
		for (int j = 0; j < n; ++j)
		{
			/* the eigensolver reads the lower triangle only */
			A[i * n + j] = (j <= i) ? a[i * n + j] : a[j * n + i];
			V[i * n + j] = (i == j) ? 1.0 : 0.0;
			norm2 += A[i * n + j] * A[i * n + j];
		}
	for (int sweep = 0; sweep < 60; ++sweep)
	{
		double off = 0;
		for (int p = 0; p < n; ++p)
			for (int q = p + 1; q < n; ++q)
				off += A[p * n + q] * A[p * n + q];
		if (off <= 1e-32 * norm2)
			break;
		for (int p = 0; p < n; ++p)
			for (int q = p + 1; q < n; ++q)
			{
				const double apq = A[p * n + q];
				if (apq == 0.0)
					continue;
				const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
				const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
				const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
				for (int k = 0; k < n; ++k)
				{
					const double akp = A[k * n + p], akq = A[k * n + q];
					A[k * n + p] = c * akp - sn * akq;
					A[k * n + q] = sn * akp + c * akq;
				}
				for (int k = 0; k < n; ++k)
				{
					const double apk = A[p * n + k], aqk = A[q * n + k];
					A[p * n + k] = c * apk - sn * aqk;
					A[q * n + k] = sn * apk + c * aqk;
				}
				for (int k = 0; k < n; ++k)
				{
					const double vkp = V[k * n + p], vkq = V[k * n + q];
					V[k * n + p] = c * vkp - sn * vkq;
					V[k * n + q] = sn * vkp + c * vkq;
				}
			}
	}
	double lmin = A[0];
	for (int k = 1; k < n; ++k)
		if (A[k * n + k] < lmin)
			lmin = A[k * n + k];
	if (!(lmin >= 0.0))
		for (int i = 0; i < n; ++i)
			for (int j = 0; j < n; ++j)
			{
				double s = 0;
				for (int k = 0; k < n; ++k)
				{
					const double l = A[k * n + k];
					if (l > 0.0)
						s += l * V[i * n + k] * V[j * n + k];
				}
				a[i * n + j] = s;
			}
	free(w);
}

static void gather_local_disp(const pfo_mesh *mesh, int64_t e, int n_loc, const double *x, double *ld)
{
	/* NeoHookeanElasticity.cpp:371-383 ; global list of size 1 with val 1.0 */
	const int size = pfo_kind_dim(mesh->kind);
	for (int i = 0; i < n_loc; ++i)
		for (int d = 0; d < size; ++d)
			ld[size * i + d] = 0 + 1.0 * x[(int64_t)mesh->conn[e * n_loc + i] * size + d];
}

static void *job_run(void *arg)
{
	job *jb = (job *)arg;
	const pfo_mesh *mesh = jb->mesh;
	const pfo_material *mat = jb->mat;
	const int n_loc = pfo_n_loc(mesh->kind);
	const int size = block_size(mat);
	pfo_quad quad;
	pfo_element_quadrature(mesh->kind, mesh->quad_order, &quad);
	pfo_vals *v = (pfo_vals *)malloc(sizeof(pfo_vals));
	double da[PFO_MAX_PTS];
	double *H = (double *)malloc(sizeof(double) * 81 * 81);
	double ld[3 * PFO_MAX_LOC], G[3 * PFO_MAX_LOC];

	for (int64_t e = jb->e0; e < jb->e1; ++e)
	{
		pfo_compute_vals(mesh, e, &quad, v);
		for (int k = 0; k < quad.n_pts; ++k)
			da[k] = v->det[k] * quad.w[k];
		const int32_t *idx = mesh->conn + e * n_loc;

		if (jb->mode == 0)
		{
			/* Assembler.cpp:149-199 */
			for (int i = 0; i < n_loc; ++i)
				for (int j = 0; j <= i; ++j)
				{
					double sv[9];
					if (mat->model == PFO_LAPLACIAN)
						pfo_local_laplacian(v, i, j, da, sv);
					else if (mat->model == PFO_HOOKE_LINEAR && size == 2)
						pfo_local_hooke_2d(v, i, j, da, mat->tensor, sv); /* the 3 x 3 tensor; res index jj*2+ii already */
					else if (mat->model == PFO_HOOKE_LINEAR)
						pfo_local_hooke(v, i, j, da, mat->tensor, sv);
					else
						local_linear_elasticity_mat(v, i, j, da, mat, e, sv);
					if (size == 2 && mat->model != PFO_HOOKE_LINEAR) /* LinearElasticity::assemble with size() == 2: the in-plane entries, index jj*2+ii (the third components of the gradients are zero) */
						sv[2] = sv[3], sv[3] = sv[4];
					for (int n = 0; n < size; ++n)
						for (int m = 0; m < size; ++m)
						{
							const double local_value = sv[n * size + m];
							if (fabs(local_value) < 1e-30)
								continue;
							const int32_t gi = idx[i] * size + m;
							const int32_t gj = idx[j] * size + n;
							tv_push(&jb->tv, gi, gj, local_value * 1.0 * 1.0);
							if (j < i)
								tv_push(&jb->tv, gj, gi, local_value * 1.0 * 1.0);
						}
				}
		}
		else
		{
			gather_local_disp(mesh, e, n_loc, jb->x, ld);
			if (jb->mode == 1)
			{
				/* Assembler.cpp:657-668 */
				const double val = nl_local_energy_mat(mat_model(mat, e), v, da, ld, mat, e);
				jb->scalar += val;
			}
			else if (jb->mode == 2)
			{
				/* Assembler.cpp:465-492 */
				nl_local_grad_mat(mat_model(mat, e), v, da, ld, mat, e, G);
				for (int j = 0; j < n_loc; ++j)
					for (int m = 0; m < size; ++m)
					{
						const double local_value = G[j * size + m];
						if (fabs(local_value) < 1e-30)
							continue;
						jb->vec[(int64_t)idx[j] * size + m] += local_value * 1.0;
					}
			}
			else
			{
				/* Assembler.cpp:550-616 : every entry, order (i, j, n, m) */
				nl_local_hessian_mat(mat_model(mat, e), v, da, ld, mat, e, H);
				const int N = n_loc * size;
				if (jb->mode == 4) /* Assembler.cpp:554-555 */
					pfo_project_to_psd(H, N);
				for (int i = 0; i < n_loc; ++i)
					for (int j = 0; j < n_loc; ++j)
						for (int n = 0; n < size; ++n)
							for (int m = 0; m < size; ++m)
							{
								const double local_value = H[(i * size + m) * N + (j * size + n)];
								tv_push(&jb->tv, idx[i] * size + m, idx[j] * size + n, local_value * 1.0 * 1.0);
							}
			}
		}
	}
	free(v);
	free(H);
	return NULL;
}

/* Static contiguous chunks per thread: utils/par_for.cpp:10-26 (the reference's TBB blocked_range is
 * also contiguous per task).  Thread-local storages are merged in thread order afterwards. */
static void run_jobs(job *jobs, int nt)
{
	if (nt == 1)
	{
		job_run(&jobs[0]);
		return;
	}
	pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nt);
	for (int t = 0; t < nt; ++t)
		pthread_create(&th[t], NULL, job_run, &jobs[t]);
	for (int t = 0; t < nt; ++t)
		pthread_join(th[t], NULL);
	free(th);
}

static job *make_jobs(const pfo_mesh *mesh, const pfo_material *mat, const double *x, int mode, int *nt_io, int64_t vec_len)
{
	int nt = *nt_io < 1 ? 1 : *nt_io;
	if (nt > mesh->n_el)
		nt = mesh->n_el > 0 ? (int)mesh->n_el : 1;
	*nt_io = nt;
	job *jobs = (job *)calloc(nt, sizeof(job));
	const int64_t chunk = (mesh->n_el + nt - 1) / nt;
	for (int t = 0; t < nt; ++t)
	{
		jobs[t].mesh = mesh;
		jobs[t].mat = mat;
		jobs[t].x = x;
		jobs[t].mode = mode;
		jobs[t].e0 = t * chunk < mesh->n_el ? t * chunk : mesh->n_el;
		jobs[t].e1 = (t + 1) * chunk < mesh->n_el ? (t + 1) * chunk : mesh->n_el;
		if (vec_len)
			jobs[t].vec = (double *)calloc((size_t)vec_len, sizeof(double)); /* LocalThreadVecStorage(rhs.size()), Assembler.cpp:442 */
	}
	return jobs;
}

static int merge_trips(job *jobs, int nt, int32_t n, pfo_csc *out)
{
	/* Assembler.cpp:235-301 : concatenate the thread-local triplets, then setFromTriplets */
	int64_t total = 0;
	for (int t = 0; t < nt; ++t)
		total += jobs[t].tv.n;
	trip *all;
	if (nt == 1)
		all = jobs[0].tv.t;
	else
	{
		all = (trip *)malloc(sizeof(trip) * (size_t)(total ? total : 1));
		int64_t off = 0;
		for (int t = 0; t < nt; ++t)
		{
			memcpy(all + off, jobs[t].tv.t, sizeof(trip) * (size_t)jobs[t].tv.n);
			off += jobs[t].tv.n;
			free(jobs[t].tv.t);
		}
	}
	const int rc = trips_to_csc(n, all, total, out);
	free(all);
	return rc;
}

/* assembler/Assembler.cpp:100-317 */
int pfo_assemble_linear(const pfo_mesh *mesh, const pfo_material *mat, int n_threads, pfo_csc *out)
{
	if (mat->model != PFO_LAPLACIAN && mat->model != PFO_LINEAR_ELASTICITY && !(mat->model == PFO_HOOKE_LINEAR && mat->tensor))
		return -1;
	if (!model_fits_mesh(mesh, mat))
		return -1;
	int nt = n_threads;
	job *jobs = make_jobs(mesh, mat, NULL, 0, &nt, 0);
	run_jobs(jobs, nt);
	const int rc = merge_trips(jobs, nt, (int32_t)(mesh->n_nodes * block_size(mat)), out);
	free(jobs);
	return rc;
}

/* assembler/Assembler.cpp:638-678 */
double pfo_assemble_energy(const pfo_mesh *mesh, const pfo_material *mat, const double *x, int n_threads)
{
	if (!model_fits_mesh(mesh, mat))
		return NAN;
	int nt = n_threads;
	job *jobs = make_jobs(mesh, mat, x, 1, &nt, 0);
	run_jobs(jobs, nt);
	double res = 0;
	for (int t = 0; t < nt; ++t)
		res += jobs[t].scalar;
	free(jobs);
	return res;
}

/* assembler/Assembler.cpp:427-504 */
int pfo_assemble_gradient(const pfo_mesh *mesh, const pfo_material *mat, const double *x, int n_threads, double *rhs)
{
	if (!is_nl_model(mat->model) || !model_fits_mesh(mesh, mat))
		return -1;
	const int64_t N = mesh->n_nodes * block_size(mat);
	int nt = n_threads;
	job *jobs = make_jobs(mesh, mat, x, 2, &nt, N);
	run_jobs(jobs, nt);
	for (int64_t k = 0; k < N; ++k)
		rhs[k] = 0;
	for (int t = 0; t < nt; ++t)
	{
		for (int64_t k = 0; k < N; ++k)
			rhs[k] += jobs[t].vec[k];
		free(jobs[t].vec);
	}
	free(jobs);
	return 0;
}

/* assembler/Assembler.cpp:506-636, first (pattern-building) pass: triplets of every local entry */
int pfo_assemble_hessian(const pfo_mesh *mesh, const pfo_material *mat, const double *x, int n_threads, pfo_csc *out)
{
	if (!is_nl_model(mat->model) || !model_fits_mesh(mesh, mat))
		return -1;
	int nt = n_threads;
	job *jobs = make_jobs(mesh, mat, x, 3, &nt, 0);
	run_jobs(jobs, nt);
	const int rc = merge_trips(jobs, nt, (int32_t)(mesh->n_nodes * block_size(mat)), out);
	free(jobs);
	return rc;
}

int pfo_assemble_hessian_psd(const pfo_mesh *mesh, const pfo_material *mat, const double *x, int n_threads, pfo_csc *out)
{
	if (!is_nl_model(mat->model) || !model_fits_mesh(mesh, mat))
		return -1;
	int nt = n_threads;
	job *jobs = make_jobs(mesh, mat, x, 4, &nt, 0);
	run_jobs(jobs, nt);
	const int rc = merge_trips(jobs, nt, (int32_t)(mesh->n_nodes * block_size(mat)), out);
	free(jobs);
	return rc;
}

/* ------------------------------------------------------------------------------------------------ */
/* SpareMatrixCache (utils/MatrixUtils.cpp:32-357)                                                  */
/* ------------------------------------------------------------------------------------------------ */

typedef struct
{
	int32_t col;
	int64_t index;
} mapent;

struct pfo_cache
{
	int64_t size;
	tripvec entries;     /* entries_                                                   */
	pfo_csc mat;         /* mat_ (only used before the mapping exists)                  */
	int has_mat;
	/* mapping_[row] = list of (col, value index) */
	mapent **mapping;
	int32_t *mapping_len;
	int64_t mapping_rows; /* 0 = empty */
	int32_t *inner_index, *outer_index;
	double *values;
	int64_t n_values;
	const pfo_cache *main_cache;
	/* second cache */
	int32_t **second_cache;
	int64_t *second_cache_len;
	int64_t second_cache_n;
	/* second_cache_entries_ : per element list of (i,j) */
	int32_t **sce;
	int64_t *sce_len, *sce_cap;
	int64_t sce_n;
	int current_e, current_e_index;
};

static const pfo_cache *cache_main(const pfo_cache *c) { return c->main_cache ? c->main_cache : c; }
int64_t pfo_cache_mapping_size(const pfo_cache *c) { return cache_main(c)->mapping_rows; }

pfo_cache *pfo_cache_create(int64_t size)
{
	pfo_cache *c = (pfo_cache *)calloc(1, sizeof(pfo_cache));
	c->size = size;
	c->current_e = -1;
	c->current_e_index = -1;
	return c;
}

/* copy constructor -> init(other): MatrixUtils.cpp:73-90 */
pfo_cache *pfo_cache_clone(const pfo_cache *other)
{
	pfo_cache *c = pfo_cache_create(other->size);
	c->main_cache = other->main_cache ? other->main_cache : other;
	c->n_values = other->n_values;
	c->values = (double *)calloc((size_t)(c->n_values ? c->n_values : 1), sizeof(double));
	return c;
}

void pfo_cache_destroy(pfo_cache *c)
{
	if (!c)
		return;
	free(c->entries.t);
	if (c->has_mat)
		pfo_csc_free(&c->mat);
	for (int64_t r = 0; r < c->mapping_rows; ++r)
		free(c->mapping[r]);
	free(c->mapping);
	free(c->mapping_len);
	free(c->inner_index);
	free(c->outer_index);
	free(c->values);
	for (int64_t e = 0; e < c->second_cache_n; ++e)
		free(c->second_cache[e]);
	free(c->second_cache);
	free(c->second_cache_len);
	for (int64_t e = 0; e < c->sce_n; ++e)
		free(c->sce[e]);
	free(c->sce);
	free(c->sce_len);
	free(c->sce_cap);
	free(c);
}

/* MatrixUtils.cpp:92-98 */
void pfo_cache_set_zero(pfo_cache *c)
{
	if (c->has_mat)
	{
		pfo_csc_free(&c->mat);
		c->has_mat = 0;
	}
	for (int64_t k = 0; k < c->n_values; ++k)
		c->values[k] = 0;
}

/* MatrixUtils.cpp:108-151 */
void pfo_cache_add_value(pfo_cache *c, int e, int i, int j, double value)
{
	const pfo_cache *mc = cache_main(c);
	if (mc->mapping_rows == 0)
	{
		tv_push(&c->entries, i, j, value);
		if (c->sce_n <= e)
		{
			const int64_t nn = e + 1;
			c->sce = (int32_t **)realloc(c->sce, sizeof(int32_t *) * nn);
			c->sce_len = (int64_t *)realloc(c->sce_len, sizeof(int64_t) * nn);
			c->sce_cap = (int64_t *)realloc(c->sce_cap, sizeof(int64_t) * nn);
			for (int64_t k = c->sce_n; k < nn; ++k)
				c->sce[k] = NULL, c->sce_len[k] = 0, c->sce_cap[k] = 0;
			c->sce_n = nn;
		}
		if (c->sce_len[e] == c->sce_cap[e])
		{
			c->sce_cap[e] = c->sce_cap[e] ? 2 * c->sce_cap[e] : 16;
			c->sce[e] = (int32_t *)realloc(c->sce[e], sizeof(int32_t) * 2 * c->sce_cap[e]);
		}
		c->sce[e][2 * c->sce_len[e]] = i;
		c->sce[e][2 * c->sce_len[e] + 1] = j;
		c->sce_len[e]++;
	}
	else
	{
		if (e != c->current_e)
		{
			c->current_e = e;
			c->current_e_index = 0;
		}
		c->values[mc->second_cache[e][c->current_e_index]] += value;
		c->current_e_index++;
	}
}

/* sparse a += b for two CSC matrices with sorted inner indices (Eigen's operator+= result pattern is the
 * sorted union) */
static void csc_add_inplace(pfo_csc *a, const pfo_csc *b)
{
	const int32_t n = a->n;
	int32_t *outer = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n + 1));
	int32_t *inner = (int32_t *)malloc(sizeof(int32_t) * (size_t)(a->nnz + b->nnz + 1));
	double *vals = (double *)malloc(sizeof(double) * (size_t)(a->nnz + b->nnz + 1));
	int64_t cnt = 0;
	for (int32_t c = 0; c < n; ++c)
	{
		outer[c] = (int32_t)cnt;
		int32_t ia = a->outer[c], ib = b->outer[c];
		const int32_t ea = a->outer[c + 1], eb = b->outer[c + 1];
		while (ia < ea || ib < eb)
		{
			if (ib >= eb || (ia < ea && a->inner[ia] < b->inner[ib]))
				inner[cnt] = a->inner[ia], vals[cnt] = a->values[ia], ++ia;
			else if (ia >= ea || b->inner[ib] < a->inner[ia])
				inner[cnt] = b->inner[ib], vals[cnt] = b->values[ib], ++ib;
			else
				inner[cnt] = a->inner[ia], vals[cnt] = a->values[ia] + b->values[ib], ++ia, ++ib;
			++cnt;
		}
	}
	outer[n] = (int32_t)cnt;
	free(a->outer), free(a->inner), free(a->values);
	a->outer = outer, a->inner = inner, a->values = vals, a->nnz = cnt;
}

/* MatrixUtils.cpp:153-169 */
void pfo_cache_prune(pfo_cache *c)
{
	if (cache_main(c)->mapping_rows != 0)
		return;
	pfo_csc tmp;
	trips_to_csc((int32_t)c->size, c->entries.t, c->entries.n, &tmp);
	if (!c->has_mat)
	{
		c->mat = tmp;
		c->has_mat = 1;
	}
	else
	{
		csc_add_inplace(&c->mat, &tmp);
		pfo_csc_free(&tmp);
	}
	c->entries.n = 0;
}

/* MatrixUtils.cpp:319-357 */
void pfo_cache_add(pfo_cache *c, const pfo_cache *o)
{
	if (cache_main(c)->mapping_rows == 0 || cache_main(o)->mapping_rows == 0)
	{
		if (o->has_mat)
		{
			if (!c->has_mat)
			{
				trips_to_csc((int32_t)c->size, NULL, 0, &c->mat);
				c->has_mat = 1;
			}
			csc_add_inplace(&c->mat, &o->mat);
		}
		for (int64_t e = 0; e < o->sce_n; ++e)
			for (int64_t k = 0; k < o->sce_len[e]; ++k)
			{
				/* append to second_cache_entries_[e] */
				if (c->sce_n <= e)
				{
					const int64_t nn = e + 1;
					c->sce = (int32_t **)realloc(c->sce, sizeof(int32_t *) * nn);
					c->sce_len = (int64_t *)realloc(c->sce_len, sizeof(int64_t) * nn);
					c->sce_cap = (int64_t *)realloc(c->sce_cap, sizeof(int64_t) * nn);
					for (int64_t q = c->sce_n; q < nn; ++q)
						c->sce[q] = NULL, c->sce_len[q] = 0, c->sce_cap[q] = 0;
					c->sce_n = nn;
				}
				if (c->sce_len[e] == c->sce_cap[e])
				{
					c->sce_cap[e] = c->sce_cap[e] ? 2 * c->sce_cap[e] : 16;
					c->sce[e] = (int32_t *)realloc(c->sce[e], sizeof(int32_t) * 2 * c->sce_cap[e]);
				}
				c->sce[e][2 * c->sce_len[e]] = o->sce[e][2 * k];
				c->sce[e][2 * c->sce_len[e] + 1] = o->sce[e][2 * k + 1];
				c->sce_len[e]++;
			}
	}
	else
	{
		for (int64_t k = 0; k < o->n_values; ++k)
			c->values[k] += o->values[k];
	}
}

static void csc_copy(const pfo_csc *src, pfo_csc *dst)
{
	dst->n = src->n;
	dst->nnz = src->nnz;
	dst->outer = (int32_t *)malloc(sizeof(int32_t) * ((size_t)src->n + 1));
	dst->inner = (int32_t *)malloc(sizeof(int32_t) * (size_t)(src->nnz ? src->nnz : 1));
	dst->values = (double *)malloc(sizeof(double) * (size_t)(src->nnz ? src->nnz : 1));
	memcpy(dst->outer, src->outer, sizeof(int32_t) * ((size_t)src->n + 1));
	memcpy(dst->inner, src->inner, sizeof(int32_t) * (size_t)src->nnz);
	memcpy(dst->values, src->values, sizeof(double) * (size_t)src->nnz);
}

/* MatrixUtils.cpp:171-264 */
int pfo_cache_get_matrix(pfo_cache *c, int compute_mapping, pfo_csc *out)
{
	pfo_cache_prune(c);
	const pfo_cache *mc = cache_main(c);
	if (mc->mapping_rows == 0)
	{
		if (!c->has_mat)
		{
			trips_to_csc((int32_t)c->size, NULL, 0, &c->mat);
			c->has_mat = 1;
		}
		if (compute_mapping && c->size > 0)
		{
			const pfo_csc *m = &c->mat;
			c->n_values = m->nnz;
			c->values = (double *)calloc((size_t)(m->nnz ? m->nnz : 1), sizeof(double));
			c->inner_index = (int32_t *)malloc(sizeof(int32_t) * (size_t)(m->nnz ? m->nnz : 1));
			c->outer_index = (int32_t *)malloc(sizeof(int32_t) * ((size_t)m->n + 1));
			memcpy(c->inner_index, m->inner, sizeof(int32_t) * (size_t)m->nnz);
			memcpy(c->outer_index, m->outer, sizeof(int32_t) * ((size_t)m->n + 1));
			c->mapping_rows = m->n;
			c->mapping = (mapent **)calloc((size_t)m->n, sizeof(mapent *));
			c->mapping_len = (int32_t *)calloc((size_t)m->n, sizeof(int32_t));
			for (int64_t k = 0; k < m->nnz; ++k)
				c->mapping_len[m->inner[k]]++;
			for (int32_t r = 0; r < m->n; ++r)
			{
				c->mapping[r] = (mapent *)malloc(sizeof(mapent) * (size_t)(c->mapping_len[r] ? c->mapping_len[r] : 1));
				c->mapping_len[r] = 0;
			}
			int64_t index = 0;
			for (int32_t i = 0; i < m->n; ++i) /* i = outer (column) */
				for (int32_t ii = m->outer[i]; ii < m->outer[i + 1]; ++ii)
				{
					const int32_t j = m->inner[ii]; /* row */
					mapent *me = &c->mapping[j][c->mapping_len[j]++];
					me->col = i;
					me->index = index;
					++index;
				}
			/* second cache: value index of the k-th add_value call of element e */
			c->second_cache_n = c->sce_n;
			c->second_cache = (int32_t **)calloc((size_t)(c->sce_n ? c->sce_n : 1), sizeof(int32_t *));
			c->second_cache_len = (int64_t *)calloc((size_t)(c->sce_n ? c->sce_n : 1), sizeof(int64_t));
			for (int64_t e = 0; e < c->sce_n; ++e)
			{
				c->second_cache[e] = (int32_t *)malloc(sizeof(int32_t) * (size_t)(c->sce_len[e] ? c->sce_len[e] : 1));
				c->second_cache_len[e] = c->sce_len[e];
				for (int64_t k = 0; k < c->sce_len[e]; ++k)
				{
					const int32_t i = c->sce[e][2 * k], j = c->sce[e][2 * k + 1];
					int64_t idx = -1;
					for (int32_t q = 0; q < c->mapping_len[i]; ++q)
						if (c->mapping[i][q].col == j)
						{
							idx = c->mapping[i][q].index;
							break;
						}
					if (idx < 0)
						return -2;
					c->second_cache[e][k] = (int32_t)idx;
				}
				free(c->sce[e]);
				c->sce[e] = NULL;
			}
			c->sce_n = 0;
		}
		csc_copy(&c->mat, out);
	}
	else
	{
		out->n = (int32_t)c->size;
		out->nnz = c->n_values;
		out->outer = (int32_t *)malloc(sizeof(int32_t) * ((size_t)c->size + 1));
		out->inner = (int32_t *)malloc(sizeof(int32_t) * (size_t)(c->n_values ? c->n_values : 1));
		out->values = (double *)malloc(sizeof(double) * (size_t)(c->n_values ? c->n_values : 1));
		memcpy(out->outer, mc->outer_index, sizeof(int32_t) * ((size_t)c->size + 1));
		memcpy(out->inner, mc->inner_index, sizeof(int32_t) * (size_t)c->n_values);
		memcpy(out->values, c->values, sizeof(double) * (size_t)c->n_values);
		c->current_e = -1;
		c->current_e_index = -1;
	}
	for (int64_t k = 0; k < c->n_values; ++k)
		c->values[k] = 0;
	return 0;
}

/* NLAssembler::assemble_hessian with a caller-owned cache (Assembler.cpp:527-632), single thread */
int pfo_assemble_hessian_cached(const pfo_mesh *mesh, const pfo_material *mat, const double *x, pfo_cache *cache, pfo_csc *out)
{
	const int n_loc = pfo_n_loc(mesh->kind);
	const int size = block_size(mat);
	pfo_cache_set_zero(cache);
	pfo_cache *local = pfo_cache_clone(cache); /* LocalThreadMatStorage(buffer_size, mat_cache) */
	if (cache_main(cache)->mapping_rows == 0)
	{
		/* first time: the clone has no mapping to share, it acts as a fresh cache */
		local->main_cache = NULL;
	}
	pfo_quad quad;
	pfo_element_quadrature(mesh->kind, mesh->quad_order, &quad);
	pfo_vals *v = (pfo_vals *)malloc(sizeof(pfo_vals));
	double da[PFO_MAX_PTS], ld[3 * PFO_MAX_LOC];
	double *H = (double *)malloc(sizeof(double) * 81 * 81);
	const int N = n_loc * size;
	for (int64_t e = 0; e < mesh->n_el; ++e)
	{
		pfo_compute_vals(mesh, e, &quad, v);
		for (int k = 0; k < quad.n_pts; ++k)
			da[k] = v->det[k] * quad.w[k];
		gather_local_disp(mesh, e, n_loc, x, ld);
		nl_local_hessian_mat(mat_model(mat, e), v, da, ld, mat, e, H);
		const int32_t *idx = mesh->conn + e * n_loc;
		for (int i = 0; i < n_loc; ++i)
			for (int j = 0; j < n_loc; ++j)
				for (int n = 0; n < size; ++n)
					for (int m = 0; m < size; ++m)
						pfo_cache_add_value(local, (int)e, idx[i] * size + m, idx[j] * size + n, H[(i * size + m) * N + (j * size + n)] * 1.0 * 1.0);
	}
	pfo_cache_prune(local);
	pfo_cache_add(cache, local);
	local->main_cache = cache; /* so destroy does not free shared data */
	pfo_cache_destroy(local);
	free(v);
	free(H);
	return pfo_cache_get_matrix(cache, 1, out);
}

/* ------------------------------------------------------------------------------------------------ */
/* full_to_reduced_matrix (utils/MatrixUtils.cpp:408-469)                                           */
/* ------------------------------------------------------------------------------------------------ */

int pfo_full_to_reduced_matrix(const pfo_csc *full, const int32_t *removed_vars, int64_t n_removed, pfo_csc *reduced)
{
	const int32_t full_size = full->n;
	int32_t *indices = (int32_t *)malloc(sizeof(int32_t) * (size_t)(full_size ? full_size : 1));
	int32_t index = 0;
	int64_t kk = 0;
	for (int32_t i = 0; i < full_size; ++i)
	{
		if (kk < n_removed && removed_vars[kk] == i)
		{
			++kk;
			indices[i] = -1;
		}
		else
			indices[i] = index++;
	}
	tripvec tv = {0};
	for (int32_t k = 0; k < full_size; ++k)
	{
		if (indices[k] < 0)
			continue;
		for (int32_t it = full->outer[k]; it < full->outer[k + 1]; ++it)
		{
			const int32_t row = full->inner[it];
			if (indices[row] < 0)
				continue;
			tv_push(&tv, indices[row], indices[k], full->values[it]);
		}
	}
	const int rc = trips_to_csc(index, tv.t, tv.n, reduced);
	free(tv.t);
	free(indices);
	return rc;
}

/* ------------------------------------------------------------------------------------------------ */
/* CG (Eigen 3.4.0 IterativeLinearSolvers/ConjugateGradient.h, DiagonalPreconditioner) -- parity      */
/* unpinned: the Eigen/polysolve sources are not in the reference tree and it has no CG goldens.     */
/* ------------------------------------------------------------------------------------------------ */

void pfo_spmv(const pfo_csc *A, const double *x, double *y)
{
	for (int32_t r = 0; r < A->n; ++r)
		y[r] = 0;
	for (int32_t c = 0; c < A->n; ++c)
	{
		const double xc = x[c];
		for (int32_t k = A->outer[c]; k < A->outer[c + 1]; ++k)
			y[A->inner[k]] += A->values[k] * xc;
	}
}

static double dot(const double *a, const double *b, int64_t n)
{
	double s = 0;
	for (int64_t k = 0; k < n; ++k)
		s += a[k] * b[k];
	return s;
}

int pfo_cg(const pfo_csc *A, const double *b, double *x, int precond_jacobi, int max_iter, double tol, int *iters_out, double *error_out)
{
	const int64_t n = A->n;
	double *invdiag = (double *)malloc(sizeof(double) * (size_t)n);
	double *residual = (double *)malloc(sizeof(double) * (size_t)n);
	double *p = (double *)malloc(sizeof(double) * (size_t)n);
	double *z = (double *)malloc(sizeof(double) * (size_t)n);
	double *tmp = (double *)malloc(sizeof(double) * (size_t)n);
	/* DiagonalPreconditioner::factorize: invdiag = 1/diag if diag != 0 else 1 */
	for (int64_t k = 0; k < n; ++k)
		invdiag[k] = 1.0;
	if (precond_jacobi)
		for (int32_t c = 0; c < A->n; ++c)
			for (int32_t k = A->outer[c]; k < A->outer[c + 1]; ++k)
				if (A->inner[k] == c && A->values[k] != 0)
					invdiag[c] = 1.0 / A->values[k];

	int iters = 0;
	double tol_error = 0;
	pfo_spmv(A, x, tmp);
	for (int64_t k = 0; k < n; ++k)
		residual[k] = b[k] - tmp[k];
	const double rhsNorm2 = dot(b, b, n);
	if (rhsNorm2 == 0)
	{
		for (int64_t k = 0; k < n; ++k)
			x[k] = 0;
		goto done;
	}
	{
		const double considerAsZero = DBL_MIN;
		double threshold = tol * tol * rhsNorm2;
		if (threshold < considerAsZero)
			threshold = considerAsZero;
		double residualNorm2 = dot(residual, residual, n);
		if (residualNorm2 < threshold)
		{
			tol_error = sqrt(residualNorm2 / rhsNorm2);
			goto done;
		}
		for (int64_t k = 0; k < n; ++k)
			p[k] = invdiag[k] * residual[k];
		double absNew = dot(residual, p, n);
		int i = 0;
		while (i < max_iter)
		{
			pfo_spmv(A, p, tmp);
			const double alpha = absNew / dot(p, tmp, n);
			for (int64_t k = 0; k < n; ++k)
				x[k] += alpha * p[k];
			for (int64_t k = 0; k < n; ++k)
				residual[k] -= alpha * tmp[k];
			residualNorm2 = dot(residual, residual, n);
			if (residualNorm2 < threshold)
				break;
			for (int64_t k = 0; k < n; ++k)
				z[k] = invdiag[k] * residual[k];
			const double absOld = absNew;
			absNew = dot(residual, z, n);
			const double beta = absNew / absOld;
			for (int64_t k = 0; k < n; ++k)
				p[k] = z[k] + beta * p[k];
			i++;
		}
		tol_error = sqrt(residualNorm2 / rhsNorm2);
		iters = i;
	}
done:
	*iters_out = iters;
	*error_out = tol_error;
	free(invdiag), free(residual), free(p), free(z), free(tmp);
	return 0;
}

/* polysolve dirichlet_solve (un-vendored; restated from its documented behaviour, parity unpinned):
 * g = b restricted to the Dirichlet dofs; b -= A g on the free dofs; rows and columns of the Dirichlet
 * dofs are replaced by the identity; b keeps the Dirichlet values.  Call site:
 * state/StateSolveLinear.cpp:27-29. */
int pfo_dirichlet_modify(pfo_csc *A, double *b, const int32_t *boundary_nodes, int64_t n_boundary)
{
	const int32_t n = A->n;
	char *is_b = (char *)calloc((size_t)(n ? n : 1), 1);
	double *g = (double *)calloc((size_t)(n ? n : 1), sizeof(double));
	double *Ag = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
	for (int64_t k = 0; k < n_boundary; ++k)
	{
		is_b[boundary_nodes[k]] = 1;
		g[boundary_nodes[k]] = b[boundary_nodes[k]];
	}
	pfo_spmv(A, g, Ag);
	for (int32_t r = 0; r < n; ++r)
		if (!is_b[r])
			b[r] -= Ag[r];
	for (int32_t c = 0; c < n; ++c)
		for (int32_t k = A->outer[c]; k < A->outer[c + 1]; ++k)
		{
			const int32_t r = A->inner[k];
			if (is_b[r] || is_b[c])
				A->values[k] = (r == c) ? 1.0 : 0.0;
		}
	free(is_b), free(g), free(Ag);
	return 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* Newton solver                                                                                    */
/* ------------------------------------------------------------------------------------------------ */

void pfo_newton_default_params(pfo_newton_params *p)
{
	/* input-spec.json:1587-1647 */
	p->f_delta = 1e-10;
	p->grad_norm = 1e-8;
	p->first_grad_norm_tol = 1e-10;
	p->use_grad_norm_tol = 1e-4;
	p->max_iterations = 1000;
	p->use_grad_norm = 1;
	p->relative_gradient = 0;
	p->line_search = 1;
	p->cg_precond_jacobi = 1;
	p->cg_max_iter = 1000;
	p->cg_tol = 1e-12;
}

typedef struct
{
	const pfo_mesh *mesh;
	const pfo_material *mat;
	const double *rhs;
	const int32_t *bnd;
	int64_t n_bnd;
	int64_t n;       /* full size */
	uint8_t *fixed;  /* n */
	int n_threads;
	/* transient step: InertiaForm with this mass matrix and predictor, the other forms weighted (NULL / 1: static) */
	const pfo_csc *mass;
	const double *x_tilde;
	double weight;
} nl_problem;

/* 1/2 d^T M d and (optionally) g += M d for d = x - x_tilde (solver/forms/InertiaForm.cpp:14-26) */
static double inertia_terms(const nl_problem *p, const double *x, double *g_add)
{
	const int64_t n = p->n;
	double *d = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double)), *md = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
	for (int64_t k = 0; k < n; ++k)
		d[k] = x[k] - p->x_tilde[k];
	pfo_spmv(p->mass, d, md);
	const double prod = dot(d, md, n);
	if (g_add)
		for (int64_t k = 0; k < n; ++k)
			g_add[k] += md[k];
	free(d), free(md);
	return 0.5 * prod;
}

/* out = a * A + B, both n x n column-compressed with ascending inner indices (the sum of the weighted forms' Hessians,
 * solver/NLProblem.cpp hessian -> FullNLProblem: hessian += each form's second_derivative) */
static int csc_scaled_sum(double a, const pfo_csc *A, const pfo_csc *B, pfo_csc *out)
{
	const int32_t n = A->n;
	out->n = n;
	out->outer = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
	out->inner = (int32_t *)malloc(sizeof(int32_t) * (size_t)(A->nnz + B->nnz + 1));
	out->values = (double *)malloc(sizeof(double) * (size_t)(A->nnz + B->nnz + 1));
	if (!out->outer || !out->inner || !out->values)
		return -1;
	int64_t w = 0;
	for (int32_t c = 0; c < n; ++c)
	{
		out->outer[c] = (int32_t)w;
		int32_t ia = A->outer[c], ib = B->outer[c];
		const int32_t ea = A->outer[c + 1], eb = B->outer[c + 1];
		while (ia < ea || ib < eb)
		{
			const int32_t ra = ia < ea ? A->inner[ia] : INT32_MAX, rb = ib < eb ? B->inner[ib] : INT32_MAX;
			if (ra == rb)
				out->inner[w] = ra, out->values[w] = a * A->values[ia++] + B->values[ib++];
			else if (ra < rb)
				out->inner[w] = ra, out->values[w] = a * A->values[ia++];
			else
				out->inner[w] = rb, out->values[w] = B->values[ib++];
			++w;
		}
	}
	out->outer[n] = (int32_t)w;
	out->nnz = w;
	return 0;
}

/* NLProblem::value (solver/NLProblem.cpp:79-83): elastic energy + body form -x.rhs on the full vector */
static double nlp_value(const nl_problem *p, const double *x)
{
	double e = pfo_assemble_energy(p->mesh, p->mat, x, p->n_threads);
	if (p->rhs)
		e -= dot(x, p->rhs, p->n);
	if (p->mass)
		e = p->weight * e + inertia_terms(p, x, NULL);
	return e;
}

/* NLProblem::gradient (:85-90): full gradient, constrained entries removed (kept here as zeros) */
static void nlp_gradient(const nl_problem *p, const double *x, double *g)
{
	pfo_assemble_gradient(p->mesh, p->mat, x, p->n_threads, g);
	if (p->rhs)
		for (int64_t k = 0; k < p->n; ++k)
			g[k] -= p->rhs[k];
	if (p->mass)
	{
		for (int64_t k = 0; k < p->n; ++k)
			g[k] *= p->weight;
		(void)inertia_terms(p, x, g);
	}
	for (int64_t k = 0; k < p->n; ++k)
		if (p->fixed[k])
			g[k] = 0;
}

/* cppoptlib checkConvergence (meta.h, un-vendored): iteration limit, then f delta, then gradient norm */
static int check_convergence(const pfo_newton_params *stop, int iterations, double f_delta, double grad_norm, double grad_tol)
{
	if (stop->max_iterations > 0 && iterations > stop->max_iterations)
		return PFO_NEWTON_ITERATION_LIMIT;
	if (stop->f_delta > 0 && f_delta < stop->f_delta) /* false for nan */
		return PFO_NEWTON_F_DELTA;
	if (grad_tol > 0 && grad_norm < grad_tol)
		return PFO_NEWTON_GRAD_NORM;
	return PFO_NEWTON_CONTINUE;
}

/* BacktrackingLineSearch::line_search (line_search/BacktrackingLineSearch.hpp:29-147) without contact:
 * nan-free step (LineSearch.tpp:90-145), then the descent step (:157-255).  Returns nan on failure. */
static double backtracking(const nl_problem *p, const double *x, const double *dx, const pfo_newton_params *par, double *xn, double *g, int *evals)
{
	const int64_t n = p->n;
	const double min_step_size = 0;
	const int max_step_size_iter = 100;
	int cur_iter = 0;
	const double old_energy_in = nlp_value(p, x);
	++*evals;
	if (isnan(old_energy_in))
		return NAN;
	double step = 1;
	for (int64_t k = 0; k < n; ++k)
		xn[k] = x[k] + step * dx[k];
	while (step > min_step_size && cur_iter < max_step_size_iter)
	{
		const double energy = nlp_value(p, xn);
		++*evals;
		/* ElasticForm::is_step_valid (forms/ElasticForm.cpp:129-144): the gradient has no nan; for the materials here
		 * that is the case exactly when the energy is finite */
		if (!isfinite(energy))
		{
			step *= 0.5;
			for (int64_t k = 0; k < n; ++k)
				xn[k] = x[k] + step * dx[k];
		}
		else
			break;
		cur_iter++;
	}
	if (cur_iter >= max_step_size_iter || step <= min_step_size)
		return NAN;
	nlp_gradient(p, x, g);
	const double gn = sqrt(dot(g, g, n));
	const int use_grad_norm = gn < par->use_grad_norm_tol;
	const double old_energy = use_grad_norm ? dot(g, g, n) : old_energy_in;
	double cur_energy = NAN;
	while (step > min_step_size && cur_iter < max_step_size_iter)
	{
		for (int64_t k = 0; k < n; ++k)
			xn[k] = x[k] + step * dx[k];
		if (use_grad_norm)
		{
			nlp_gradient(p, xn, g);
			cur_energy = dot(g, g, n);
		}
		else
			cur_energy = nlp_value(p, xn);
		++*evals;
		if (!isfinite(cur_energy) || cur_energy > old_energy)
			step /= 2.0;
		else
			break;
		cur_iter++;
	}
	if (cur_iter >= max_step_size_iter || step <= min_step_size)
		return NAN;
	return step;
}

int pfo_newton_solve(const pfo_mesh *mesh, const pfo_material *mat, const double *rhs, const int32_t *boundary_nodes, int64_t n_boundary,
					 double *x, const pfo_newton_params *par, int n_threads, pfo_newton_info *info)
{
	return pfo_newton_solve_inertia(mesh, mat, rhs, boundary_nodes, n_boundary, x, par, n_threads, info, NULL, NULL, 1.0);
}

int pfo_newton_solve_inertia(const pfo_mesh *mesh, const pfo_material *mat, const double *rhs, const int32_t *boundary_nodes, int64_t n_boundary,
							 double *x, const pfo_newton_params *par, int n_threads, pfo_newton_info *info,
							 const pfo_csc *mass, const double *x_tilde, double weight)
{
	const int64_t n = mesh->n_nodes * block_size(mat);
	nl_problem P = {mesh, mat, rhs, boundary_nodes, n_boundary, n, NULL, n_threads < 1 ? 1 : n_threads, mass, x_tilde, weight};
	if (mass && (!x_tilde || mass->n != n))
		return -1;
	P.fixed = (uint8_t *)calloc((size_t)n, 1);
	for (int64_t k = 0; k < n_boundary; ++k)
		P.fixed[boundary_nodes[k]] = 1;
	const int64_t nr = n - n_boundary;
	double *grad = (double *)calloc((size_t)n, sizeof(double));
	double *dx = (double *)calloc((size_t)n, sizeof(double));
	double *xn = (double *)calloc((size_t)n, sizeof(double));
	double *gw = (double *)calloc((size_t)n, sizeof(double));
	double *br = (double *)calloc((size_t)(nr ? nr : 1), sizeof(double));
	double *xr = (double *)calloc((size_t)(nr ? nr : 1), sizeof(double));
	double *yr = (double *)calloc((size_t)(nr ? nr : 1), sizeof(double));
	memset(info, 0, sizeof(*info));
	int rc = 0;

	/* NonlinearSolver::minimize (solver/NonlinearSolver.tpp:33-275) */
	double old_energy = NAN;
	nlp_gradient(&P, x, grad);
	const double first_grad_norm = sqrt(dot(grad, grad, n));
	int status = PFO_NEWTON_CONTINUE, iterations = 0, strategy = 0;
	double reg_weight = 0; /* SparseNewtonDescentSolver.hpp:39-42,62 */
	const double reg_min = 1e-8, reg_max = 1e8, reg_inc = 10, reg_dec = 2;
	double grad_norm = first_grad_norm, energy = NAN;
	if (isnan(first_grad_norm))
	{
		status = PFO_NEWTON_FAILED;
		goto done;
	}
	if (check_convergence(par, 0, NAN, first_grad_norm / (par->relative_gradient ? first_grad_norm : 1), par->first_grad_norm_tol) != PFO_NEWTON_CONTINUE)
	{
		energy = nlp_value(&P, x);
		status = PFO_NEWTON_NOT_STARTED;
		goto done;
	}
	do
	{
		energy = nlp_value(&P, x);
		if (!isfinite(energy))
		{
			status = PFO_NEWTON_FAILED;
			break;
		}
		nlp_gradient(&P, x, grad);
		grad_norm = sqrt(dot(grad, grad, n));
		if (isnan(grad_norm))
		{
			status = PFO_NEWTON_FAILED;
			break;
		}
		/* SparseNewtonDescentSolver::compute_update_direction (SparseNewtonDescentSolver.tpp:63-100).
		 * Strategy 1 is the reference's "projected Newton": element Hessians projected to PSD plus the regularisation
		 * weight schedule (:40-47, :119-122). */
		int have_direction = 0;
		while (!have_direction)
		{
			if (strategy == 2)
			{
				for (int64_t k = 0; k < n; ++k)
					dx[k] = -grad[k];
				have_direction = 1;
				break;
			}
			pfo_csc H, Hr;
			/* assemble_hessian (:105-123): strategy 1 projects the element Hessians to PSD */
			if (strategy == 1 ? pfo_assemble_hessian_psd(mesh, mat, x, P.n_threads, &H) : pfo_assemble_hessian(mesh, mat, x, P.n_threads, &H))
			{
				rc = -1;
				goto cleanup;
			}
			if (mass)
			{
				pfo_csc Hs;
				if (csc_scaled_sum(weight, &H, mass, &Hs))
				{
					rc = -1;
					goto cleanup;
				}
				pfo_csc_free(&H);
				H = Hs;
			}
			pfo_full_to_reduced_matrix(&H, boundary_nodes, n_boundary, &Hr);
			pfo_csc_free(&H);
			if (reg_weight > 0)
				for (int32_t c = 0; c < Hr.n; ++c)
					for (int32_t q = Hr.outer[c]; q < Hr.outer[c + 1]; ++q)
						if (Hr.inner[q] == c)
							Hr.values[q] += reg_weight;
			int64_t w = 0;
			for (int64_t k = 0; k < n; ++k)
				if (!P.fixed[k])
				{
					br[w] = -grad[k];
					xr[w] = dx[k]; /* the previous direction is the initial guess (solveWithGuess) */
					++w;
				}
			int its = 0;
			double err = 0;
			pfo_cg(&Hr, br, xr, par->cg_precond_jacobi, par->cg_max_iter, par->cg_tol, &its, &err);
			info->cg_iterations += its;
			/* check_direction (:160-193) */
			pfo_spmv(&Hr, xr, yr);
			double res2 = 0, gd = 0;
			for (int64_t k = 0; k < nr; ++k)
			{
				const double r = yr[k] - br[k];
				res2 += r * r;
				gd += -br[k] * xr[k];
			}
			pfo_csc_free(&Hr);
			const double residual = sqrt(res2);
			const double lim = 1e-8 * grad_norm > 1e-5 ? 1e-8 * grad_norm : 1e-5;
			if (isnan(residual) || residual > lim || gd >= 0)
			{
				/* increase_descent_strategy (:40-47) */
				if (strategy == 0 || reg_weight > reg_max)
					strategy++;
				else
					reg_weight = reg_inc * reg_weight > reg_min ? reg_inc * reg_weight : reg_min;
				continue;
			}
			w = 0;
			for (int64_t k = 0; k < n; ++k)
				dx[k] = P.fixed[k] ? 0.0 : xr[w++];
			reg_weight /= reg_dec;
			if (reg_weight < reg_min)
				reg_weight = 0;
			have_direction = 1;
		}
		const double dg = dot(dx, grad, n);
		if (grad_norm != 0 && dg >= 0)
		{
			if (strategy == 0 || reg_weight > reg_max)
				strategy++;
			else
				reg_weight = reg_inc * reg_weight > reg_min ? reg_inc * reg_weight : reg_min;
			if (strategy > 2)
			{
				status = PFO_NEWTON_FAILED;
				break;
			}
			status = PFO_NEWTON_CONTINUE;
			continue;
		}
		double cur_grad_norm;
		if (!par->use_grad_norm)
		{
			cur_grad_norm = 0;
			for (int64_t k = 0; k < n; ++k)
				if (fabs(dx[k]) > cur_grad_norm)
					cur_grad_norm = fabs(dx[k]);
		}
		else
			cur_grad_norm = grad_norm / (par->relative_gradient ? first_grad_norm : 1);
		status = check_convergence(par, iterations, fabs(old_energy - energy), cur_grad_norm, par->grad_norm);
		old_energy = energy;
		double rate = 1;
		if (par->line_search)
		{
			rate = backtracking(&P, x, dx, par, xn, gw, &info->line_search_iterations);
			if (isnan(rate))
			{
				/* NonlinearSolver::line_search (:278-303) */
				if (strategy < 2)
				{
					if (strategy == 0 || reg_weight > reg_max)
						strategy++;
					else
						reg_weight = reg_inc * reg_weight > reg_min ? reg_inc * reg_weight : reg_min;
					status = PFO_NEWTON_CONTINUE;
					continue;
				}
				status = PFO_NEWTON_FAILED;
				break;
			}
		}
		for (int64_t k = 0; k < n; ++k)
			x[k] += rate * dx[k];
		info->descent_strategy = strategy;
		strategy = 0;
		++iterations;
	} while (status == PFO_NEWTON_CONTINUE);
done:
	info->status = status;
	info->iterations = iterations;
	info->energy = energy;
	info->grad_norm = grad_norm;
cleanup:
	free(P.fixed);
	free(grad);
	free(dx);
	free(xn);
	free(gw);
	free(br);
	free(xr);
	free(yr);
	return rc;
}

/* ------------------------------------------------------------------------------------------------ */
/* mass matrix and body-force right-hand side                                                       */
/* ------------------------------------------------------------------------------------------------ */

typedef struct
{
	const pfo_mesh *mesh;
	int size;
	double density;
	const double *density_e;
	const double *density_pts; /* one density per (element, mass quadrature point): Density::operator() is evaluated per point */
	int64_t e0, e1;
	tripvec tv;
} mass_job;

static void *mass_job_run(void *arg)
{
	mass_job *jb = (mass_job *)arg;
	const pfo_mesh *mesh = jb->mesh;
	const int n_loc = pfo_n_loc(mesh->kind), size = jb->size;
	pfo_quad quad;
	pfo_mass_quadrature(mesh->kind, &quad);
	pfo_vals *v = (pfo_vals *)malloc(sizeof(pfo_vals));
	double da[PFO_MAX_PTS];
	for (int64_t e = jb->e0; e < jb->e1; ++e)
	{
		/* Assembler.cpp:135-139: vals.compute at the mass quadrature points */
		pfo_compute_vals(mesh, e, &quad, v);
		for (int k = 0; k < quad.n_pts; ++k)
			da[k] = v->det[k] * quad.w[k];
		const double rho = jb->density_e ? jb->density_e[e] : jb->density;
		const int32_t *idx = mesh->conn + e * n_loc;
		for (int i = 0; i < n_loc; ++i)
			for (int j = 0; j <= i; ++j)
			{
				/* Mass.cpp:5-20 */
				double tmp = 0;
				for (int q = 0; q < quad.n_pts; ++q)
				{
					/* utils/ElasticityUtils.cpp:561-570: rho at the point */
					const double rho_q = jb->density_pts ? jb->density_pts[e * quad.n_pts + q] : rho;
					tmp += rho_q * v->val[i][q] * v->val[j][q] * da[q];
				}
				for (int n = 0; n < size; ++n)
					for (int m = 0; m < size; ++m)
					{
						const double local_value = (n == m) ? tmp : 0.0;
						if (fabs(local_value) < 1e-30)
							continue;
						const int32_t gi = idx[i] * size + m, gj = idx[j] * size + n;
						tv_push(&jb->tv, gi, gj, local_value * 1.0 * 1.0);
						if (j < i)
							tv_push(&jb->tv, gj, gi, local_value * 1.0 * 1.0);
					}
			}
	}
	free(v);
	return NULL;
}

static int assemble_mass_impl(const pfo_mesh *mesh, int size, double density, const double *density_e, const double *density_pts, int n_threads, pfo_csc *out);

int pfo_assemble_mass(const pfo_mesh *mesh, int size, double density, const double *density_e, int n_threads, pfo_csc *out)
{
	return assemble_mass_impl(mesh, size, density, density_e, NULL, n_threads, out);
}

/* the same with one density per (element, point of the mass quadrature rule), n_el x n_pts values */
int pfo_assemble_mass_pointwise(const pfo_mesh *mesh, int size, const double *density_pts, int n_threads, pfo_csc *out)
{
	return assemble_mass_impl(mesh, size, 0.0, NULL, density_pts, n_threads, out);
}

static int assemble_mass_impl(const pfo_mesh *mesh, int size, double density, const double *density_e, const double *density_pts, int n_threads, pfo_csc *out)
{
	int nt = n_threads < 1 ? 1 : n_threads;
	if (nt > mesh->n_el)
		nt = mesh->n_el > 0 ? (int)mesh->n_el : 1;
	mass_job *jobs = (mass_job *)calloc(nt, sizeof(mass_job));
	pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nt);
	const int64_t chunk = (mesh->n_el + nt - 1) / nt;
	for (int t = 0; t < nt; ++t)
	{
		jobs[t].mesh = mesh, jobs[t].size = size, jobs[t].density = density, jobs[t].density_e = density_e, jobs[t].density_pts = density_pts;
		jobs[t].e0 = t * chunk < mesh->n_el ? t * chunk : mesh->n_el;
		jobs[t].e1 = (t + 1) * chunk < mesh->n_el ? (t + 1) * chunk : mesh->n_el;
	}
	if (nt == 1)
		mass_job_run(&jobs[0]);
	else
	{
		for (int t = 0; t < nt; ++t)
			pthread_create(&th[t], NULL, mass_job_run, &jobs[t]);
		for (int t = 0; t < nt; ++t)
			pthread_join(th[t], NULL);
	}
	int64_t total = 0;
	for (int t = 0; t < nt; ++t)
		total += jobs[t].tv.n;
	trip *all = (trip *)malloc(sizeof(trip) * (size_t)(total ? total : 1));
	int64_t off = 0;
	for (int t = 0; t < nt; ++t)
	{
		if (jobs[t].tv.n)
			memcpy(all + off, jobs[t].tv.t, sizeof(trip) * (size_t)jobs[t].tv.n);
		off += jobs[t].tv.n;
		free(jobs[t].tv.t);
	}
	const int rc = trips_to_csc((int32_t)(mesh->n_nodes * size), all, total, out);
	free(all);
	free(jobs);
	free(th);
	return rc;
}

/* Physical positions of the quadrature points (ElementAssemblyValues::val, assembler/ElementAssemblyValues.cpp:328-345):
 * sum_j N_j^geo(xi_q) x_j; pts is n_el x n_pts x 3.  Returns n_pts (or -1). */
static int quadrature_points_impl(const pfo_mesh *mesh, int mass_rule, double *pts);

int pfo_quadrature_points(const pfo_mesh *mesh, double *pts)
{
	return quadrature_points_impl(mesh, 0, pts);
}

/* the same for the points of the mass quadrature rule */
int pfo_mass_quadrature_points(const pfo_mesh *mesh, double *pts)
{
	return quadrature_points_impl(mesh, 1, pts);
}

static int quadrature_points_impl(const pfo_mesh *mesh, int mass_rule, double *pts)
{
	const int n_g = pfo_n_loc(mesh->gkind), dim = pfo_kind_dim(mesh->kind);
	pfo_quad quad;
	if (mass_rule)
		pfo_mass_quadrature(mesh->kind, &quad);
	else if (pfo_element_quadrature(mesh->kind, mesh->quad_order, &quad))
		return -1;
	for (int64_t e = 0; e < mesh->n_el; ++e)
		for (int q = 0; q < quad.n_pts; ++q)
		{
			double p[3] = {0, 0, 0};
			for (int j = 0; j < n_g; ++j)
			{
				double nv;
				pfo_basis_value(mesh->gkind, j, quad.pts[q], &nv);
				const double *node = mesh->xyz + dim * (int64_t)mesh->gconn[e * n_g + j];
				for (int c = 0; c < dim; ++c)
					p[c] += nv * node[c];
			}
			for (int c = 0; c < dim; ++c)
				pts[((int64_t)e * quad.n_pts + q) * dim + c] = p[c];
		}
	return quad.n_pts;
}

/* RhsAssembler::assemble (assembler/RhsAssembler.cpp:55-98) with the force density given at every quadrature point
 * (what problem.rhs() returns there): force_pts is n_el x n_pts x size. */
int pfo_assemble_rhs_pointwise(const pfo_mesh *mesh, int size, const double *force_pts, double density, const double *density_e, double *rhs)
{
	const int n_loc = pfo_n_loc(mesh->kind);
	pfo_quad quad;
	if (pfo_element_quadrature(mesh->kind, mesh->quad_order, &quad))
		return -1;
	pfo_vals *v = (pfo_vals *)malloc(sizeof(pfo_vals));
	double rhs_fun[PFO_MAX_PTS][3];
	for (int64_t k = 0; k < mesh->n_nodes * size; ++k)
		rhs[k] = 0;
	for (int64_t e = 0; e < mesh->n_el; ++e)
	{
		pfo_compute_vals(mesh, e, &quad, v);
		const double rho = density_e ? density_e[e] : density;
		for (int d = 0; d < size; ++d)
			for (int q = 0; q < quad.n_pts; ++q)
				rhs_fun[q][d] = force_pts[((int64_t)e * quad.n_pts + q) * size + d] * (v->det[q] * quad.w[q] * rho);
		for (int i = 0; i < n_loc; ++i)
			for (int d = 0; d < size; ++d)
			{
				double rhs_value = 0;
				for (int q = 0; q < quad.n_pts; ++q)
					rhs_value += rhs_fun[q][d] * v->val[i][q];
				rhs[(int64_t)mesh->conn[e * n_loc + i] * size + d] += rhs_value * 1.0;
			}
	}
	free(v);
	return 0;
}

/* RhsAssembler::assemble with the density evaluated at every quadrature point (n_el x n_pts values, the stiffness rule),
 * force constant (f) or per point (force_pts); exactly one of f / force_pts is given */
int pfo_assemble_rhs_density_pointwise(const pfo_mesh *mesh, int size, const double *f, const double *force_pts, const double *density_pts, double *rhs)
{
	const int n_loc = pfo_n_loc(mesh->kind);
	pfo_quad quad;
	if (pfo_element_quadrature(mesh->kind, mesh->quad_order, &quad))
		return -1;
	pfo_vals *v = (pfo_vals *)malloc(sizeof(pfo_vals));
	double rhs_fun[PFO_MAX_PTS][3];
	for (int64_t k = 0; k < mesh->n_nodes * size; ++k)
		rhs[k] = 0;
	for (int64_t e = 0; e < mesh->n_el; ++e)
	{
		pfo_compute_vals(mesh, e, &quad, v);
		for (int d = 0; d < size; ++d)
			for (int q = 0; q < quad.n_pts; ++q)
			{
				const double fq = force_pts ? force_pts[((int64_t)e * quad.n_pts + q) * size + d] : f[d];
				rhs_fun[q][d] = fq * (v->det[q] * quad.w[q] * density_pts[e * quad.n_pts + q]);
			}
		for (int i = 0; i < n_loc; ++i)
			for (int d = 0; d < size; ++d)
			{
				double rhs_value = 0;
				for (int q = 0; q < quad.n_pts; ++q)
					rhs_value += rhs_fun[q][d] * v->val[i][q];
				rhs[(int64_t)mesh->conn[e * n_loc + i] * size + d] += rhs_value * 1.0;
			}
	}
	free(v);
	return 0;
}

int pfo_assemble_rhs(const pfo_mesh *mesh, int size, const double *f, double density, const double *density_e, double *rhs)
{
	const int n_loc = pfo_n_loc(mesh->kind);
	pfo_quad quad;
	if (pfo_element_quadrature(mesh->kind, mesh->quad_order, &quad))
		return -1;
	pfo_vals *v = (pfo_vals *)malloc(sizeof(pfo_vals));
	double rhs_fun[PFO_MAX_PTS][3];
	for (int64_t k = 0; k < mesh->n_nodes * size; ++k)
		rhs[k] = 0;
	for (int64_t e = 0; e < mesh->n_el; ++e)
	{
		pfo_compute_vals(mesh, e, &quad, v);
		const double rho = density_e ? density_e[e] : density;
		for (int d = 0; d < size; ++d)
			for (int q = 0; q < quad.n_pts; ++q)
				rhs_fun[q][d] = f[d] * (v->det[q] * quad.w[q] * rho);
		for (int i = 0; i < n_loc; ++i)
			for (int d = 0; d < size; ++d)
			{
				double rhs_value = 0; /* (rhs_fun.col(d).array() * v.val.array()).sum() */
				for (int q = 0; q < quad.n_pts; ++q)
					rhs_value += rhs_fun[q][d] * v->val[i][q];
				rhs[(int64_t)mesh->conn[e * n_loc + i] * size + d] += rhs_value * 1.0;
			}
	}
	free(v);
	return 0;
}
