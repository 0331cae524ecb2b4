// pfx_amg_graph.h -- device-side graph routines of the AMG set-up (pfx_amg_graph.hip): pairwise matching, quotient
// graphs with the lists the Galerkin products walk, merging of undersized aggregates, member lists.
#pragma once
#include "pfx_internal.h"

// vertex graph in device memory: CSR with ascending columns, self loops allowed; w = null: every edge has weight 1
struct pfx_dgraph
{
	int64_t n = 0, ne = 0;
	const int64_t *ptr = nullptr;
	const int32_t *col = nullptr;
	const uint32_t *w = nullptr;
};

struct pfx_dgraph_own
{
	int64_t n = 0, ne = 0;
	pfx_tmp<int64_t> ptr;
	pfx_tmp<int32_t> col;
	pfx_tmp<uint32_t> w;
	pfx_dgraph view() const
	{
		pfx_dgraph g;
		g.n = n, g.ne = ne, g.ptr = ptr.p, g.col = col.p, g.w = w.p;
		return g;
	}
	void take(pfx_dgraph_own &o)
	{
		ptr.reset(), col.reset(), w.reset();
		n = o.n, ne = o.ne;
		ptr.p = o.ptr.p, col.p = o.col.p, w.p = o.w.p;
		o.ptr.p = nullptr, o.col.p = nullptr, o.w.p = nullptr;
		o.n = o.ne = 0;
	}
};

// per coarse edge k (in the coarse graph's edge order): the fine edges behind it are pair_src[pair_ptr[k] .. pair_ptr[k+1])
// (fine row ascending, then fine edge ascending), pair_row their row vertices, arow[k] the coarse row of edge k
struct pfx_quot_lists
{
	pfx_tmp<int64_t> pair_ptr;
	pfx_tmp<uint32_t> pair_src;
	pfx_tmp<int32_t> pair_row, arow;
	int64_t n_pairs = 0;
};

// One round of pairwise matching: m[v] = cluster of vertex v in [0, *nc), or -1 for the vertices `excl` marks.  A free
// vertex proposes to the free neighbour with which it would form the smallest cluster (vsize: how many vertices of the
// level every vertex already stands for; null = 1), among those the nearest one (pos: vertex positions; null = no
// distances), then the one of the largest weight, then by a hash of the edge; mutual proposals become pairs; repeated
// until no free vertex has a free neighbour; the rest stay alone.  Size before weight: with the weight first, large
// clusters (many connections) pair among themselves round after round and leave single vertices behind, and the coarse
// graphs degenerate into stars.  Distance before weight: compact aggregates (1 M quadratic tetrahedra: 66 PCG
// iterations against 78 with the weight first).  Clusters are numbered by their smallest vertex; npos: their positions
// (mean of the members).
pfx_status pfx_graph_match(pfx_ctx *ctx, const pfx_dgraph &g, const uint8_t *excl, const double *pos, const int32_t *vsize, uint64_t seed, int32_t *m, int64_t *nc,
						   pfx_tmp<double> *npos = nullptr);

// size[c] = number of vertices with cur[v] == c
pfx_status pfx_graph_cluster_sizes(pfx_ctx *ctx, int64_t n_vert, const int32_t *cur, int64_t nc, int32_t *size);

// Quotient of g by `map` (vertex -> coarse vertex in [0, nc) or -1 = dropped): one coarse edge (a, b) per pair of coarse
// vertices joined by a fine edge, weight = sum of the fine weights, self loops kept, columns ascending.
pfx_status pfx_graph_quotient(pfx_ctx *ctx, const pfx_dgraph &g, const int32_t *map, int64_t nc, pfx_dgraph_own &out, pfx_quot_lists *lists);

// cur[v] = m[cur[v]] where cur[v] >= 0
pfx_status pfx_graph_compose(pfx_ctx *ctx, int64_t n, int32_t *cur, const int32_t *m);

// Clusters of fewer than min_size vertices join their most strongly connected neighbour cluster (gq: the quotient graph
// of the clusters); clusters are renumbered, *nc updated.
pfx_status pfx_graph_merge_small(pfx_ctx *ctx, int64_t n_vert, int32_t *cur, int64_t *nc, const pfx_dgraph &gq, int min_size);

// members of every cluster, ascending: mlist[mptr[a] .. mptr[a+1]); mptr has nc + 1 entries, mlist n_vert
pfx_status pfx_graph_members(pfx_ctx *ctx, int64_t n_vert, const int32_t *cur, int64_t nc, int64_t *mptr, int32_t *mlist);
