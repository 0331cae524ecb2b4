// pfx_dist.h -- types shared by the translation units of the partitioned solve (pfx_dist.hip, pfx_dist_amg.hip)
#pragma once

#include "pfx_internal.h"

struct pfx_comm;

// Halo lists of one level of a partition, by VERTEX (level 0: a mesh node; coarser AMG levels: an aggregate of the level above):
// to neighbour k go the values of the owned vertices d_send_idx[send_ptr[k] .. send_ptr[k+1]), from it come the halo
// vertices n_owned + recv_ptr[k] .. n_owned + recv_ptr[k+1] - 1.  A vector of `width` doubles per vertex is exchanged with
// one pack kernel and one grouped batch of transfers.
struct pfx_halo
{
	int64_t n_owned = 0, n_send = 0, n_recv = 0;
	int max_width = 0;
	std::vector<int32_t> peer;
	std::vector<int64_t> send_ptr, recv_ptr;
	int32_t *d_send_idx = nullptr;
	std::vector<int32_t> h_send_idx; // host copy (the set-up of a coarser level derives its lists from it)
	double *d_sendbuf = nullptr; // n_send * max_width
};

struct pfx_dist_amg; // hierarchy whose levels span the ranks (pfx_dist_amg.hip)

struct pfx_dist
{
	pfx_ctx *ctx = nullptr;
	pfx_comm *comm = nullptr;
	pfx_halo halo;             // level 0: mesh nodes
	double *d_scal = nullptr;  // 64 doubles: scalars on their way through an all-reduce
	int64_t pattern_gen = -1;  // pattern counter of the context at creation
	pfx_dist_amg *amg = nullptr;
	std::string err;
};

pfx_status pfx_comm_all_reduce(pfx_comm *comm, hipStream_t st, double *d_buf, int count);
int pfx_comm_rank(const pfx_comm *comm);
int pfx_comm_world(const pfx_comm *comm);
const std::string &pfx_comm_error(const pfx_comm *comm);
pfx_status pfx_halo_setup(pfx_comm *comm, pfx_halo *h, int64_t n_owned, int n_neigh, const int32_t *peer, const int64_t *send_ptr, const int32_t *send_idx,
						  const int64_t *recv_ptr, int max_width);
void pfx_halo_release(pfx_halo *h);
pfx_status pfx_halo_start(pfx_comm *comm, pfx_halo *h, hipStream_t st, double *d_vec, int width);
pfx_status pfx_halo_finish(pfx_comm *comm, pfx_halo *h, hipStream_t st);
void pfx_dist_amg_release(pfx_dist *dist);
pfx_status pfx_dist_pcg_amg_run(pfx_dist *dist, const double *d_b, double *d_x, int max_iter, double tol, int check_every, int *iters, double *rel_res);
pfx_status pfx_dist_amg_apply_run(pfx_dist *dist, const double *d_r, double *d_z);
pfx_status pfx_dist_cg_run(pfx_dist *dist, const double *d_b, double *d_x, int precond, int max_iter, double tol, int check_every, int *iters, double *rel_res);
pfx_status pfx_dist_newton_run(pfx_dist *dist, const double *d_rhs, double *d_x, const pfx_newton_params *par, pfx_newton_info *info);
pfx_status pfx_dist_implicit_euler_run(pfx_dist *dist, double dt, const double *d_rhs, double *d_x, double *d_v, double *d_a, const pfx_newton_params *params,
									   pfx_newton_info *info);
