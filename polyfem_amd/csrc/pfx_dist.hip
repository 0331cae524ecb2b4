// pfx_dist.hip -- the partitioned solve inside the library: communicators (RCCL over xGMI; the caller's own transport for
// host buffers), halo exchange of node vectors, and the distributed CG loop.
//
// The reference is single-process (no MPI / NCCL anywhere in its tree).  One process per GPU holds one part of a node
// partition (pfx_set_owned): local numbering [owned | halo], owned rows complete after the local assembly.  A product
// y = A x needs the halo entries of x from their owners; a dot product needs one all-reduce.  The CG below is the single-GPU
// iteration of pfx_solve.hip (Eigen 3.4's operation order) with those two exchanges inserted:
//     pack + start halo exchange of p | interior rows of A p | wait | boundary rows, p.Ap | all-reduce 1 double
//     update x, r ; r.r, r.z | all-reduce 2 doubles | direction
// With RCCL the transfers run on a stream of their own and the interior rows hide them; nothing is read on the host except
// the convergence flag every `check_every` iterations.
#include "pfx_dist.h"

#include <rccl/rccl.h>

#include <cmath>

namespace
{
	// sendbuf[(j * width) + c] = vec[idx[j] * width + c]
	__global__ void k_halo_pack(const double *__restrict__ vec, const int32_t *__restrict__ idx, int64_t n, int width, double *__restrict__ buf)
	{
		const int64_t total = n * width;
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (int64_t)gridDim.x * blockDim.x)
		{
			const int64_t j = k / width;
			const int c = (int)(k - j * width);
			buf[k] = vec[(int64_t)idx[j] * width + c];
		}
	}
} // namespace

#define COMM_FAIL(code, msg) \
	do                       \
	{                        \
		comm->err = (msg);   \
		return (code);       \
	} while (0)

#define COMM_HIP(call)                                                       \
	do                                                                       \
	{                                                                        \
		hipError_t e_ = (call);                                              \
		if (e_ != hipSuccess)                                                \
		{                                                                    \
			comm->err = std::string(#call) + ": " + hipGetErrorString(e_);  \
			return PFX_ERR_HIP;                                              \
		}                                                                    \
	} while (0)

#define COMM_NCCL(call)                                                      \
	do                                                                       \
	{                                                                        \
		ncclResult_t r_ = (call);                                            \
		if (r_ != ncclSuccess)                                               \
		{                                                                    \
			comm->err = std::string(#call) + ": " + ncclGetErrorString(r_); \
			return PFX_ERR_HIP;                                              \
		}                                                                    \
	} while (0)

struct pfx_comm
{
	int kind = 0; // 0: RCCL, 1: host transport
	int rank = 0, world = 1, device = 0;
	std::string err;
	// RCCL
	ncclComm_t nccl = nullptr;
	hipStream_t cs = nullptr; // communication stream
	hipEvent_t ev_ready = nullptr, ev_done = nullptr;
	// host transport
	pfx_host_transport t{};
	char *h_stage = nullptr; // pinned staging memory
	size_t h_stage_bytes = 0;
	int64_t exchanges = 0, reductions = 0; // counters for tests and the benchmark
};

namespace
{
	pfx_status comm_stage(pfx_comm *comm, size_t bytes)
	{
		if (comm->h_stage_bytes >= bytes)
			return PFX_OK;
		if (comm->h_stage)
			(void)hipHostFree(comm->h_stage);
		comm->h_stage = nullptr;
		comm->h_stage_bytes = 0;
		COMM_HIP(hipHostMalloc((void **)&comm->h_stage, bytes, hipHostMallocDefault));
		comm->h_stage_bytes = bytes;
		return PFX_OK;
	}
} // namespace

pfx_status pfx_comm_all_reduce(pfx_comm *comm, hipStream_t st, double *d_buf, int count)
{
	comm->reductions++;
	if (comm->world <= 1 && comm->kind != 0)
		return PFX_OK;
	if (comm->kind == 0)
	{
		// (a one-rank RCCL communicator still takes the call: the sum over one rank, enqueued like any other)
		COMM_NCCL(ncclAllReduce(d_buf, d_buf, (size_t)count, ncclDouble, ncclSum, comm->nccl, st));
		return PFX_OK;
	}
	pfx_status rc = comm_stage(comm, sizeof(double) * (size_t)count);
	if (rc)
		return rc;
	double *h = (double *)comm->h_stage;
	COMM_HIP(hipMemcpyAsync(h, d_buf, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, st));
	COMM_HIP(hipStreamSynchronize(st));
	if (comm->t.all_reduce_sum(comm->t.user, h, count) != 0)
		COMM_FAIL(PFX_ERR_HIP, "the host transport's all_reduce_sum failed");
	COMM_HIP(hipMemcpyAsync(d_buf, h, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, st));
	COMM_HIP(hipStreamSynchronize(st)); // the staging buffer is reused by the next call
	return PFX_OK;
}

// Starts the exchange of `width` doubles per vertex: packs on `st`, then (RCCL) hands the transfers to the communication
// stream / (host) performs them.  d_vec: the vector [owned | halo] of that width.
pfx_status pfx_halo_start(pfx_comm *comm, pfx_halo *h, hipStream_t st, double *d_vec, int width)
{
	comm->exchanges++;
	if (comm->world <= 1 || h->peer.empty())
		return PFX_OK;
	if (width > h->max_width)
		COMM_FAIL(PFX_ERR_ARG, "halo exchange wider than the buffers of this level");
	if (h->n_send > 0)
		hipLaunchKernelGGL(k_halo_pack, dim3(grid_for(h->n_send * width)), dim3(256), 0, st, d_vec, h->d_send_idx, h->n_send, width, h->d_sendbuf);
	const int n = (int)h->peer.size();
	if (comm->kind == 0)
	{
		COMM_HIP(hipEventRecord(comm->ev_ready, st));
		COMM_HIP(hipStreamWaitEvent(comm->cs, comm->ev_ready, 0));
		COMM_NCCL(ncclGroupStart());
		for (int k = 0; k < n; ++k)
		{
			const int64_t ns = h->send_ptr[k + 1] - h->send_ptr[k], nr = h->recv_ptr[k + 1] - h->recv_ptr[k];
			if (ns > 0)
				COMM_NCCL(ncclSend(h->d_sendbuf + h->send_ptr[k] * width, (size_t)(ns * width), ncclDouble, h->peer[k], comm->nccl, comm->cs));
			if (nr > 0)
				COMM_NCCL(ncclRecv(d_vec + (h->n_owned + h->recv_ptr[k]) * width, (size_t)(nr * width), ncclDouble, h->peer[k], comm->nccl, comm->cs));
		}
		COMM_NCCL(ncclGroupEnd());
		COMM_HIP(hipEventRecord(comm->ev_done, comm->cs));
		return PFX_OK;
	}
	// host transport: device -> pinned -> callback -> pinned -> device (the halo part of the vector is one contiguous range)
	const size_t sb = sizeof(double) * (size_t)(h->n_send * width), rb = sizeof(double) * (size_t)(h->n_recv * width);
	pfx_status rc = comm_stage(comm, sb + rb + 16);
	if (rc)
		return rc;
	char *hs = comm->h_stage, *hr = comm->h_stage + sb;
	if (sb)
		COMM_HIP(hipMemcpyAsync(hs, h->d_sendbuf, sb, hipMemcpyDeviceToHost, st));
	COMM_HIP(hipStreamSynchronize(st));
	std::vector<const void *> sp((size_t)n);
	std::vector<void *> rp((size_t)n);
	std::vector<int64_t> sbytes((size_t)n), rbytes((size_t)n);
	for (int k = 0; k < n; ++k)
	{
		sp[(size_t)k] = hs + sizeof(double) * (size_t)(h->send_ptr[k] * width);
		rp[(size_t)k] = hr + sizeof(double) * (size_t)(h->recv_ptr[k] * width);
		sbytes[(size_t)k] = (int64_t)sizeof(double) * (h->send_ptr[k + 1] - h->send_ptr[k]) * width;
		rbytes[(size_t)k] = (int64_t)sizeof(double) * (h->recv_ptr[k + 1] - h->recv_ptr[k]) * width;
	}
	if (comm->t.exchange(comm->t.user, n, h->peer.data(), sp.data(), sbytes.data(), rp.data(), rbytes.data()) != 0)
		COMM_FAIL(PFX_ERR_HIP, "the host transport's exchange failed");
	if (rb)
		COMM_HIP(hipMemcpyAsync(d_vec + h->n_owned * width, hr, rb, hipMemcpyHostToDevice, st));
	COMM_HIP(hipStreamSynchronize(st)); // the staging buffer is reused by the next call
	return PFX_OK;
}

// work enqueued on `st` after this call sees the halo entries
pfx_status pfx_halo_finish(pfx_comm *comm, pfx_halo *h, hipStream_t st)
{
	if (comm->world <= 1 || h->peer.empty())
		return PFX_OK;
	if (comm->kind == 0)
		COMM_HIP(hipStreamWaitEvent(st, comm->ev_done, 0));
	return PFX_OK;
}

void pfx_halo_release(pfx_halo *h)
{
	if (h->d_send_idx)
		(void)hipFree(h->d_send_idx);
	if (h->d_sendbuf)
		(void)hipFree(h->d_sendbuf);
	h->d_send_idx = nullptr;
	h->d_sendbuf = nullptr;
	h->peer.clear();
	h->h_send_idx.clear();
	h->send_ptr.clear();
	h->recv_ptr.clear();
	h->n_send = h->n_recv = 0;
}

// lists -> device; max_width: the widest vector (doubles per vertex) this level exchanges
pfx_status pfx_halo_setup(pfx_comm *comm, pfx_halo *h, int64_t n_owned, int n_neigh, const int32_t *peer, const int64_t *send_ptr, const int32_t *send_idx,
						  const int64_t *recv_ptr, int max_width)
{
	pfx_halo_release(h);
	h->n_owned = n_owned;
	h->max_width = max_width;
	h->peer.assign(peer, peer + n_neigh);
	h->send_ptr.assign(send_ptr, send_ptr + n_neigh + 1);
	h->recv_ptr.assign(recv_ptr, recv_ptr + n_neigh + 1);
	h->n_send = n_neigh > 0 ? send_ptr[n_neigh] : 0;
	h->n_recv = n_neigh > 0 ? recv_ptr[n_neigh] : 0;
	if (h->n_send > 0)
		h->h_send_idx.assign(send_idx, send_idx + h->n_send);
	if (h->n_send > 0)
	{
		COMM_HIP(hipMalloc((void **)&h->d_send_idx, sizeof(int32_t) * (size_t)h->n_send));
		COMM_HIP(hipMalloc((void **)&h->d_sendbuf, sizeof(double) * (size_t)(h->n_send * max_width)));
		COMM_HIP(hipMemcpy(h->d_send_idx, send_idx, sizeof(int32_t) * (size_t)h->n_send, hipMemcpyHostToDevice));
	}
	return PFX_OK;
}

int pfx_comm_rank(const pfx_comm *comm) { return comm->rank; }
int pfx_comm_world(const pfx_comm *comm) { return comm->world; }
const std::string &pfx_comm_error(const pfx_comm *comm) { return comm->err; }

#define DIST_FAIL(code, msg) \
	do                       \
	{                        \
		dist->err = (msg);   \
		return (code);       \
	} while (0)

// a failed call of the context or of the communicator: their message becomes the object's
static pfx_status dist_rc(pfx_dist *dist, pfx_status rc, bool from_comm)
{
	if (rc)
		dist->err = from_comm ? dist->comm->err : dist->ctx->err;
	return rc;
}
#define DIST_CTX(call)                              \
	do                                              \
	{                                               \
		pfx_status rc_ = dist_rc(dist, (call), false); \
		if (rc_)                                    \
			return rc_;                             \
	} while (0)
#define DIST_COMM(call)                            \
	do                                             \
	{                                              \
		pfx_status rc_ = dist_rc(dist, (call), true); \
		if (rc_)                                   \
			return rc_;                            \
	} while (0)

static pfx_status dist_valid(pfx_dist *dist)
{
	pfx_ctx *ctx = dist->ctx;
	if (!ctx->d_node_ptr || ctx->pattern_gen != dist->pattern_gen)
		DIST_FAIL(PFX_ERR_ARG, "the context's mesh or pattern changed since pfx_dist_create");
	if (ctx->n_owned != dist->halo.n_owned)
		DIST_FAIL(PFX_ERR_ARG, "pfx_set_owned changed since pfx_dist_create");
	return PFX_OK;
}

pfx_status pfx_dist_cg_run(pfx_dist *dist, const double *d_b, double *d_x, int precond, int max_iter, double tol, int check_every, int *iters, double *rel_res)
{
	pfx_ctx *ctx = dist->ctx;
	pfx_comm *comm = dist->comm;
	hipStream_t st = ctx->stream;
	const int bs = ctx->bs;
	if (check_every <= 0)
		check_every = 50;
	if (ctx->d_work_ext)
		DIST_FAIL(PFX_ERR_ARG, "pfx_dist_cg_dev uses the context's own workspace (clear pfx_cgs_set_workspace first)");
	double *scal = dist->d_scal;
	DIST_CTX(pfx_cgs_prepare_impl(ctx, d_x, precond));
	DIST_COMM(pfx_halo_start(comm, &dist->halo, st, d_x, bs));
	DIST_COMM(pfx_halo_finish(comm, &dist->halo, st));
	DIST_CTX(pfx_cgs_residual_impl(ctx, d_b, d_x, scal));
	DIST_COMM(pfx_comm_all_reduce(comm, st, scal, 2));
	DIST_CTX(pfx_cgs_start_impl(ctx, d_x, tol, scal));
	DIST_COMM(pfx_comm_all_reduce(comm, st, scal + 2, 1));
	DIST_CTX(pfx_cgs_start2_impl(ctx, scal));
	double *p = nullptr;
	DIST_CTX(pfx_cg_direction_vector(ctx, &p));
	int it = 0, done = 0, its = 0;
	double rr = 0, bb = 0;
	while (it < max_iter && !done)
	{
		const int stop = (it + check_every < max_iter) ? it + check_every : max_iter;
		for (; it < stop; ++it)
		{
			// the rows that need no halo entry of p are multiplied while the exchange is in flight
			DIST_COMM(pfx_halo_start(comm, &dist->halo, st, p, bs));
			DIST_CTX(pfx_cgs_spmv_phase_impl(ctx, PFX_SPMV_INTERIOR, nullptr));
			DIST_COMM(pfx_halo_finish(comm, &dist->halo, st));
			DIST_CTX(pfx_cgs_spmv_phase_impl(ctx, PFX_SPMV_BOUNDARY, scal));
			DIST_COMM(pfx_comm_all_reduce(comm, st, scal, 1));
			DIST_CTX(pfx_cgs_update_impl(ctx, it, d_x, scal));
			DIST_COMM(pfx_comm_all_reduce(comm, st, scal + 1, 2));
			DIST_CTX(pfx_cgs_direction_impl(ctx, it, scal));
		}
		DIST_CTX(pfx_cgs_status_impl(ctx, &done, &its, &rr, &bb));
		// every rank reaches this point at the same iteration and searches for itself (no communication inside)
		if (!done && !ctx->sell_placed && it >= PFX_PLACE_AFTER && max_iter - it >= PFX_PLACE_WORTH)
			DIST_CTX(pfx_sell_place(ctx));
	}
	if (max_iter <= 0)
		DIST_CTX(pfx_cgs_status_impl(ctx, &done, &its, &rr, &bb));
	// the solution with a current halo, as a caller that goes on with x (line search, next assembly) needs it
	DIST_COMM(pfx_halo_start(comm, &dist->halo, st, d_x, bs));
	DIST_COMM(pfx_halo_finish(comm, &dist->halo, st));
	*iters = its;
	*rel_res = (bb == 0.0) ? 0.0 : std::sqrt(rr / bb);
	return PFX_OK;
}

extern "C"
{
	pfx_status pfx_comm_rccl_unique_id(void *id)
	{
		if (!id)
			return PFX_ERR_ARG;
		static_assert(sizeof(ncclUniqueId) == PFX_RCCL_ID_BYTES, "RCCL id size");
		ncclUniqueId u;
		if (ncclGetUniqueId(&u) != ncclSuccess)
			return PFX_ERR_HIP;
		memcpy(id, &u, sizeof(u));
		return PFX_OK;
	}

	pfx_status pfx_comm_create_rccl(pfx_comm **out, const void *id, int rank, int world, int device)
	{
		if (!out || !id || world < 1 || rank < 0 || rank >= world)
			return PFX_ERR_ARG;
		*out = nullptr;
		if (hipSetDevice(device) != hipSuccess)
			return PFX_ERR_HIP;
		pfx_comm *comm = new (std::nothrow) pfx_comm();
		if (!comm)
			return PFX_ERR_ARG;
		comm->kind = 0, comm->rank = rank, comm->world = world, comm->device = device;
		ncclUniqueId u;
		memcpy(&u, id, sizeof(u));
		if (ncclCommInitRank(&comm->nccl, world, u, rank) != ncclSuccess || hipStreamCreateWithFlags(&comm->cs, hipStreamNonBlocking) != hipSuccess
			|| hipEventCreateWithFlags(&comm->ev_ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&comm->ev_done, hipEventDisableTiming) != hipSuccess)
		{
			pfx_comm_destroy(comm);
			return PFX_ERR_HIP;
		}
		*out = comm;
		return PFX_OK;
	}

	pfx_status pfx_comm_create_host(pfx_comm **out, const pfx_host_transport *t, int rank, int world)
	{
		if (!out || !t || !t->exchange || !t->all_reduce_sum || world < 1 || rank < 0 || rank >= world)
			return PFX_ERR_ARG;
		pfx_comm *comm = new (std::nothrow) pfx_comm();
		if (!comm)
			return PFX_ERR_ARG;
		comm->kind = 1, comm->rank = rank, comm->world = world;
		comm->t = *t;
		*out = comm;
		return PFX_OK;
	}

	void pfx_comm_destroy(pfx_comm *comm)
	{
		if (!comm)
			return;
		if (comm->nccl)
			(void)ncclCommDestroy(comm->nccl);
		if (comm->cs)
			(void)hipStreamDestroy(comm->cs);
		if (comm->ev_ready)
			(void)hipEventDestroy(comm->ev_ready);
		if (comm->ev_done)
			(void)hipEventDestroy(comm->ev_done);
		if (comm->h_stage)
			(void)hipHostFree(comm->h_stage);
		delete comm;
	}

	const char *pfx_comm_last_error(const pfx_comm *comm) { return comm ? comm->err.c_str() : "null communicator"; }

	pfx_status pfx_dist_create(pfx_dist **out, pfx_ctx *ctx, pfx_comm *comm, int n_neigh, const int32_t *peer, const int64_t *send_ptr, const int32_t *send_nodes,
							   const int64_t *recv_ptr)
	{
		if (!out || !ctx || !comm || n_neigh < 0 || (n_neigh > 0 && (!peer || !send_ptr || !recv_ptr)))
			return PFX_ERR_ARG;
		*out = nullptr;
		if (!ctx->d_node_ptr)
			PFX_FAIL(PFX_ERR_ARG, "pfx_dist_create: pfx_build_pattern must be called first");
		const int64_t n_halo = ctx->n_nodes - ctx->n_owned;
		static const int64_t zero[1] = {0};
		if (n_neigh == 0)
			send_ptr = recv_ptr = zero;
		// the pack kernel and the transfers index device memory with these lists: validate them here
		if (send_ptr[0] != 0 || recv_ptr[0] != 0)
			PFX_FAIL(PFX_ERR_ARG, "pfx_dist_create: send_ptr[0] and recv_ptr[0] must be 0");
		for (int k = 0; k < n_neigh; ++k)
		{
			if (peer[k] < 0 || peer[k] >= comm->world || peer[k] == comm->rank)
				PFX_FAIL(PFX_ERR_ARG, "pfx_dist_create: neighbour rank out of range");
			if (send_ptr[k + 1] < send_ptr[k] || recv_ptr[k + 1] < recv_ptr[k])
				PFX_FAIL(PFX_ERR_ARG, "pfx_dist_create: send_ptr / recv_ptr must not decrease");
		}
		if (recv_ptr[n_neigh] != n_halo)
			PFX_FAIL(PFX_ERR_ARG, "pfx_dist_create: the receive ranges must cover the halo nodes exactly");
		if (send_ptr[n_neigh] > 0 && !send_nodes)
			return PFX_ERR_ARG;
		for (int64_t j = 0; j < send_ptr[n_neigh]; ++j)
			if (send_nodes[j] < 0 || send_nodes[j] >= ctx->n_owned)
				PFX_FAIL(PFX_ERR_ARG, "pfx_dist_create: send_nodes must be owned nodes");
		PFX_HIP(hipSetDevice(ctx->device));
		pfx_dist *dist = new (std::nothrow) pfx_dist();
		if (!dist)
			return PFX_ERR_ARG;
		dist->ctx = ctx;
		dist->comm = comm;
		dist->pattern_gen = ctx->pattern_gen;
		if (pfx_halo_setup(comm, &dist->halo, ctx->n_owned, n_neigh, peer, send_ptr, send_nodes, recv_ptr, 8) != PFX_OK
			|| hipMalloc((void **)&dist->d_scal, sizeof(double) * 64) != hipSuccess || hipMemset(dist->d_scal, 0, sizeof(double) * 64) != hipSuccess)
		{
			ctx->err = comm->err.empty() ? "pfx_dist_create: device allocation failed" : comm->err;
			pfx_dist_destroy(dist);
			return PFX_ERR_HIP;
		}
		*out = dist;
		return PFX_OK;
	}

	void pfx_dist_destroy(pfx_dist *dist)
	{
		if (!dist)
			return;
		pfx_dist_amg_release(dist);
		pfx_halo_release(&dist->halo);
		if (dist->d_scal)
			(void)hipFree(dist->d_scal);
		delete dist;
	}

	const char *pfx_dist_last_error(const pfx_dist *dist) { return dist ? dist->err.c_str() : "null object"; }

	pfx_status pfx_dist_halo_exchange_dev(pfx_dist *dist, double *d_vec)
	{
		if (!dist || !d_vec)
			return PFX_ERR_ARG;
		if (dist_valid(dist))
			return PFX_ERR_ARG;
		DIST_COMM(pfx_halo_start(dist->comm, &dist->halo, dist->ctx->stream, d_vec, dist->ctx->bs));
		DIST_COMM(pfx_halo_finish(dist->comm, &dist->halo, dist->ctx->stream));
		return PFX_OK;
	}

	pfx_status pfx_dist_all_reduce_dev(pfx_dist *dist, double *d_buf, int count)
	{
		if (!dist || !d_buf || count < 1 || count > 64)
			return PFX_ERR_ARG;
		DIST_COMM(pfx_comm_all_reduce(dist->comm, dist->ctx->stream, d_buf, count));
		return PFX_OK;
	}

	pfx_status pfx_dist_energy_dev(pfx_dist *dist, const double *d_u, double *d_energy)
	{
		if (!dist || !d_u || !d_energy)
			return PFX_ERR_ARG;
		if (dist_valid(dist))
			return PFX_ERR_ARG;
		DIST_CTX(pfx_energy_dev(dist->ctx, d_u, d_energy));
		DIST_COMM(pfx_comm_all_reduce(dist->comm, dist->ctx->stream, d_energy, 1));
		return PFX_OK;
	}

	pfx_status pfx_dist_cg_dev(pfx_dist *dist, const double *d_b, double *d_x, int precond, int max_iter, double tol, int check_every, int *iters, double *rel_res)
	{
		if (!dist || !d_b || !d_x || !iters || !rel_res)
			return PFX_ERR_ARG;
		if (dist_valid(dist))
			return PFX_ERR_ARG;
		if (hipSetDevice(dist->ctx->device) != hipSuccess)
			DIST_FAIL(PFX_ERR_HIP, "hipSetDevice failed");
		if (precond == PFX_PRECOND_AMG)
			return pfx_dist_pcg_amg_run(dist, d_b, d_x, max_iter, tol, check_every, iters, rel_res);
		if (precond != PFX_PRECOND_IDENTITY && precond != PFX_PRECOND_JACOBI)
			DIST_FAIL(PFX_ERR_ARG, "unknown preconditioner");
		return pfx_dist_cg_run(dist, d_b, d_x, precond, max_iter, tol, check_every, iters, rel_res);
	}

	pfx_status pfx_dist_newton_dev(pfx_dist *dist, const double *d_rhs, double *d_x, const pfx_newton_params *params, pfx_newton_info *info)
	{
		if (!dist || !d_x || !params || !info)
			return PFX_ERR_ARG;
		if (dist_valid(dist))
			return PFX_ERR_ARG;
		pfx_ctx *ctx = dist->ctx;
		if (ctx->kind < 0)
			DIST_FAIL(PFX_ERR_ARG, "the context holds no mesh");
		if (ctx->model == PFX_LAPLACIAN || ctx->bs < 2)
			DIST_FAIL(PFX_ERR_UNSUPPORTED, "pfx_dist_newton_dev needs an elastic model");
		if (ctx->d_work_ext)
			DIST_FAIL(PFX_ERR_ARG, "pfx_dist_newton_dev uses the context's own workspace (clear pfx_cgs_set_workspace first)");
		if (params->cg_precond != PFX_PRECOND_IDENTITY && params->cg_precond != PFX_PRECOND_JACOBI && params->cg_precond != PFX_PRECOND_AMG)
			DIST_FAIL(PFX_ERR_ARG, "unknown preconditioner");
		if (hipSetDevice(ctx->device) != hipSuccess)
			DIST_FAIL(PFX_ERR_HIP, "hipSetDevice failed");
		return pfx_dist_newton_run(dist, d_rhs, d_x, params, info);
	}

	pfx_status pfx_dist_implicit_euler_step_dev(pfx_dist *dist, double dt, const double *d_rhs, double *d_x, double *d_v, double *d_a,
												const pfx_newton_params *params, pfx_newton_info *info)
	{
		if (!dist || !d_x || !d_v || !params || !info)
			return PFX_ERR_ARG;
		if (dist_valid(dist))
			return PFX_ERR_ARG;
		pfx_ctx *ctx = dist->ctx;
		if (!(dt > 0.0))
			DIST_FAIL(PFX_ERR_ARG, "pfx_dist_implicit_euler_step_dev: dt must be positive");
		if (ctx->kind < 0 || ctx->model == PFX_LAPLACIAN || ctx->bs < 2)
			DIST_FAIL(PFX_ERR_ARG, "pfx_dist_implicit_euler_step_dev needs a mesh and an elastic model");
		if (ctx->mass_count == 0)
			DIST_FAIL(PFX_ERR_ARG, "pfx_inertia_set_mass must be called first");
		if (ctx->d_work_ext)
			DIST_FAIL(PFX_ERR_ARG, "pfx_dist_implicit_euler_step_dev uses the context's own workspace (clear pfx_cgs_set_workspace first)");
		if (hipSetDevice(ctx->device) != hipSuccess)
			DIST_FAIL(PFX_ERR_HIP, "hipSetDevice failed");
		return pfx_dist_implicit_euler_run(dist, dt, d_rhs, d_x, d_v, d_a, params, info);
	}

	pfx_status pfx_dist_amg_apply_dev(pfx_dist *dist, const double *d_r, double *d_z)
	{
		if (!dist || !d_r || !d_z)
			return PFX_ERR_ARG;
		if (dist_valid(dist))
			return PFX_ERR_ARG;
		if (hipSetDevice(dist->ctx->device) != hipSuccess)
			DIST_FAIL(PFX_ERR_HIP, "hipSetDevice failed");
		return pfx_dist_amg_apply_run(dist, d_r, d_z);
	}

	pfx_status pfx_dist_halo_bytes(pfx_dist *dist, int64_t *bytes)
	{
		if (!dist || !bytes)
			return PFX_ERR_ARG;
		*bytes = (int64_t)sizeof(double) * dist->ctx->bs * (dist->halo.n_send + dist->halo.n_recv);
		return PFX_OK;
	}
}

