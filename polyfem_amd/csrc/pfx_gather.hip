// pfx_gather.hip -- local-to-global accumulation of vectors and scalars without atomics, and the export of
// compressed (value-dependent / reduced) patterns.
//
// The reference gives every thread a full-length vector and merges them afterwards
// (src/polyfem/assembler/Assembler.cpp:442-501); here every global entry is OWNED by one lane, which sums the
// contributions listed for it by the pattern build (ascending element order) and writes the entry once.  The matrix
// counterpart is pfx_rowasm.h.
#include "pfx_internal.h"

#include <atomic>

namespace
{
	template <int NLOC, int BS>
	__global__ __launch_bounds__(256) void k_gather_vector(const int64_t *__restrict__ ne_ptr, const uint32_t *__restrict__ ne_src,
															const double *__restrict__ gloc, int64_t n_nodes, int drop_tiny, pfx_xcd_split split,
															double *__restrict__ out)
	{
		const int64_t n = n_nodes * BS;
		// one block per 256 rows; every XCD (blocks are dealt to the eight of them round robin, each has its own L2) takes a
		// contiguous range of the rows, so that the element rows of gloc shared by neighbouring nodes are fetched into one
		// L2; the ranges hold equal numbers of incidences
		const int xcd = blockIdx.x % PFX_XCDS;
		const int64_t vb = split.first[xcd] + (int64_t)(blockIdx.x / PFX_XCDS);
		const int64_t r = vb * blockDim.x + threadIdx.x;
		if (vb < split.first[xcd + 1] && r < n)
		{
			const int64_t g = r / BS;
			const int d = (int)(r - g * BS);
			double sum = 0.0;
			for (int64_t c = ne_ptr[g]; c < ne_ptr[g + 1]; ++c)
			{
				const double v = gloc[(int64_t)ne_src[c] * BS + d];
				if (drop_tiny && fabs(v) < 1e-30) // Assembler.cpp:476
					continue;
				sum += v;
			}
			out[r] = sum;
		}
	}

	// two-stage, fixed-order sum of the per-element energies
	__global__ __launch_bounds__(256) void k_reduce_stage1(const double *__restrict__ in, const uint8_t *__restrict__ owned, int64_t n, double *__restrict__ partial)
	{
		__shared__ double sh[256];
		double acc = 0.0;
		for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
			if (!owned || owned[k])
				acc += in[k];
		sh[threadIdx.x] = acc;
		__syncthreads();
		for (int s = 128; s > 0; s >>= 1)
		{
			if ((int)threadIdx.x < s)
				sh[threadIdx.x] += sh[threadIdx.x + s];
			__syncthreads();
		}
		if (threadIdx.x == 0)
			partial[blockIdx.x] = sh[0];
	}

	__global__ __launch_bounds__(256) void k_reduce_stage2(const double *__restrict__ partial, int n, double *__restrict__ out)
	{
		__shared__ double sh[256];
		double acc = 0.0;
		for (int k = threadIdx.x; k < n; k += 256)
			acc += partial[k];
		sh[threadIdx.x] = acc;
		__syncthreads();
		for (int s = 128; s > 0; s >>= 1)
		{
			if ((int)threadIdx.x < s)
				sh[threadIdx.x] += sh[threadIdx.x + s];
			__syncthreads();
		}
		if (threadIdx.x == 0)
			*out = sh[0];
	}

	// ---- export of value-dependent / reduced patterns ------------------------------------------------------

	// per scalar row: number of kept entries; keep = mask (if any) && row/col not constrained (if drop_fixed)
	__global__ void k_count_kept(const int64_t *__restrict__ node_ptr, const int32_t *__restrict__ node_col, int64_t n_nodes, int bs,
								 const uint8_t *__restrict__ mask, const uint8_t *__restrict__ fixed, int64_t *__restrict__ cnt)
	{
		const int64_t n_rows = n_nodes * bs;
		for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * blockDim.x)
		{
			if (fixed && fixed[r])
				continue; // constrained rows are compacted away: they get no slot in cnt (handled by the caller's map)
			const int64_t g = r / bs;
			const int m = (int)(r - g * bs);
			const int64_t b0 = node_ptr[g], deg = node_ptr[g + 1] - b0;
			int64_t c = 0;
			for (int64_t s = 0; s < deg; ++s)
				for (int n = 0; n < bs; ++n)
				{
					const int64_t col = (int64_t)node_col[b0 + s] * bs + n;
					if (fixed && fixed[col])
						continue;
					if (mask && !mask[(b0 + s) * bs * bs + m * bs + n])
						continue;
					++c;
				}
			cnt[r] = c;
		}
	}

	__global__ void k_write_kept(const int64_t *__restrict__ node_ptr, const int32_t *__restrict__ node_col, int64_t n_nodes, int bs,
								 const uint8_t *__restrict__ mask, const uint8_t *__restrict__ fixed, const int64_t *__restrict__ newrow,
								 const int64_t *__restrict__ off, const double *__restrict__ values,
								 int64_t *__restrict__ row_ptr, int32_t *__restrict__ col_idx, double *__restrict__ out_values)
	{
		const int64_t n_rows = n_nodes * bs;
		for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * blockDim.x)
		{
			if (fixed && fixed[r])
				continue;
			const int64_t g = r / bs;
			const int m = (int)(r - g * bs);
			const int64_t b0 = node_ptr[g], deg = node_ptr[g + 1] - b0;
			int64_t w = off[r];
			row_ptr[newrow ? newrow[r] : r] = w;
			for (int64_t s = 0; s < deg; ++s)
				for (int n = 0; n < bs; ++n)
				{
					const int64_t col = (int64_t)node_col[b0 + s] * bs + n;
					if (fixed && fixed[col])
						continue;
					const int64_t at = (b0 + s) * bs * bs + m * bs + n;
					if (mask && !mask[at])
						continue;
					col_idx[w] = (int32_t)(newrow ? newrow[col] : col);
					out_values[w] = values[at];
					++w;
				}
		}
	}

	// node-block storage -> scalar CSR order of the structural pattern (row g*bs+m = the bs*deg values of its slots)
	__global__ void k_values_to_csr(const int64_t *__restrict__ node_ptr, int64_t n_nodes, int bs, const double *__restrict__ blocks,
									double *__restrict__ csr)
	{
		const int lane = threadIdx.x & 63;
		const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
		const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
		const int b2 = bs * bs;
		for (int64_t g = wave; g < n_nodes; g += n_waves)
		{
			const int64_t b0 = node_ptr[g];
			const int deg = (int)(node_ptr[g + 1] - b0);
			const int rowlen = deg * bs;
			for (int t = lane; t < rowlen * bs; t += 64)
			{
				const int m = t / rowlen, r = t - m * rowlen;
				const int s = r / bs, n = r - s * bs;
				csr[b0 * b2 + t] = blocks[(b0 + s) * b2 + m * bs + n];
			}
		}
	}

	__global__ void k_free_flag(const uint8_t *__restrict__ fixed, int64_t n, int64_t *__restrict__ flag)
	{
		for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x)
			flag[r] = fixed[r] ? 0 : 1;
	}

	template <int NLOC, int BS>
	void launch_gv(pfx_ctx *ctx, double *d_g, bool drop_tiny)
	{
		hipLaunchKernelGGL((k_gather_vector<NLOC, BS>), dim3((unsigned)(PFX_XCDS * ctx->gv_split.widest)), dim3(256), 0, ctx->stream, ctx->d_ne_ptr, ctx->d_ne_src,
						   ctx->d_gloc, ctx->n_nodes, drop_tiny ? 1 : 0, ctx->gv_split, d_g);
	}
} // namespace

#define PFX_DISPATCH_NLOC_BS(FN, ...)                          \
	switch (ctx->n_loc * 10 + ctx->bs)                         \
	{                                                          \
	case 41:                                                   \
		FN<4, 1>(__VA_ARGS__);                                 \
		break;                                                 \
	case 43:                                                   \
		FN<4, 3>(__VA_ARGS__);                                 \
		break;                                                 \
	case 101:                                                  \
		FN<10, 1>(__VA_ARGS__);                                \
		break;                                                 \
	case 103:                                                  \
		FN<10, 3>(__VA_ARGS__);                                \
		break;                                                 \
	case 81:                                                   \
		FN<8, 1>(__VA_ARGS__);                                 \
		break;                                                 \
	case 83:                                                   \
		FN<8, 3>(__VA_ARGS__);                                 \
		break;                                                 \
	case 271:                                                  \
		FN<27, 1>(__VA_ARGS__);                                \
		break;                                                 \
	case 273:                                                  \
		FN<27, 3>(__VA_ARGS__);                                \
		break;                                                 \
	case 31:                                                   \
		FN<3, 1>(__VA_ARGS__);                                 \
		break;                                                 \
	case 32:                                                   \
		FN<3, 2>(__VA_ARGS__);                                 \
		break;                                                 \
	case 61:                                                   \
		FN<6, 1>(__VA_ARGS__);                                 \
		break;                                                 \
	case 62:                                                   \
		FN<6, 2>(__VA_ARGS__);                                 \
		break;                                                 \
	default:                                                   \
		PFX_FAIL(PFX_ERR_UNSUPPORTED, "unsupported n_loc/bs"); \
	}

pfx_status pfx_launch_gather_vector(pfx_ctx *ctx, double *d_g, bool drop_tiny)
{
	PFX_DISPATCH_NLOC_BS(launch_gv, ctx, d_g, drop_tiny);
	PFX_HIP(hipGetLastError());
	return PFX_OK;
}

pfx_status pfx_launch_reduce_energy(pfx_ctx *ctx, double *d_energy, const double *d_eloc, double *d_partials)
{
	const int nb = 256;
	if (!d_eloc)
		d_eloc = ctx->d_eloc;
	if (!d_partials)
		d_partials = ctx->d_partials;
	hipLaunchKernelGGL(k_reduce_stage1, dim3(nb), dim3(256), 0, ctx->stream, d_eloc, ctx->d_elem_owned, ctx->n_el, d_partials);
	hipLaunchKernelGGL(k_reduce_stage2, dim3(1), dim3(256), 0, ctx->stream, d_partials, nb, d_energy);
	PFX_HIP(hipGetLastError());
	return PFX_OK;
}

// ---- energy evaluation that allocates nothing ------------------------------------------------------------------
// Scratch of pfx_energy_dev_noalloc: the per-element energies (PFX_NOALLOC_MAX_EL doubles) followed by the 256 partial
// sums of the first reduction stage.  A file-scope device array: every GPU the library runs on holds its own copy
// (8 MiB) at its own address until the process ends.  Host code never uses the array's name as an address: the
// device address is looked up once per device (pfx_noalloc_init, called by pfx_create with that device current) and
// passed to the kernels as an ordinary pointer argument.
static __device__ __align__(256) double pfx_noalloc_scratch[PFX_NOALLOC_MAX_EL + 256];
// one slot per device, published with release / read with acquire: contexts may be created from several host threads
static std::atomic<double *> g_noalloc_ptr[PFX_MAX_DEVICES];

pfx_status pfx_noalloc_init(pfx_ctx *ctx)
{
	if (ctx->device < 0 || ctx->device >= PFX_MAX_DEVICES)
		PFX_FAIL(PFX_ERR_ARG, "device index beyond PFX_MAX_DEVICES");
	if (g_noalloc_ptr[ctx->device].load(std::memory_order_acquire))
		return PFX_OK;
	double *ptr = nullptr;
	PFX_HIP(hipSetDevice(ctx->device));
	PFX_HIP(hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(pfx_noalloc_scratch)));
	if (!ptr)
		PFX_FAIL(PFX_ERR_HIP, "hipGetSymbolAddress returned a null address for the no-allocation scratch");
	// two threads racing here look the same address up: either store is the right one
	g_noalloc_ptr[ctx->device].store(ptr, std::memory_order_release);
	return PFX_OK;
}

pfx_status pfx_energy_noalloc_run(pfx_ctx *ctx, const double *d_u, double *d_energy)
{
	double *ptr = (ctx->device >= 0 && ctx->device < PFX_MAX_DEVICES) ? g_noalloc_ptr[ctx->device].load(std::memory_order_acquire) : nullptr;
	if (!ptr)
		PFX_FAIL(PFX_ERR_HIP, "the no-allocation scratch of this device was not initialised");
	PFX_HIP(hipSetDevice(ctx->device)); // sets a thread-local index: no allocation
	if (ctx->n_el > PFX_NOALLOC_MAX_EL)
		PFX_FAIL(PFX_ERR_SIZE, "pfx_energy_dev_noalloc: more than PFX_NOALLOC_MAX_EL (1048576) elements; use pfx_energy_dev");
	// every slot read below is written first in this call (n_el element energies, then all 256 partials): nothing
	// has to be cleared although the array keeps the previous call's contents
	pfx_status rc = pfx_launch_local_nl(ctx, d_u, true, false, false, ptr);
	if (rc)
		return rc;
	return pfx_launch_reduce_energy(ctx, d_energy, ptr, ptr + PFX_NOALLOC_MAX_EL);
}

// copies the matrix values to the host in the order of the scalar CSR returned by pfx_export_pattern
pfx_status pfx_values_to_csr(pfx_ctx *ctx, double *host_values)
{
	pfx_tmp<double> d_csr;
	PFX_HIP(d_csr.alloc((size_t)ctx->nnz));
	hipLaunchKernelGGL(k_values_to_csr, dim3(grid_for(ctx->n_nodes * 64)), dim3(256), 0, ctx->stream, ctx->d_node_ptr, ctx->n_nodes, ctx->bs,
					   ctx->d_values, d_csr.p);
	PFX_HIP(hipMemcpyAsync(host_values, d_csr.p, sizeof(double) * (size_t)ctx->nnz, hipMemcpyDeviceToHost, ctx->stream));
	PFX_HIP(hipStreamSynchronize(ctx->stream));
	PFX_HIP(hipGetLastError());
	return PFX_OK;
}

#include <rocprim/rocprim.hpp>

// Exports the device matrix as a compressed scalar CSR after dropping (a) entries absent from the reference's
// value-dependent pattern (use_mask; Assembler.cpp:170-173) and/or (b) constrained rows and columns with
// renumbering (drop_fixed; utils/MatrixUtils.cpp:429-464).  row_ptr == nullptr: only sizes are returned.
pfx_status pfx_export_compressed(pfx_ctx *ctx, bool use_mask, bool drop_fixed, int64_t *n_rows_out, int64_t *nnz_out,
								 int64_t *row_ptr, int32_t *col_idx, double *values)
{
	const int64_t n_rows = ctx->n_dofs();
	hipStream_t st = ctx->stream;
	const uint8_t *mask = use_mask ? ctx->d_mask : nullptr;
	const uint8_t *fixed = (drop_fixed && ctx->n_dirichlet > 0) ? ctx->d_fixed : nullptr;
	pfx_tmp<int64_t> cnt, off, newrow, fl;
	pfx_tmp<char> tmp;
	size_t tb = 0;
	PFX_HIP(cnt.alloc((size_t)(n_rows + 1)));
	PFX_HIP(off.alloc((size_t)(n_rows + 1)));
	PFX_HIP(hipMemsetAsync(cnt.p, 0, sizeof(int64_t) * (size_t)(n_rows + 1), st));
	hipLaunchKernelGGL(k_count_kept, dim3(grid_for(n_rows)), dim3(256), 0, st, ctx->d_node_ptr, ctx->d_node_col, ctx->n_nodes, ctx->bs, mask, fixed, cnt.p);
	PFX_HIP(rocprim::exclusive_scan(nullptr, tb, cnt.p, off.p, (int64_t)0, (size_t)(n_rows + 1), rocprim::plus<int64_t>(), st));
	PFX_HIP(tmp.alloc(tb));
	PFX_HIP(rocprim::exclusive_scan(tmp.p, tb, cnt.p, off.p, (int64_t)0, (size_t)(n_rows + 1), rocprim::plus<int64_t>(), st));
	int64_t total = 0, n_red = n_rows;
	PFX_HIP(hipMemcpyAsync(&total, off.p + n_rows, sizeof(int64_t), hipMemcpyDeviceToHost, st));
	PFX_HIP(hipStreamSynchronize(st));
	if (fixed)
	{
		PFX_HIP(fl.alloc((size_t)(n_rows + 1)));
		PFX_HIP(newrow.alloc((size_t)(n_rows + 1)));
		PFX_HIP(hipMemsetAsync(fl.p, 0, sizeof(int64_t) * (size_t)(n_rows + 1), st));
		hipLaunchKernelGGL(k_free_flag, dim3(grid_for(n_rows)), dim3(256), 0, st, fixed, n_rows, fl.p);
		tb = 0;
		PFX_HIP(rocprim::exclusive_scan(nullptr, tb, fl.p, newrow.p, (int64_t)0, (size_t)(n_rows + 1), rocprim::plus<int64_t>(), st));
		PFX_HIP(tmp.alloc(tb));
		PFX_HIP(rocprim::exclusive_scan(tmp.p, tb, fl.p, newrow.p, (int64_t)0, (size_t)(n_rows + 1), rocprim::plus<int64_t>(), st));
		PFX_HIP(hipMemcpyAsync(&n_red, newrow.p + n_rows, sizeof(int64_t), hipMemcpyDeviceToHost, st));
		PFX_HIP(hipStreamSynchronize(st));
	}
	tmp.reset();
	*n_rows_out = n_red;
	*nnz_out = total;
	if (row_ptr)
	{
		pfx_tmp<int64_t> d_rp;
		pfx_tmp<int32_t> d_ci;
		pfx_tmp<double> d_v;
		PFX_HIP(d_rp.alloc((size_t)(n_red + 1)));
		PFX_HIP(d_ci.alloc((size_t)total));
		PFX_HIP(d_v.alloc((size_t)total));
		hipLaunchKernelGGL(k_write_kept, dim3(grid_for(n_rows)), dim3(256), 0, st, ctx->d_node_ptr, ctx->d_node_col, ctx->n_nodes, ctx->bs, mask, fixed,
						   (const int64_t *)newrow.p, (const int64_t *)off.p, ctx->d_values, d_rp.p, d_ci.p, d_v.p);
		PFX_HIP(hipMemcpyAsync(row_ptr, d_rp.p, sizeof(int64_t) * (size_t)n_red, hipMemcpyDeviceToHost, st));
		PFX_HIP(hipMemcpyAsync(col_idx, d_ci.p, sizeof(int32_t) * (size_t)total, hipMemcpyDeviceToHost, st));
		PFX_HIP(hipMemcpyAsync(values, d_v.p, sizeof(double) * (size_t)total, hipMemcpyDeviceToHost, st));
		PFX_HIP(hipStreamSynchronize(st));
		row_ptr[n_red] = total;
	}
	PFX_HIP(hipGetLastError());
	return PFX_OK;
}
