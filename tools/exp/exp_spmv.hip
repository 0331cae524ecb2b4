// Experiment harness: SpMV kernel variants on a dumped node-block pattern (bs = 3). Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <cmath>
#include <string>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ double wave_sum(double v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64); return v; }

// V0/V1: pure streaming reads
__global__ __launch_bounds__(256) void k_read64(const double* __restrict__ v, int64_t n, double* out) {
	double a = 0; for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) a += v[i];
	a = wave_sum(a); if ((threadIdx.x & 63) == 0) atomicAdd(out, a);
}
__global__ __launch_bounds__(256) void k_read128(const double2* __restrict__ v, int64_t n2, double* out) {
	double a = 0; for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) { double2 t = v[i]; a += t.x + t.y; }
	a = wave_sum(a); if ((threadIdx.x & 63) == 0) atomicAdd(out, a);
}
// contiguous ranges per block (like the spmv) with b128
__global__ __launch_bounds__(256) void k_read128_ranges(const double2* __restrict__ v, int64_t n2, double* out) {
	const int nblk = gridDim.x; const int per = (nblk + 7) / 8; const int vb = (blockIdx.x % 8) * per + blockIdx.x / 8;
	const int64_t npb = (n2 + nblk - 1) / nblk; int64_t b = vb * npb, e = b + npb; if (e > n2) e = n2;
	double a = 0; for (int64_t i = b + threadIdx.x; i < e; i += 256) { double2 t = v[i]; a += t.x + t.y; }
	a = wave_sum(a); if ((threadIdx.x & 63) == 0) atomicAdd(out, a);
}

template <int LPN, int MODE> // MODE 0 full, 1 no x gather (x const), 2 no col/x
__global__ __launch_bounds__(256) void k_group(const int64_t* __restrict__ node_ptr, const int32_t* __restrict__ node_col, const double* __restrict__ values,
	int64_t n_nodes, const double* __restrict__ x, double* __restrict__ y) {
	constexpr int BS = 3; constexpr int GPB = 256 / LPN;
	const int nblk = gridDim.x; const int per = (nblk + 7) / 8; const int vb = (blockIdx.x % 8) * per + blockIdx.x / 8;
	const int64_t npb = (n_nodes + nblk - 1) / nblk; const int64_t g0 = (int64_t)vb * npb; int64_t g1 = g0 + npb; if (g1 > n_nodes) g1 = n_nodes;
	const int sl = threadIdx.x % LPN, grp = threadIdx.x / LPN;
	for (int64_t g = g0 + grp; g < g1; g += GPB) {
		const int64_t b0 = node_ptr[g]; const int deg = (int)(node_ptr[g + 1] - b0); const int rowlen = deg * BS; const double* v = values + b0 * 9;
		double acc[3] = {0, 0, 0};
		for (int k = sl; k < rowlen; k += LPN) {
			const int s = k / 3, n = k - 3 * s;
			double xv;
			if (MODE == 0) xv = x[(int64_t)node_col[b0 + s] * 3 + n];
			else if (MODE == 1) xv = (double)node_col[b0 + s];
			else xv = 1.0 + k;
			for (int m = 0; m < 3; ++m) acc[m] += v[m * rowlen + k] * xv;
		}
		for (int m = 0; m < 3; ++m) for (int o = LPN / 2; o > 0; o >>= 1) acc[m] += __shfl_xor(acc[m], o, LPN);
		if (sl < 3) y[g * 3 + sl] = sl == 0 ? acc[0] : (sl == 1 ? acc[1] : acc[2]);
	}
}

// group kernel, x staged per node in LDS (one gather per block), rows streamed
template <int LPN>
__global__ __launch_bounds__(256) void k_group_lds(const int64_t* __restrict__ node_ptr, const int32_t* __restrict__ node_col, const double* __restrict__ values,
	int64_t n_nodes, const double* __restrict__ x, double* __restrict__ y) {
	constexpr int BS = 3; constexpr int GPB = 256 / LPN; constexpr int XMAX = 3 * 128;
	__shared__ double xs[GPB][XMAX];
	const int nblk = gridDim.x; const int per = (nblk + 7) / 8; const int vb = (blockIdx.x % 8) * per + blockIdx.x / 8;
	const int64_t npb = (n_nodes + nblk - 1) / nblk; const int64_t g0 = (int64_t)vb * npb; int64_t g1 = g0 + npb; if (g1 > n_nodes) g1 = n_nodes;
	const int sl = threadIdx.x % LPN, grp = threadIdx.x / LPN;
	for (int64_t g = g0 + grp; g < g1; g += GPB) {
		const int64_t b0 = node_ptr[g]; const int deg = (int)(node_ptr[g + 1] - b0); const int rowlen = deg * BS; const double* v = values + b0 * 9;
		for (int s = sl; s < deg; s += LPN) { const int64_t c = (int64_t)node_col[b0 + s] * 3; xs[grp][3 * s] = x[c]; xs[grp][3 * s + 1] = x[c + 1]; xs[grp][3 * s + 2] = x[c + 2]; }
		double acc[3] = {0, 0, 0};
		// (LPN <= 64: the group lives in one wave, LDS writes are visible to the wave's later reads)
		for (int k = sl; k < rowlen; k += LPN) { const double xv = xs[grp][k]; for (int m = 0; m < 3; ++m) acc[m] += v[m * rowlen + k] * xv; }
		for (int m = 0; m < 3; ++m) for (int o = LPN / 2; o > 0; o >>= 1) acc[m] += __shfl_xor(acc[m], o, LPN);
		if (sl < 3) y[g * 3 + sl] = sl == 0 ? acc[0] : (sl == 1 ? acc[1] : acc[2]);
	}
}


// block-SELL-64: one lane per node (3 rows), slices of 64 nodes sorted by degree inside windows, values [slice][slot][c][lane]
template <int UNROLL, int MODE>
__global__ __launch_bounds__(256) void k_sell(const int64_t* __restrict__ slice_vbase, const int64_t* __restrict__ slice_cbase, const int32_t* __restrict__ slice_len,
	const int32_t* __restrict__ perm, const int32_t* __restrict__ cols, const double* __restrict__ values, int64_t n_slices, const double* __restrict__ x, double* __restrict__ y) {
	const int lane = threadIdx.x & 63;
	const int nblk = gridDim.x; const int per = (nblk + 7) / 8; const int vb = (blockIdx.x % 8) * per + blockIdx.x / 8;
	const int64_t spb = (n_slices + nblk - 1) / nblk; // slices per block
	int64_t s0 = (int64_t)vb * spb, s1 = s0 + spb; if (s1 > n_slices) s1 = n_slices;
	for (int64_t s = s0 + (threadIdx.x >> 6); s < s1; s += 4) {
		const int len = slice_len[s];
		const double* v = values + slice_vbase[s] + lane;
		const int32_t* c = cols + slice_cbase[s] + lane;
		double a0 = 0, a1 = 0, a2 = 0;
#pragma unroll UNROLL
		for (int j = 0; j < len; ++j) {
			double x0, x1, x2;
			if (MODE == 0) { const int64_t cc = (int64_t)c[j * 64] * 3; x0 = x[cc]; x1 = x[cc + 1]; x2 = x[cc + 2]; }
			else if (MODE == 1) { const double t = (double)c[j * 64]; x0 = t; x1 = t + 1; x2 = t + 2; }
			else if (MODE == 3) { const int64_t cc = (int64_t)c[j * 64] * 4; const double2 a = *reinterpret_cast<const double2*>(x + cc); const double2 b = *reinterpret_cast<const double2*>(x + cc + 2); x0 = a.x; x1 = a.y; x2 = b.x; }
			else { x0 = j; x1 = j + 1; x2 = j + 2; }
			const double* vj = v + (int64_t)j * 9 * 64;
			a0 += vj[0] * x0 + vj[64] * x1 + vj[128] * x2;
			a1 += vj[192] * x0 + vj[256] * x1 + vj[320] * x2;
			a2 += vj[384] * x0 + vj[448] * x1 + vj[512] * x2;
		}
		const int g = perm[s * 64 + lane];
		if (g >= 0) { y[3 * (int64_t)g] = a0; y[3 * (int64_t)g + 1] = a1; y[3 * (int64_t)g + 2] = a2; }
	}
}


// SELL, one slice per wave, slices dealt to blocks so that each XCD walks a contiguous part of the matrix
template <int UNROLL, int MODE>
__global__ __launch_bounds__(256) void k_sell1(const int64_t* __restrict__ slice_vbase, const int64_t* __restrict__ slice_cbase, const int32_t* __restrict__ slice_len,
	const int32_t* __restrict__ perm, const int32_t* __restrict__ cols, const double* __restrict__ values, int64_t n_slices, const double* __restrict__ x, double* __restrict__ y) {
	const int lane = threadIdx.x & 63;
	const int nblk = gridDim.x; const int per = (nblk + 7) / 8; const int vb = (blockIdx.x % 8) * per + blockIdx.x / 8;
	const int64_t s = (int64_t)vb * 4 + (threadIdx.x >> 6);
	if (vb >= nblk || s >= n_slices) return;
	const int len = slice_len[s];
	const double* v = values + slice_vbase[s] + lane;
	const int32_t* c = cols + slice_cbase[s] + lane;
	double a0 = 0, a1 = 0, a2 = 0;
#pragma unroll UNROLL
	for (int j = 0; j < len; ++j) {
		double x0, x1, x2;
		if (MODE == 0) { const int64_t cc = (int64_t)c[j * 64] * 3; x0 = x[cc]; x1 = x[cc + 1]; x2 = x[cc + 2]; }
		else { x0 = j; x1 = j + 1; x2 = j + 2; }
		const double* vj = v + (int64_t)j * 9 * 64;
		a0 += vj[0] * x0 + vj[64] * x1 + vj[128] * x2;
		a1 += vj[192] * x0 + vj[256] * x1 + vj[320] * x2;
		a2 += vj[384] * x0 + vj[448] * x1 + vj[512] * x2;
	}
	const int g = perm[s * 64 + lane];
	if (g >= 0) { y[3 * (int64_t)g] = a0; y[3 * (int64_t)g + 1] = a1; y[3 * (int64_t)g + 2] = a2; }
}


// SELL with 16-byte value loads: per slot, entries (0,1),(2,3),(4,5),(6,7) as double2 per lane, entry 8 single
template <int UNROLL>
__global__ __launch_bounds__(256) void k_sell2(const int64_t* __restrict__ slice_vbase, const int64_t* __restrict__ slice_cbase, const int32_t* __restrict__ slice_len,
	const int32_t* __restrict__ perm, const int32_t* __restrict__ cols, const double* __restrict__ values, int64_t n_slices, const double* __restrict__ x, double* __restrict__ y) {
	const int lane = threadIdx.x & 63;
	const int nblk = gridDim.x; const int per = (nblk + 7) / 8; const int vb = (blockIdx.x % 8) * per + blockIdx.x / 8;
	const int64_t s = (int64_t)vb * 4 + (threadIdx.x >> 6);
	if (vb >= nblk || s >= n_slices) return;
	const int len = slice_len[s];
	const double* v = values + slice_vbase[s];
	const int32_t* c = cols + slice_cbase[s] + lane;
	double a0 = 0, a1 = 0, a2 = 0;
#pragma unroll UNROLL
	for (int j = 0; j < len; ++j) {
		const int64_t cc = (int64_t)c[j * 64] * 3; const double x0 = x[cc], x1 = x[cc + 1], x2 = x[cc + 2];
		const double* vj = v + (int64_t)j * 9 * 64;
		const double2 p0 = reinterpret_cast<const double2*>(vj)[lane];
		const double2 p1 = reinterpret_cast<const double2*>(vj + 128)[lane];
		const double2 p2 = reinterpret_cast<const double2*>(vj + 256)[lane];
		const double2 p3 = reinterpret_cast<const double2*>(vj + 384)[lane];
		const double p4 = vj[512 + lane];
		a0 += p0.x * x0 + p0.y * x1 + p1.x * x2;
		a1 += p1.y * x0 + p2.x * x1 + p2.y * x2;
		a2 += p3.x * x0 + p3.y * x1 + p4 * x2;
	}
	const int g = perm[s * 64 + lane];
	if (g >= 0) { y[3 * (int64_t)g] = a0; y[3 * (int64_t)g + 1] = a1; y[3 * (int64_t)g + 2] = a2; }
}

// k_sell2 with non-temporal value/column loads (NT) and optionally x padded to 4 doubles per node (PADX)
template <int UNROLL, bool NT, bool PADX, int WPB = 4>
__global__ __launch_bounds__(64 * WPB) void k_sell3(const int64_t* __restrict__ slice_vbase, const int64_t* __restrict__ slice_cbase, const int32_t* __restrict__ slice_len,
	const int32_t* __restrict__ perm, const int32_t* __restrict__ cols, const double* __restrict__ values, int64_t n_slices, const double* __restrict__ x, double* __restrict__ y) {
	const int lane = threadIdx.x & 63;
	const int nblk = gridDim.x; const int per = (nblk + 7) / 8; const int vb = (blockIdx.x % 8) * per + blockIdx.x / 8;
	const int64_t s = (int64_t)vb * WPB + (threadIdx.x >> 6);
	if (vb >= nblk || s >= n_slices) return;
	const int len = slice_len[s];
	const double* v = values + slice_vbase[s];
	const int32_t* c = cols + slice_cbase[s] + lane;
	typedef double d2 __attribute__((ext_vector_type(2)));
	double a0 = 0, a1 = 0, a2 = 0;
#pragma unroll UNROLL
	for (int j = 0; j < len; ++j) {
		const int32_t ci = NT ? __builtin_nontemporal_load(c + j * 64) : c[j * 64];
		double x0, x1, x2;
		if (PADX) { const d2* xp = reinterpret_cast<const d2*>(x + (int64_t)ci * 4); const d2 q = xp[0]; x0 = q.x; x1 = q.y; x2 = x[(int64_t)ci * 4 + 2]; }
		else { const int64_t cc = (int64_t)ci * 3; x0 = x[cc]; x1 = x[cc + 1]; x2 = x[cc + 2]; }
		const double* vj = v + (int64_t)j * 9 * 64;
		const d2* vp = reinterpret_cast<const d2*>(vj) + lane;
		d2 p0, p1, p2, p3; double p4;
		if (NT) { p0 = __builtin_nontemporal_load(vp); p1 = __builtin_nontemporal_load(vp + 64); p2 = __builtin_nontemporal_load(vp + 128); p3 = __builtin_nontemporal_load(vp + 192); p4 = __builtin_nontemporal_load(vj + 512 + lane); }
		else { p0 = vp[0]; p1 = vp[64]; p2 = vp[128]; p3 = vp[192]; p4 = vj[512 + lane]; }
		a0 += p0.x * x0 + p0.y * x1 + p1.x * x2;
		a1 += p1.y * x0 + p2.x * x1 + p2.y * x2;
		a2 += p3.x * x0 + p3.y * x1 + p4 * x2;
	}
	const int g = perm[s * 64 + lane];
	if (g >= 0) { y[3 * (int64_t)g] = a0; y[3 * (int64_t)g + 1] = a1; y[3 * (int64_t)g + 2] = a2; }
}

// group kernel with the first three k-iterations issued together (rowlen <= 96 needs no further loop)
__global__ __launch_bounds__(256) void k_group32_u3(const int64_t* __restrict__ node_ptr, const int32_t* __restrict__ node_col, const double* __restrict__ values,
	int64_t n_nodes, const double* __restrict__ x, double* __restrict__ y) {
	constexpr int LPN = 32, GPB = 8;
	const int nblk = gridDim.x; const int per = (nblk + 7) / 8; const int vb = (blockIdx.x % 8) * per + blockIdx.x / 8;
	const int64_t npb = (n_nodes + nblk - 1) / nblk; const int64_t g0 = (int64_t)vb * npb; int64_t g1 = g0 + npb; if (g1 > n_nodes) g1 = n_nodes;
	const int sl = threadIdx.x % LPN, grp = threadIdx.x / LPN;
	int64_t g = g0 + grp;
	int64_t b0 = 0, b1 = 0;
	if (g < g1) { b0 = node_ptr[g]; b1 = node_ptr[g + 1]; }
	while (g < g1) {
		const int64_t gn = g + GPB;
		int64_t nb0 = 0, nb1 = 0;
		if (gn < g1) { nb0 = node_ptr[gn]; nb1 = node_ptr[gn + 1]; }
		const int deg = (int)(b1 - b0); const int rowlen = deg * 3; const double* v = values + b0 * 9;
		double acc[3] = {0, 0, 0};
		for (int kb = 0; kb < rowlen; kb += 96) {
			double vv[3][3]; double xv[3]; int kk[3];
#pragma unroll
			for (int u = 0; u < 3; ++u) { kk[u] = kb + u * 32 + sl; const bool ok = kk[u] < rowlen; const int k = ok ? kk[u] : 0;
				for (int m = 0; m < 3; ++m) vv[u][m] = ok ? v[m * rowlen + k] : 0.0;
				const int sidx = k / 3, n = k - 3 * sidx; xv[u] = x[(int64_t)node_col[b0 + sidx] * 3 + n]; }
#pragma unroll
			for (int u = 0; u < 3; ++u) for (int m = 0; m < 3; ++m) acc[m] += vv[u][m] * xv[u];
		}
		for (int m = 0; m < 3; ++m) for (int o = LPN / 2; o > 0; o >>= 1) acc[m] += __shfl_xor(acc[m], o, LPN);
		if (sl < 3) y[g * 3 + sl] = sl == 0 ? acc[0] : (sl == 1 ? acc[1] : acc[2]);
		g = gn; b0 = nb0; b1 = nb1;
	}
}

int main(int argc, char** argv) {
	const char* path = argc > 1 ? argv[1] : "/tmp/pat.bin";
	FILE* f = fopen(path, "rb"); if (!f) { printf("no file\n"); return 1; }
	int64_t hdr[2]; fread(hdr, 8, 2, f); const int64_t n_nodes = hdr[0], n_blocks = hdr[1];
	std::vector<int64_t> np(n_nodes + 1); std::vector<int32_t> nc(n_blocks);
	fread(np.data(), 8, n_nodes + 1, f); fread(nc.data(), 4, n_blocks, f); fclose(f);
	const int64_t nnz = n_blocks * 9, n = n_nodes * 3;
	printf("nodes %ld blocks %ld nnz %ld\n", (long)n_nodes, (long)n_blocks, (long)nnz);
	int64_t* d_np; int32_t* d_nc; double *d_v, *d_x, *d_y, *d_out;
	CK(hipMalloc(&d_np, 8 * (n_nodes + 1))); CK(hipMalloc(&d_nc, 4 * n_blocks)); CK(hipMalloc(&d_v, 8 * (nnz + 2))); CK(hipMalloc(&d_x, 8 * n)); CK(hipMalloc(&d_y, 8 * n)); CK(hipMalloc(&d_out, 8));
	CK(hipMemcpy(d_np, np.data(), 8 * (n_nodes + 1), hipMemcpyHostToDevice)); CK(hipMemcpy(d_nc, nc.data(), 4 * n_blocks, hipMemcpyHostToDevice));
	{ std::vector<double> hv(nnz + 2); for (int64_t i = 0; i < nnz + 2; ++i) hv[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5; CK(hipMemcpy(d_v, hv.data(), 8 * (nnz + 2), hipMemcpyHostToDevice));
	  std::vector<double> hx(n); for (int64_t i = 0; i < n; ++i) hx[i] = (double)((i * 40503u) % 997) / 997.0; CK(hipMemcpy(d_x, hx.data(), 8 * n, hipMemcpyHostToDevice)); }
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	const double bytes = 8.0 * nnz + 4.0 * n_blocks + 8.0 * (n_nodes + 1) + 16.0 * n;
	auto timeit = [&](const char* name, auto fn, double by) {
		fn(); CK(hipDeviceSynchronize()); float best = 1e9, tot = 0; const int R = 10;
		for (int r = 0; r < R; ++r) { CK(hipEventRecord(e0)); fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms; if (ms < best) best = ms; }
		printf("%-28s avg %.4f ms  best %.4f ms  %.0f GB/s (avg)\n", name, tot / R, best, by / (tot / R) / 1e6); CK(hipGetLastError());
	};
	const int G = argc > 2 ? atoi(argv[2]) : 2048;
	timeit("read64 grid-stride", [&] { k_read64<<<G, 256>>>(d_v, nnz, d_out); }, 8.0 * nnz);
	timeit("read128 grid-stride", [&] { k_read128<<<G, 256>>>((const double2*)d_v, nnz / 2, d_out); }, 8.0 * nnz);
	timeit("read128 grid-stride x4 grid", [&] { k_read128<<<4 * G, 256>>>((const double2*)d_v, nnz / 2, d_out); }, 8.0 * nnz);
	timeit("read128 ranges", [&] { k_read128_ranges<<<G, 256>>>((const double2*)d_v, nnz / 2, d_out); }, 8.0 * nnz);
	timeit("group32 full", [&] { k_group<32, 0><<<G, 256>>>(d_np, d_nc, d_v, n_nodes, d_x, d_y); }, bytes);
	timeit("group32 col only", [&] { k_group<32, 1><<<G, 256>>>(d_np, d_nc, d_v, n_nodes, d_x, d_y); }, bytes);
	timeit("group32 no col/x", [&] { k_group<32, 2><<<G, 256>>>(d_np, d_nc, d_v, n_nodes, d_x, d_y); }, bytes);
	timeit("group16 full", [&] { k_group<16, 0><<<G, 256>>>(d_np, d_nc, d_v, n_nodes, d_x, d_y); }, bytes);
	timeit("group32 full grid x2", [&] { k_group<32, 0><<<2 * G, 256>>>(d_np, d_nc, d_v, n_nodes, d_x, d_y); }, bytes);
	timeit("group32 full grid x4", [&] { k_group<32, 0><<<4 * G, 256>>>(d_np, d_nc, d_v, n_nodes, d_x, d_y); }, bytes);
	timeit("group32 lds-x", [&] { k_group_lds<32><<<G, 256>>>(d_np, d_nc, d_v, n_nodes, d_x, d_y); }, bytes);
	timeit("group64 lds-x", [&] { k_group_lds<64><<<G, 256>>>(d_np, d_nc, d_v, n_nodes, d_x, d_y); }, bytes);
	timeit("group16 lds-x", [&] { k_group_lds<16><<<G, 256>>>(d_np, d_nc, d_v, n_nodes, d_x, d_y); }, bytes);

	// ---- block-SELL-64 build (host) ----
	const int SIG = argc > 3 ? atoi(argv[3]) : 2048;
	std::vector<int32_t> perm; perm.reserve(n_nodes + 64);
	{ std::vector<int32_t> w; for (int64_t b = 0; b < n_nodes; b += SIG) { int64_t e = b + SIG < n_nodes ? b + SIG : n_nodes; w.clear(); for (int64_t g = b; g < e; ++g) w.push_back((int32_t)g);
		std::stable_sort(w.begin(), w.end(), [&](int32_t a, int32_t c) { return (np[a + 1] - np[a]) > (np[c + 1] - np[c]); }); for (auto g : w) perm.push_back(g); } }
	while (perm.size() % 64) perm.push_back(-1);
	const int64_t n_slices = perm.size() / 64;
	std::vector<int64_t> svb(n_slices + 1), scb(n_slices + 1); std::vector<int32_t> slen(n_slices);
	int64_t vtot = 0, ctot = 0;
	for (int64_t sidx = 0; sidx < n_slices; ++sidx) { int mx = 0; for (int l = 0; l < 64; ++l) { int g = perm[sidx * 64 + l]; if (g >= 0) { int d = (int)(np[g + 1] - np[g]); if (d > mx) mx = d; } }
		slen[sidx] = mx; svb[sidx] = vtot; scb[sidx] = ctot; vtot += (int64_t)mx * 9 * 64; ctot += (int64_t)mx * 64; }
	printf("SELL: slices %ld padded values %ld (%.3fx nnz)\n", (long)n_slices, (long)vtot, (double)vtot / nnz);
	std::vector<int32_t> scol(ctot);
	for (int64_t sidx = 0; sidx < n_slices; ++sidx) for (int l = 0; l < 64; ++l) { int g = perm[sidx * 64 + l]; for (int j = 0; j < slen[sidx]; ++j) { int32_t c = 0; if (g >= 0) { int d = (int)(np[g + 1] - np[g]); c = j < d ? nc[np[g] + j] : g; } scol[scb[sidx] + (int64_t)j * 64 + l] = c; } }
	int64_t *d_svb, *d_scb; int32_t *d_slen, *d_perm, *d_scol; double* d_sv;
	CK(hipMalloc(&d_svb, 8 * (n_slices + 1))); CK(hipMalloc(&d_scb, 8 * (n_slices + 1))); CK(hipMalloc(&d_slen, 4 * n_slices)); CK(hipMalloc(&d_perm, 4 * perm.size())); CK(hipMalloc(&d_scol, 4 * ctot)); CK(hipMalloc(&d_sv, 8 * vtot));
	CK(hipMemcpy(d_svb, svb.data(), 8 * n_slices, hipMemcpyHostToDevice)); CK(hipMemcpy(d_scb, scb.data(), 8 * n_slices, hipMemcpyHostToDevice)); CK(hipMemcpy(d_slen, slen.data(), 4 * n_slices, hipMemcpyHostToDevice));
	CK(hipMemcpy(d_perm, perm.data(), 4 * perm.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(d_scol, scol.data(), 4 * ctot, hipMemcpyHostToDevice));
	{ std::vector<double> hv(vtot); for (int64_t i = 0; i < vtot; ++i) hv[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5; CK(hipMemcpy(d_sv, hv.data(), 8 * vtot, hipMemcpyHostToDevice)); }
	timeit("group32 unroll3+prefetch", [&] { k_group32_u3<<<G, 256>>>(d_np, d_nc, d_v, n_nodes, d_x, d_y); }, bytes);
	timeit("sell64 unroll1", [&] { k_sell<1, 0><<<G, 256>>>(d_svb, d_scb, d_slen, d_perm, d_scol, d_sv, n_slices, d_x, d_y); }, bytes);
	timeit("sell64 unroll2", [&] { k_sell<2, 0><<<G, 256>>>(d_svb, d_scb, d_slen, d_perm, d_scol, d_sv, n_slices, d_x, d_y); }, bytes);
	timeit("sell64 unroll4", [&] { k_sell<4, 0><<<G, 256>>>(d_svb, d_scb, d_slen, d_perm, d_scol, d_sv, n_slices, d_x, d_y); }, bytes);
	timeit("sell64 u4 col only", [&] { k_sell<4, 1><<<G, 256>>>(d_svb, d_scb, d_slen, d_perm, d_scol, d_sv, n_slices, d_x, d_y); }, bytes);
	timeit("sell64 u4 no col/x", [&] { k_sell<4, 2><<<G, 256>>>(d_svb, d_scb, d_slen, d_perm, d_scol, d_sv, n_slices, d_x, d_y); }, bytes);
	{ double* d_x4; CK(hipMalloc(&d_x4, 8 * 4 * (n_nodes + 1))); CK(hipMemset(d_x4, 0, 8 * 4 * (n_nodes + 1)));
	  timeit("sell64 u4 padded x (4/node)", [&] { k_sell<4, 3><<<G, 256>>>(d_svb, d_scb, d_slen, d_perm, d_scol, d_sv, n_slices, d_x4, d_y); }, bytes);
	  timeit("sell64 u2 padded x (4/node)", [&] { k_sell<2, 3><<<G, 256>>>(d_svb, d_scb, d_slen, d_perm, d_scol, d_sv, n_slices, d_x4, d_y); }, bytes);
	  timeit("sell64 u4 padded x grid x2", [&] { k_sell<4, 3><<<2 * G, 256>>>(d_svb, d_scb, d_slen, d_perm, d_scol, d_sv, n_slices, d_x4, d_y); }, bytes); }
	{ const int GB = (int)((n_slices + 3) / 4); const int GB8 = ((GB + 7) / 8) * 8;
	  timeit("sell64 slice/wave u4", [&] { k_sell1<4, 0><<<GB8, 256>>>(d_svb, d_scb, d_slen, d_perm, d_scol, d_sv, n_slices, d_x, d_y); }, bytes);
	  timeit("sell64 slice/wave u8", [&] { k_sell1<8, 0><<<GB8, 256>>>(d_svb, d_scb, d_slen, d_perm, d_scol, d_sv, n_slices, d_x, d_y); }, bytes);
	  timeit("sell64 b128 slice/wave u2", [&] { k_sell2<2><<<GB8, 256>>>(d_svb, d_scb, d_slen, d_perm, d_scol, d_sv, n_slices, d_x, d_y); }, bytes);
	  timeit("sell64 b128 slice/wave u4", [&] { k_sell2<4><<<GB8, 256>>>(d_svb, d_scb, d_slen, d_perm, d_scol, d_sv, n_slices, d_x, d_y); }, bytes);
	  timeit("sell64 b128 slice/wave u8", [&] { k_sell2<8><<<GB8, 256>>>(d_svb, d_scb, d_slen, d_perm, d_scol, d_sv, n_slices, d_x, d_y); }, bytes);
	  timeit("sell3 u4 nt", [&] { k_sell3<4, true, false><<<GB8, 256>>>(d_svb, d_scb, d_slen, d_perm, d_scol, d_sv, n_slices, d_x, d_y); }, bytes);
	  timeit("sell3 u12 nt", [&] { k_sell3<12, true, false><<<GB8, 256>>>(d_svb, d_scb, d_slen, d_perm, d_scol, d_sv, n_slices, d_x, d_y); }, bytes);
	  timeit("sell3 u16 nt", [&] { k_sell3<16, true, false><<<GB8, 256>>>(d_svb, d_scb, d_slen, d_perm, d_scol, d_sv, n_slices, d_x, d_y); }, bytes);
	  { const int G2 = (int)((n_slices + 1) / 2); const int G28 = ((G2 + 7) / 8) * 8;
	    timeit("sell3 u8 nt wpb2", [&] { k_sell3<8, true, false, 2><<<G28, 128>>>(d_svb, d_scb, d_slen, d_perm, d_scol, d_sv, n_slices, d_x, d_y); }, bytes); }
	  { const int G1 = (int)n_slices; const int G18 = ((G1 + 7) / 8) * 8;
	    timeit("sell3 u8 nt wpb1", [&] { k_sell3<8, true, false, 1><<<G18, 64>>>(d_svb, d_scb, d_slen, d_perm, d_scol, d_sv, n_slices, d_x, d_y); }, bytes); }
	  { const int G8 = (int)((n_slices + 7) / 8); const int G88 = ((G8 + 7) / 8) * 8;
	    timeit("sell3 u8 nt wpb8", [&] { k_sell3<8, true, false, 8><<<G88, 512>>>(d_svb, d_scb, d_slen, d_perm, d_scol, d_sv, n_slices, d_x, d_y); }, bytes); }
	  timeit("sell3 u8 nt", [&] { k_sell3<8, true, false><<<GB8, 256>>>(d_svb, d_scb, d_slen, d_perm, d_scol, d_sv, n_slices, d_x, d_y); }, bytes);
	  { double* d_x4; CK(hipMalloc(&d_x4, 8 * 4 * (n_nodes + 1))); CK(hipMemset(d_x4, 0, 8 * 4 * (n_nodes + 1)));
	  timeit("sell3 u4 padx", [&] { k_sell3<4, false, true><<<GB8, 256>>>(d_svb, d_scb, d_slen, d_perm, d_scol, d_sv, n_slices, d_x4, d_y); }, bytes);
	  timeit("sell3 u4 nt padx", [&] { k_sell3<4, true, true><<<GB8, 256>>>(d_svb, d_scb, d_slen, d_perm, d_scol, d_sv, n_slices, d_x4, d_y); }, bytes); }
	  timeit("sell3 u4 plain (=sell2)", [&] { k_sell3<4, false, false><<<GB8, 256>>>(d_svb, d_scb, d_slen, d_perm, d_scol, d_sv, n_slices, d_x, d_y); }, bytes);
	  timeit("sell64 slice/wave u4 nox", [&] { k_sell1<4, 2><<<GB8, 256>>>(d_svb, d_scb, d_slen, d_perm, d_scol, d_sv, n_slices, d_x, d_y); }, bytes); }
	timeit("sell64 unroll2 grid x2", [&] { k_sell<2, 0><<<2 * G, 256>>>(d_svb, d_scb, d_slen, d_perm, d_scol, d_sv, n_slices, d_x, d_y); }, bytes);
	return 0;
}
