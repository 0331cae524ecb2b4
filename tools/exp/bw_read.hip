// Streaming-read ceiling of the device: sum of a large buffer read with 16-byte loads, over launch shapes and load
// flavours.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/bw_read tools/exp/bw_read.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void k_read(const d2 *__restrict__ p, size_t n, double *__restrict__ out)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	double acc = 0;
	for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride)
	{
		d2 v[UNROLL];
#pragma unroll
		for (int u = 0; u < UNROLL; ++u)
			v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
		for (int u = 0; u < UNROLL; ++u)
			acc += v[u].x + v[u].y;
	}
	for (; i < n; i += stride)
		acc += p[i].x + p[i].y;
	if (acc == 1.2345e300)
		out[0] = acc;
}

// every block reads one contiguous chunk (the way a slice of the matrix is streamed by one wave)
template <int UNROLL, bool NT>
__global__ __launch_bounds__(64) void k_read_chunks(const d2 *__restrict__ p, size_t n, size_t chunk, double *__restrict__ out)
{
	const size_t b0 = (size_t)blockIdx.x * chunk;
	const size_t b1 = b0 + chunk < n ? b0 + chunk : n;
	double acc = 0;
	size_t i = b0 + threadIdx.x;
	for (; i + (UNROLL - 1) * 64 < b1; i += UNROLL * 64)
	{
		d2 v[UNROLL];
#pragma unroll
		for (int u = 0; u < UNROLL; ++u)
			v[u] = NT ? __builtin_nontemporal_load(p + i + u * 64) : p[i + u * 64];
#pragma unroll
		for (int u = 0; u < UNROLL; ++u)
			acc += v[u].x + v[u].y;
	}
	for (; i < b1; i += 64)
		acc += p[i].x + p[i].y;
	if (acc == 1.2345e300)
		out[0] = acc;
}

__global__ void k_fill(double *p, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
	{
		unsigned long long z = i * 0x9E3779B97F4A7C15ull + 12345;
		z ^= z >> 31;
		z *= 0xBF58476D1CE4E5B9ull;
		p[i] = (double)(z >> 11) / 9007199254740992.0 - 0.5;
	}
}

// the access pattern of the block-SELL slots: per slot four 16-byte loads 1 KB apart and one 8-byte load (4608 bytes per
// wave and slot); PAIR: the 8-byte entries of two slots packed into one 16-byte load (9216 bytes per two slots)
template <int U, bool PAIR>
__global__ __launch_bounds__(64) void k_read_slots(const double *__restrict__ p, size_t n_slots, size_t slots_per_block, double *__restrict__ out)
{
	const size_t s0 = (size_t)blockIdx.x * slots_per_block;
	const size_t s1 = s0 + slots_per_block < n_slots ? s0 + slots_per_block : n_slots;
	const int lane = threadIdx.x;
	double acc = 0;
	for (size_t s = s0; s + U <= s1; s += U)
	{
		d2 a[U][4];
		double b[U];
		d2 bb[U / 2];
#pragma unroll
		for (int u = 0; u < U; ++u)
		{
			const double *vj = p + (s + u) * 576;
			const d2 *vp = reinterpret_cast<const d2 *>(vj) + lane;
			a[u][0] = __builtin_nontemporal_load(vp);
			a[u][1] = __builtin_nontemporal_load(vp + 64);
			a[u][2] = __builtin_nontemporal_load(vp + 128);
			a[u][3] = __builtin_nontemporal_load(vp + 192);
			if (!PAIR)
				b[u] = __builtin_nontemporal_load(vj + 512 + lane);
		}
		if (PAIR)
#pragma unroll
			for (int u = 0; u < U / 2; ++u)
			{
				// slots 2u, 2u+1 of the batch: their ninth entries side by side after the two slots' 8 KB
				const double *vj = p + (s + 2 * u) * 576;
				bb[u] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(vj + 1024) + lane);
			}
#pragma unroll
		for (int u = 0; u < U; ++u)
			acc += a[u][0].x + a[u][0].y + a[u][1].x + a[u][1].y + a[u][2].x + a[u][2].y + a[u][3].x + a[u][3].y + (PAIR ? 0.0 : b[u]);
		if (PAIR)
#pragma unroll
			for (int u = 0; u < U / 2; ++u)
				acc += bb[u].x + bb[u].y;
	}
	if (acc == 1.2345e300)
		out[0] = acc;
}

#define CK(x)                                                                  \
	do                                                                         \
	{                                                                          \
		hipError_t e = (x);                                                    \
		if (e != hipSuccess)                                                   \
		{                                                                      \
			printf("%s: %s\n", #x, hipGetErrorString(e));                      \
			return 1;                                                          \
		}                                                                      \
	} while (0)

template <class F>
double time_ms(F f, int reps)
{
	hipEvent_t a, b;
	hipEventCreate(&a);
	hipEventCreate(&b);
	f();
	hipDeviceSynchronize();
	hipEventRecord(a);
	for (int r = 0; r < reps; ++r)
		f();
	hipEventRecord(b);
	hipEventSynchronize(b);
	float ms = 0;
	hipEventElapsedTime(&ms, a, b);
	return ms / reps;
}

int main(int argc, char **argv)
{
	const double gb = argc > 1 ? atof(argv[1]) : 16.0;
	const size_t n = (size_t)(gb * 1e9 / 16);
	d2 *p;
	double *out;
	CK(hipMalloc(&p, n * 16));
	CK(hipMalloc(&out, 8));
	if (argc > 2)
		hipLaunchKernelGGL(k_fill, dim3(65536), dim3(256), 0, 0, (double *)p, n * 2); // pseudo-random doubles instead of zeros
	else
		CK(hipMemset(p, 0, n * 16));
	CK(hipDeviceSynchronize());
	const int reps = 5;
	for (int blocks : {2048, 4096, 8192, 16384, 65536})
	{
		double t1 = time_ms([&] { hipLaunchKernelGGL((k_read<4, false>), dim3(blocks), dim3(256), 0, 0, p, n, out); }, reps);
		double t2 = time_ms([&] { hipLaunchKernelGGL((k_read<8, false>), dim3(blocks), dim3(256), 0, 0, p, n, out); }, reps);
		double t3 = time_ms([&] { hipLaunchKernelGGL((k_read<8, true>), dim3(blocks), dim3(256), 0, 0, p, n, out); }, reps);
		double t4 = time_ms([&] { hipLaunchKernelGGL((k_read<16, true>), dim3(blocks), dim3(256), 0, 0, p, n, out); }, reps);
		printf("grid-stride %6d blocks x 256: unroll4 %.0f GB/s, unroll8 %.0f, unroll8 nt %.0f, unroll16 nt %.0f\n", blocks, n * 16 / t1 / 1e6, n * 16 / t2 / 1e6,
			   n * 16 / t3 / 1e6, n * 16 / t4 / 1e6);
	}
	for (size_t kb : {64, 256, 1024})
	{
		const size_t chunk = kb * 1024 / 16;
		const unsigned blocks = (unsigned)((n + chunk - 1) / chunk);
		double t1 = time_ms([&] { hipLaunchKernelGGL((k_read_chunks<12, false>), dim3(blocks), dim3(64), 0, 0, p, n, chunk, out); }, reps);
		double t2 = time_ms([&] { hipLaunchKernelGGL((k_read_chunks<12, true>), dim3(blocks), dim3(64), 0, 0, p, n, chunk, out); }, reps);
		double t3 = time_ms([&] { hipLaunchKernelGGL((k_read_chunks<24, true>), dim3(blocks), dim3(64), 0, 0, p, n, chunk, out); }, reps);
		printf("one wave per %4zu KB chunk (%u blocks): unroll12 %.0f GB/s, unroll12 nt %.0f, unroll24 nt %.0f\n", kb, blocks, n * 16 / t1 / 1e6, n * 16 / t2 / 1e6,
			   n * 16 / t3 / 1e6);
	}
	{
		const size_t n_slots = n * 16 / 4608;
		for (size_t spb : {24, 48, 96})
		{
			const unsigned blocks = (unsigned)((n_slots + spb - 1) / spb);
			double t1 = time_ms([&] { hipLaunchKernelGGL((k_read_slots<6, false>), dim3(blocks), dim3(64), 0, 0, (const double *)p, n_slots, spb, out); }, reps);
			double t2 = time_ms([&] { hipLaunchKernelGGL((k_read_slots<12, false>), dim3(blocks), dim3(64), 0, 0, (const double *)p, n_slots, spb, out); }, reps);
			double t3 = time_ms([&] { hipLaunchKernelGGL((k_read_slots<6, true>), dim3(blocks), dim3(64), 0, 0, (const double *)p, n_slots, spb, out); }, reps);
			double t4 = time_ms([&] { hipLaunchKernelGGL((k_read_slots<12, true>), dim3(blocks), dim3(64), 0, 0, (const double *)p, n_slots, spb, out); }, reps);
			const double bytes = (double)n_slots * 4608;
			printf("SELL slots, %zu per wave (%u blocks): 4x16B+8B unroll6 %.0f GB/s, unroll12 %.0f; paired ninth entries unroll6 %.0f, unroll12 %.0f\n", spb, blocks,
				   bytes / t1 / 1e6, bytes / t2 / 1e6, bytes / t3 / 1e6, bytes / t4 / 1e6);
		}
	}
	return 0;
}
