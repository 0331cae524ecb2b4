// Checks where the lanes' data of a 16-byte global->LDS load land: expected LDS base + lane * 16.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d2_t __attribute__((ext_vector_type(2)));
__global__ void k(const double *__restrict__ src, const int *__restrict__ idx, double *__restrict__ out, int nact)
{
	__shared__ d2_t buf[2][64];
	const int lane = threadIdx.x;
	buf[0][lane] = d2_t{-1.0, -1.0};
	buf[1][lane] = d2_t{-2.0, -2.0};
	__syncthreads();
	if (lane < nact)
	{
		const double *p = src + 2 * (long)idx[lane];
		__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p, (__attribute__((address_space(3))) void *)&buf[0][0], 16, 0, 0);
		__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p + 2), (__attribute__((address_space(3))) void *)&buf[1][0], 16, 0, 0);
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	out[4 * lane + 0] = buf[0][lane].x;
	out[4 * lane + 1] = buf[0][lane].y;
	out[4 * lane + 2] = buf[1][lane].x;
	out[4 * lane + 3] = buf[1][lane].y;
}
int main()
{
	const int n = 4096;
	std::vector<double> h(2 * n);
	for (int i = 0; i < 2 * n; ++i)
		h[i] = i;
	std::vector<int> idx(64);
	for (int i = 0; i < 64; ++i)
		idx[i] = (i * 37 + 11) % (n - 2);
	double *d_src, *d_out;
	int *d_idx;
	hipMalloc(&d_src, sizeof(double) * 2 * n);
	hipMalloc(&d_out, sizeof(double) * 256);
	hipMalloc(&d_idx, sizeof(int) * 64);
	hipMemcpy(d_src, h.data(), sizeof(double) * 2 * n, hipMemcpyHostToDevice);
	hipMemcpy(d_idx, idx.data(), sizeof(int) * 64, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_src, d_idx, d_out, 50);
	std::vector<double> o(256);
	if (hipMemcpy(o.data(), d_out, sizeof(double) * 256, hipMemcpyDeviceToHost) != hipSuccess)
	{
		printf("copy failed\n");
		return 1;
	}
	int bad = 0;
	for (int l = 0; l < 64; ++l)
	{
		const double e0 = l < 50 ? 2.0 * idx[l] : -1.0, e2 = l < 50 ? 2.0 * idx[l] + 2 : -2.0;
		if (o[4 * l] != e0 || o[4 * l + 1] != (l < 50 ? e0 + 1 : -1.0) || o[4 * l + 2] != e2 || o[4 * l + 3] != (l < 50 ? e2 + 1 : -2.0))
		{
			if (bad < 8)
				printf("lane %d: got %g %g %g %g expected %g.. %g..\n", l, o[4 * l], o[4 * l + 1], o[4 * l + 2], o[4 * l + 3], e0, e2);
			++bad;
		}
	}
	printf(bad ? "MISMATCH in %d lanes\n" : "lds dma layout ok (%d)\n", bad);
	return bad != 0;
}
