#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4_t __attribute__((ext_vector_type(4)));
template<int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters, double x, double y, long long* cyc)
{
	d4_t acc[NACC];
	for (int i = 0; i < NACC; i++) acc[i] = (d4_t){0, 0, 0, 0};
	double a = x + threadIdx.x * 1e-9, b = y - threadIdx.x * 1e-9;
	long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int i = 0; i < NACC; i++)
			asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
	}
	long long t1 = __builtin_amdgcn_s_memtime();
	double s = 0;
	for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
	double* out; long long* c; hipMalloc(&out, 8 * 256 * 4096); hipMalloc(&c, 8);
	const int iters = 20000;
	for (int wgs_per_cu = 1; wgs_per_cu <= 4; wgs_per_cu *= 2) {
		for (int rep = 0; rep < 2; rep++) {
			int blocks = 256 * wgs_per_cu;
			hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
			hipEventRecord(e0, 0);
			hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001, 0.9999, c);
			hipEventRecord(e1, 0); hipDeviceSynchronize();
			float ms; hipEventElapsedTime(&ms, e0, e1);
			long long hc; hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
			double flops = (double)blocks * 4 * iters * 8 * 2048.0;
			printf("wgs/cu %d: %.3f ms  %.1f TF/s  cycles/mfma (wave0) %.1f  clock est %.2f GHz\n", wgs_per_cu, ms, flops / ms / 1e9,
			       (double)hc / (iters * 8.0), hc / (ms * 1e6));
		}
	}
	return 0;
}
