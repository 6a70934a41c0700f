// fp64 VALU issue rate on gfx950: C independent v_fma_f64 chains per wave, W waves per SIMD, every SIMD of the chip busy.
// prints cycles per wave-instruction per SIMD (s_memtime deltas of wave 0 of each workgroup, averaged) and the lane-FMA rate.
//   hipcc --offload-arch=gfx950 -O3 -o fma_rate fma_rate.hip && ./fma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int C>
__global__ __launch_bounds__(256) void chains(double *out, unsigned long long *clk, int iters, double a, double b)
{
	double x[C];
#pragma unroll
	for (int c = 0; c < C; c++) x[c] = threadIdx.x * 1e-3 + c;
	const unsigned long long t0 = clock64();
	for (int i = 0; i < iters; i++) {
#pragma unroll
		for (int u = 0; u < 8; u++)
#pragma unroll
			for (int c = 0; c < C; c++) x[c] = __builtin_fma(x[c], a, b);
	}
	const unsigned long long t1 = clock64();
	double s = 0;
#pragma unroll
	for (int c = 0; c < C; c++) s += x[c];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
template <int C>
static void run(int wgs_per_cu)
{
	const int iters = 2000, ncu = 256, grid = ncu * wgs_per_cu;
	double *out; unsigned long long *clk;
	hipMalloc(&out, (size_t)grid * 256 * 8); hipMalloc(&clk, (size_t)grid * 8);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(chains<C>, dim3(grid), dim3(256), 0, 0, out, clk, 10, 0.999, 1e-3);
	hipEventRecord(e0);
	hipLaunchKernelGGL(chains<C>, dim3(grid), dim3(256), 0, 0, out, clk, iters, 0.999, 1e-3);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	std::vector<unsigned long long> h(grid);
	hipMemcpy(h.data(), clk, (size_t)grid * 8, hipMemcpyDeviceToHost);
	double avg = 0; for (auto v : h) avg += (double)v; avg /= grid;
	const double instr_per_wave = (double)iters * 8 * C;
	// each SIMD holds wgs_per_cu waves (a 256-thread workgroup = one wave per SIMD)
	printf("chains %d  waves/SIMD %d : %.2f clocks per wave-instruction per SIMD (s_memtime, 100 MHz-independent shader clock), "
	       "%.2f TFMA-lanes/s = %.1f TFLOP/s\n", C, wgs_per_cu, avg / (instr_per_wave * wgs_per_cu),
	       instr_per_wave * 64.0 * 4 * grid / (ms * 1e-3) / 1e12, 2 * instr_per_wave * 64.0 * 4 * grid / (ms * 1e-3) / 1e12);
	hipFree(out); hipFree(clk);
}
int main()
{
	for (int w : {1, 2, 4, 8}) { run<1>(w); run<2>(w); run<4>(w); run<8>(w); }
	return 0;
}
