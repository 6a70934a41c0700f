// What bounds the leaf solve in a lock-step batch?  The product kernel (both I/O forms) against variants without the diagonal
// inverses, without the solve chain, without both, on block columns of the N=4096 B=64 and N=8192 B=16 shapes.
// hipcc --offload-arch=gfx950 -O3 -I ../../include -I ../../madaiemulator_amd/csrc/hip leaf_variants.hip -o leaf_variants
#include "../../madaiemulator_amd/csrc/hip/kernels_linalg.hip"
#include <cstdio>
using namespace gpemu;
__global__ void init_kernel(double *T, long n, long ld)
{
	for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
		const long r = (i / ld) % 8256, c = i % ld;
		T[i] = (r % 4160 == c) ? 4.0 : 1e-3 * (double)((r * 7 + c * 13) % 17 - 8);
	}
}
template <bool ST, int DBG> static double run(hipStream_t s, double *T, int N, int B, long bstride, hipEvent_t e0, hipEvent_t e1, long *bytes_out)
{
	double best = 1e30;
	for (int rep = 0; rep < 4; rep++) {
		long bytes = 0;
		hipEventRecord(e0, s);
		for (int c0 = 0; c0 + 128 < N; c0 += 256) {
			const int m = N + 64 - c0 - 64;
			hipLaunchKernelGGL((leaf_solve_kernel<ST, false, DBG>), dim3((m + 63) / 64, B), dim3(256), 0, s, T, (long)N, c0, m, (unsigned long long *)nullptr, bstride, -1);
			bytes += (long)m * 64 * 16 * B;
		}
		hipEventRecord(e1, s); hipEventSynchronize(e1);
		float ms = 0; hipEventElapsedTime(&ms, e0, e1);
		if (rep && ms < best) best = ms;
		*bytes_out = bytes;
	}
	return best;
}
int main()
{
	hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	struct Case { int N, B; };
	for (Case cs : {Case{4096, 64}, Case{8192, 16}}) {
		const int N = cs.N, B = cs.B;
		const long bstride = (long)(N + 64) * N;
		double *T; hipMalloc(&T, (size_t)B * bstride * 8);
		init_kernel<<<4096, 256, 0, s>>>(T, (long)B * bstride, N);
		long by = 0;
#define ROW(ST, DBG, name) { double ms = run<ST, DBG>(s, T, N, B, bstride, e0, e1, &by); printf("N=%d B=%d %-58s %8.1f us per launch  %.2f TB/s\n", N, B, name, ms * 1e3 / ((N - 128 + 255) / 256), by / (ms * 1e-3) / 1e12); }
		ROW(false, 0, "element-wise I/O, full kernel");
		ROW(true, 0, "staged I/O, full kernel");
		ROW(true, 1, "staged I/O, no diagonal inverses");
		ROW(true, 2, "staged I/O, no chain");
		ROW(true, 3, "staged I/O, neither (L staged, tile through the strip)");
		ROW(false, 3, "element-wise I/O, neither");
		hipFree(T);
	}
	return 0;
}
