// What bounds the covariance fill (4 N^2 bytes of lower tiles, 66 us per N=8192 matrix = 4.0 TB/s)?  The store shape of the
// Gram-form fill (per wave-instruction 4 rows x 128 bytes: the matrix unit's D layout) against whole 512-byte row pieces,
// with no arithmetic and with a dependent fp64 chain of the fill's length per element.  16 matrices of N=8192, lower tiles.
// hipcc --offload-arch=gfx950 -O3 fill_shape.hip -o fill_shape
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void lower_tile(long t, int &tr, int &tc)
{
	int r = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
	while ((long)r * (r + 1) / 2 > t) r--;
	while ((long)(r + 1) * (r + 2) / 2 <= t) r++;
	tr = r; tc = (int)(t - (long)r * (r + 1) / 2);
}
template <int SHAPE, int OPS>
__global__ __launch_bounds__(256) void fill(double *T, long ld, long bstride, int nt, double seed)
{
	double *out = T + (long)blockIdx.y * bstride;
	const long ntiles = (long)nt * (nt + 1) / 2;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	for (int i = 0; i < 4; i++) {
		const long t = ntiles - 1 - ((long)blockIdx.x * 4 + i);
		if (t < 0) break;
		int tr, tc; lower_tile(t, tr, tc);
		double v[16];
#pragma unroll
		for (int e = 0; e < 16; e++) {
			double x = seed + (double)(lane + e);
#pragma unroll
			for (int o = 0; o < OPS; o++) x = fma(x, 0.999, 1e-3);     // a dependent chain, two elements interleaved by the compiler at best
			v[e] = x;
		}
		if (SHAPE == 0) {
			const int q = lane & 15, g = lane >> 4;
#pragma unroll
			for (int r = 0; r < 4; r++) {
				double *orow = out + (long)(tr * 64 + 16 * wave + g + 4 * r) * ld + tc * 64 + q;
#pragma unroll
				for (int j = 0; j < 4; j++) orow[16 * j] = v[4 * r + j];
			}
		} else {
			const int crow = lane >> 5, ccol = 2 * (lane & 31);
#pragma unroll
			for (int u = 0; u < 8; u++) {
				d2_t w = {v[2 * u], v[2 * u + 1]};
				*reinterpret_cast<d2_t *>(out + (long)(tr * 64 + 16 * wave + 2 * u + crow) * ld + tc * 64 + ccol) = w;
			}
		}
	}
}
int main()
{
	const int N = 8192, B = 16, nt = N / 64;
	const long ld = N, bstride = (long)(N + 64) * N;
	double *T; hipMalloc(&T, (size_t)B * bstride * 8);
	hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const long ntiles = (long)nt * (nt + 1) / 2;
	dim3 grid((unsigned)((ntiles + 3) / 4), B);
#define RUN(SHAPE, OPS, name) { float best = 1e30f; for (int rep = 0; rep < 5; rep++) { hipEventRecord(e0, s); hipLaunchKernelGGL((fill<SHAPE, OPS>), grid, dim3(256), 0, s, T, ld, bstride, nt, 1.0 + rep); hipEventRecord(e1, s); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms; } \
	printf("%-70s %6.1f us per matrix = %.2f TB/s\n", name, best * 1e3 / B, (double)ntiles * 4096 * 8 / (best * 1e-3 / B) / 1e12); }
	RUN(0, 0, "D-layout stores (4 rows x 128 B per wave-instruction), no arithmetic");
	RUN(1, 0, "whole 512-byte row pieces (2 rows per wave-instruction), no arithmetic");
	RUN(0, 26, "D-layout stores, 26 dependent fp64 operations per element");
	RUN(1, 26, "row pieces, 26 dependent fp64 operations per element");
	RUN(0, 13, "D-layout stores, 13 dependent fp64 operations per element");
	RUN(1, 13, "row pieces, 13 dependent fp64 operations per element");
	return 0;
}
