// Does the leading dimension of the tall matrices (a power of two: 4096 or 8192 doubles) cost HBM bandwidth in the
// block-column kernels of the factorisation?  Read-modify-write of a 64-column block column (the leaf solve's traffic) and
// read A + read-modify-write C (the K=64 update's traffic) for several leading dimensions and two access shapes:
//   P1 = the leaf solve's: lane (q, g) touches row q, doubles 16 j + g + 4 r  (16 rows x 32 B per wave-instruction)
//   P2 = coalesced: 16 B per lane, 32 lanes per 512-byte row piece, 2 rows per wave-instruction
// hipcc --offload-arch=gfx950 -O2 ld_stride.hip -o ld_stride
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d2_t __attribute__((ext_vector_type(2)));

template <int PAT, int UPD>
__global__ __launch_bounds__(256) void col_rmw(double *T, long ld, int c0, int m, long bstride)
{
	T += (long)blockIdx.y * bstride;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int row0 = (blockIdx.x * 4 + wave) * 16;
	if (row0 >= m) return;
	if (PAT == 1) {
		const int q = lane & 15, g = lane >> 4;
		double *bp = T + (long)(c0 + 64 + row0 + q) * ld + c0 + (UPD ? 64 : 0);
		double v[16];
#pragma unroll
		for (int j = 0; j < 4; j++)
#pragma unroll
			for (int r = 0; r < 4; r++) v[4 * j + r] = bp[16 * j + g + 4 * r];
		if (UPD) {
			const double *ap = bp - 64;
#pragma unroll
			for (int j = 0; j < 4; j++)
#pragma unroll
				for (int r = 0; r < 4; r++) v[4 * j + r] += ap[16 * j + g + 4 * r];
		}
#pragma unroll
		for (int j = 0; j < 4; j++)
#pragma unroll
			for (int r = 0; r < 4; r++) bp[16 * j + g + 4 * r] = v[4 * j + r] * 1.0000001;
	} else {
		const int c = (lane & 31) * 2, rr = lane >> 5;
		double *bp = T + (long)(c0 + 64 + row0 + rr) * ld + c0 + c + (UPD ? 64 : 0);
		d2_t v[8];
#pragma unroll
		for (int u = 0; u < 8; u++) v[u] = *reinterpret_cast<const d2_t *>(bp + (long)(2 * u) * ld);
		if (UPD) {
#pragma unroll
			for (int u = 0; u < 8; u++) v[u] += *reinterpret_cast<const d2_t *>(bp - 64 + (long)(2 * u) * ld);
		}
#pragma unroll
		for (int u = 0; u < 8; u++) *reinterpret_cast<d2_t *>(bp + (long)(2 * u) * ld) = v[u] * 1.0000001;
	}
}

int main()
{
	hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	struct Case { int N, B; };
	for (Case cs : {Case{4096, 64}, Case{8192, 16}}) {
		const int N = cs.N, B = cs.B;
		for (int pad : {0, 8, 16, 32, 64, 128, 192}) {
			const long ld = N + pad;
			const long rows = N + 64;
			const long bstride = rows * ld + (pad ? 0 : 0);
			double *T; hipMalloc(&T, (size_t)B * bstride * 8); hipMemsetAsync(T, 0, (size_t)B * bstride * 8, s);
			for (int upd = 0; upd < 2; upd++)
				for (int pat = 1; pat <= 2; pat++) {
					double best = 1e30, sum = 0; int cnt = 0;
					// the block columns of a whole factorisation's leaf level: c0 = 0, 64, ... (every fourth, to keep it short)
					for (int rep = 0; rep < 3; rep++) {
						hipEventRecord(e0, s);
						long bytes = 0;
						for (int c0 = 0; c0 + 128 < N; c0 += 256) {
							const int m = (int)rows - c0 - 64;
							dim3 grid((m + 63) / 64, B);
							if (upd == 0 && pat == 1) col_rmw<1, 0><<<grid, 256, 0, s>>>(T, ld, c0, m, bstride);
							if (upd == 0 && pat == 2) col_rmw<2, 0><<<grid, 256, 0, s>>>(T, ld, c0, m, bstride);
							if (upd == 1 && pat == 1) col_rmw<1, 1><<<grid, 256, 0, s>>>(T, ld, c0, m, bstride);
							if (upd == 1 && pat == 2) col_rmw<2, 1><<<grid, 256, 0, s>>>(T, ld, c0, m, bstride);
							bytes += (long)m * 64 * 8 * (upd ? 3 : 2) * B;
						}
						hipEventRecord(e1, s); hipEventSynchronize(e1);
						float ms = 0; hipEventElapsedTime(&ms, e0, e1);
						const double tbs = bytes / (ms * 1e-3) / 1e12;
						if (rep) { sum += tbs; cnt++; if (ms < best) best = ms; }
					}
					printf("N=%d B=%d ld=N+%-3d %s pattern %s: %.2f TB/s\n", N, B, pad, upd ? "A + RMW C (update)" : "RMW (leaf solve)  ",
					       pat == 1 ? "P1 16 rows x 32 B" : "P2 coalesced     ", sum / cnt);
				}
			hipFree(T);
		}
	}
	return 0;
}
