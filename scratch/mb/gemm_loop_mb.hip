// What does the GEMM inner loop cost beyond its MFMAs?  One wave = 8 accumulator tiles (2 A x 4 B fragments), as
// the 128x128 / 8-wave configuration of gemm_nt_kernel; per "quarter": 8 MFMAs.  Variants add, one at a time:
//   mode 0: MFMAs only (fragments constant)
//   mode 1: + 6 ds_read_b64 per quarter, issued one quarter ahead (as the pipelined kernel)
//   mode 2: + a workgroup barrier every 4 quarters (one k-step of 16)
//   mode 3: + 4 ds_write_b128 per k-step before the barrier
//   mode 4: + 4 global_load_dwordx4 per k-step feeding those writes (the kernel's register staging, L2-resident source)
//   mode 5: the same bytes by 4 global_load_lds_dwordx4 (LDS-DMA, no VGPRs, no ds_write) instead of mode 4's pair
// 512 threads per workgroup, 1 or 2 workgroups per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(512) void k(double *out, int ksteps, long long *cyc, const double *src)
{
	__shared__ double lds[2][2 * 128 * 18];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	for (int i = tid; i < 2 * 2 * 128 * 18; i += 512) (&lds[0][0])[i] = 1.0 + 1e-9 * i;
	__syncthreads();
	d4_t acc[8];
	for (int i = 0; i < 8; i++) acc[i] = (d4_t){0, 0, 0, 0};
	const int a_base = ((wave >> 1) * 32 + (lane & 15)) * 18 + (lane >> 4);
	const int b_base = 128 * 18 + ((wave & 1) * 64 + (lane & 15)) * 18 + (lane >> 4);
	const int wofs = (tid >> 3) * 18 + 2 * (tid & 7);
	double fa[2][2], fb[2][4];
	for (int i = 0; i < 2; i++) fa[0][i] = fa[1][i] = 1.0 + lane * 1e-9;
	for (int j = 0; j < 4; j++) fb[0][j] = fb[1][j] = 1.0 - lane * 1e-9;
	d2_t st[4] = {{1, 2}, {3, 4}, {5, 6}, {7, 8}};
	const double *gp = src + (size_t)(blockIdx.x & 63) * 8192 + tid * 2;      // 64 KB per workgroup slot, L2-resident
	long long t0 = __builtin_amdgcn_s_memtime();
	int cur = 0;
	for (int ks = 0; ks < ksteps; ks++) {
		const double *buf = lds[cur];
		if (MODE == 4) {
#pragma unroll
			for (int it = 0; it < 4; it++) st[it] = *reinterpret_cast<const d2_t *>(gp + it * 1024 + (ks & 7) * 4096 % 4096);
		}
		if (MODE == 5) {
#pragma unroll
			for (int it = 0; it < 4; it++)
				__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gp + it * 1024),
				                                 (__attribute__((address_space(3))) void *)(&lds[cur ^ 1][(wave * 4 + it) * 128]), 16, 0, 0);
		}
#pragma unroll
		for (int q = 0; q < 4; q++) {
			const int set = q & 1;
			if (MODE >= 1) {
#pragma unroll
				for (int i = 0; i < 2; i++) fa[set ^ 1][i] = buf[a_base + i * 16 * 18 + 4 * ((q + 1) & 3)];
#pragma unroll
				for (int j = 0; j < 4; j++) fb[set ^ 1][j] = buf[b_base + j * 16 * 18 + 4 * ((q + 1) & 3)];
				__builtin_amdgcn_sched_barrier(0);
			}
			if (q == 3) {
				if (MODE == 3 || MODE == 4) {
#pragma unroll
					for (int it = 0; it < 4; it++) *reinterpret_cast<d2_t *>(&lds[cur ^ 1][wofs + it * 64 * 18]) = st[it];
				}
				if (MODE >= 2) __syncthreads();
			}
#pragma unroll
			for (int i = 0; i < 2; i++)
#pragma unroll
				for (int j = 0; j < 4; j++)
					asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i * 4 + j]) : "v"(fa[set][i]), "v"(fb[set][j]));
			if (MODE >= 1) __builtin_amdgcn_sched_barrier(0);
		}
		if (MODE >= 3) cur ^= 1;
	}
	long long t1 = __builtin_amdgcn_s_memtime();
	double s = 0;
	for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
	out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
	if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

static double *g_src;
template <int MODE>
void run(double *out, long long *c, int wgs_per_cu)
{
	const int ksteps = 4000;
	const int blocks = 256 * wgs_per_cu;
	for (int rep = 0; rep < 2; rep++) {
		hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
		hipEventRecord(e0, 0);
		hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, out, ksteps, c, g_src);
		hipEventRecord(e1, 0); hipDeviceSynchronize();
		float ms; hipEventElapsedTime(&ms, e0, e1);
		long long hc; hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
		const double flops = (double)blocks * 8 * ksteps * 32 * 2048.0;
		if (rep) printf("mode %d wgs/cu %d: %.3f ms  %.1f TF/s  cycles per MFMA slot (wave 0) %.1f\n", MODE, wgs_per_cu, ms, flops / ms / 1e9,
		                (double)hc / (ksteps * 32.0) / (2.0 * wgs_per_cu));
	}
}

int main()
{
	double *out; long long *c;
	hipMalloc(&out, 8 * 512 * 1024); hipMalloc(&c, 8);
	hipMalloc(&g_src, 64 * 8192 * 8 + 65536); hipMemset(g_src, 0, 64 * 8192 * 8 + 65536);
	for (int w = 1; w <= 2; w++) { run<0>(out, c, w); run<1>(out, c, w); run<2>(out, c, w); run<3>(out, c, w); run<4>(out, c, w); run<5>(out, c, w); }
	return 0;
}
