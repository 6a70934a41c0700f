// What slows a latency-bound fp64 chain that runs beside a matrix-instruction kernel on the same CUs?
// (look-ahead schedule.)  Chain kernels: one wave per workgroup, 256 workgroups:
//   valu : dependent v_fma_f64 chain in registers       lds : dependent LDS store -> load chain
//   mem  : dependent global loads (pointer chase in a 1 MB, L2-resident ring)
// Background kernels (second stream, low priority, run until a flag in pinned host memory is set):
//   mfma W : W waves per SIMD of back-to-back v_mfma_f64_16x16x4_f64, no memory
//   lds    : 8 waves per CU of ds_read_b128 loops        stream : 16-byte global loads over a 2 GB buffer
//   (the stop flag lives in device memory)
// Output: cycles per chain step alone and beside each background, with and without s_setprio 3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef double d4_t __attribute__((ext_vector_type(4)));

__global__ void chain_valu(double *out, unsigned long long *cyc, int steps, int prio)
{
	if (prio) __builtin_amdgcn_s_setprio(3);
	double x = 1.0 + threadIdx.x * 1e-9, a = 0.999999, b = 1e-7;
	unsigned long long t0 = clock64();
	for (int i = 0; i < steps; i++) { x = fma(x, a, b); x = fma(x, a, b); x = fma(x, a, b); x = fma(x, a, b); }
	unsigned long long t1 = clock64();
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
	out[blockIdx.x * 64 + threadIdx.x] = x;
}
__global__ void chain_lds(double *out, unsigned long long *cyc, int steps, int prio)
{
	if (prio) __builtin_amdgcn_s_setprio(3);
	__shared__ double s[64 * 2];
	double x = 1.0 + threadIdx.x;
	s[threadIdx.x] = x;
	unsigned long long t0 = clock64();
	for (int i = 0; i < steps; i++) {
		s[threadIdx.x] = x; __builtin_amdgcn_s_waitcnt(0); x = s[(threadIdx.x + 1) & 63] + 1.0;
		s[threadIdx.x + 64] = x; __builtin_amdgcn_s_waitcnt(0); x = s[64 + ((threadIdx.x + 1) & 63)] + 1.0;
	}
	unsigned long long t1 = clock64();
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
	out[blockIdx.x * 64 + threadIdx.x] = x;
}
__global__ void chain_mem(const int *ring, double *out, unsigned long long *cyc, int steps, int prio)
{
	if (prio) __builtin_amdgcn_s_setprio(3);
	int p = (blockIdx.x * 977 + threadIdx.x * 16) & 0x3ffff;
	unsigned long long t0 = clock64();
	for (int i = 0; i < steps; i++) { p = ring[p]; p = ring[p]; }
	unsigned long long t1 = clock64();
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
	out[blockIdx.x * 64 + threadIdx.x] = p;
}
__global__ void bg_mfma(double *out, volatile int *stop)
{
	d4_t acc[8];
	for (int i = 0; i < 8; i++) acc[i] = (d4_t){0.0, 0.0, 0.0, 0.0};
	double a = 1.0 + threadIdx.x * 1e-6, b = 0.5;
	while (true) {
		for (int it = 0; it < 64; it++)
#pragma unroll
			for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
		if (*stop) break;
	}
	double s = 0;
	for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void bg_lds(double *out, volatile int *stop)
{
	__shared__ double s[8192];
	for (int i = threadIdx.x; i < 8192; i += blockDim.x) s[i] = i;
	__syncthreads();
	double x = 0;
	int p = threadIdx.x * 2;
	while (true) {
		for (int it = 0; it < 256; it++) { x += s[p & 8190] + s[(p + 1) & 8191]; p += 130; }
		if (*stop) break;
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
__global__ void bg_stream(const double *buf, size_t n, double *out, volatile int *stop)
{
	double x = 0;
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	while (true) {
		for (int it = 0; it < 64; it++) { x += buf[(2 * i) % n] + buf[(2 * i + 1) % n]; i += stride; }
		if (*stop) break;
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

int main()
{
	hipStream_t hi, lo;
	int least, greatest;
	CHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
	CHK(hipStreamCreateWithPriority(&hi, hipStreamNonBlocking, greatest));
	CHK(hipStreamCreateWithPriority(&lo, hipStreamNonBlocking, least));
	double *out, *outbg, *big; unsigned long long *cyc; int *ring; int *stop;
	CHK(hipMalloc(&out, 256 * 64 * 8)); CHK(hipMalloc(&outbg, (size_t)4096 * 1024 * 8)); CHK(hipMalloc(&cyc, 256 * 8));
	const size_t nbig = (size_t)1 << 28;                      // 2 GB of doubles
	CHK(hipMalloc(&big, nbig * 8)); CHK(hipMemset(big, 0, nbig * 8));
	std::vector<int> h(1 << 18);
	for (int i = 0; i < (1 << 18); i++) h[i] = (int)(((long)i * 40503 + 12345) & 0x3ffff);
	CHK(hipMalloc(&ring, h.size() * 4)); CHK(hipMemcpy(ring, h.data(), h.size() * 4, hipMemcpyHostToDevice));
	CHK(hipMalloc((void **)&stop, 4)); CHK(hipMemset(stop, 0, 4));          // device flag: polling pinned host memory from every wave
	hipStream_t ctl; CHK(hipStreamCreateWithFlags(&ctl, hipStreamNonBlocking));   // floods the fabric and starves every other load
	const int steps = 20000;
	auto run_chain = [&](int which, int prio) {
		if (which == 0) hipLaunchKernelGGL(chain_valu, dim3(256), dim3(64), 0, hi, out, cyc, steps, prio);
		if (which == 1) hipLaunchKernelGGL(chain_lds, dim3(256), dim3(64), 0, hi, out, cyc, steps / 4, prio);
		if (which == 2) hipLaunchKernelGGL(chain_mem, dim3(256), dim3(64), 0, hi, ring, out, cyc, steps / 8, prio);
		CHK(hipStreamSynchronize(hi));
		std::vector<unsigned long long> c(256);
		CHK(hipMemcpy(c.data(), cyc, 256 * 8, hipMemcpyDeviceToHost));
		double s = 0; for (auto v : c) s += (double)v;
		const double per = s / 256 / (which == 0 ? steps * 4.0 : which == 1 ? steps / 4 * 2.0 : steps / 8 * 2.0);
		return per;
	};
	const char *cname[] = {"valu fp64 fma", "lds store->load", "global pointer chase"};
	const char *bname[] = {"alone", "mfma 1 wave/SIMD", "mfma 2 waves/SIMD", "mfma 4 waves/SIMD", "lds readers 8 waves/CU", "global stream 16 waves/CU"};
	for (int bg = 0; bg < 6; bg++) {
		CHK(hipMemsetAsync(stop, 0, 4, ctl)); CHK(hipStreamSynchronize(ctl));
		if (bg == 1) hipLaunchKernelGGL(bg_mfma, dim3(256), dim3(256), 0, lo, outbg, stop);
		if (bg == 2) hipLaunchKernelGGL(bg_mfma, dim3(256), dim3(512), 0, lo, outbg, stop);
		if (bg == 3) hipLaunchKernelGGL(bg_mfma, dim3(256), dim3(1024), 0, lo, outbg, stop);
		if (bg == 4) hipLaunchKernelGGL(bg_lds, dim3(256), dim3(512), 0, lo, outbg, stop);
		if (bg == 5) hipLaunchKernelGGL(bg_stream, dim3(256 * 4), dim3(256), 0, lo, big, nbig, outbg, stop);
		for (int which = 0; which < 3; which++)
			for (int prio = 0; prio < 2; prio++) {
				run_chain(which, prio);
				const double per = run_chain(which, prio);
				printf("%-28s beside %-28s prio %d : %8.1f cycles per step\n", cname[which], bname[bg], prio * 3, per);
				fflush(stdout);
			}
		CHK(hipMemsetAsync(stop, 1, 4, ctl)); CHK(hipStreamSynchronize(ctl));
		CHK(hipStreamSynchronize(lo));
	}
	return 0;
}
