// micro-benchmark of leaf factor variants: build with hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
typedef double d2_t __attribute__((ext_vector_type(2)));
constexpr int LEAF=64, LP=66;
__device__ __forceinline__ double bcast_lane(double v, int srclane)
{
	int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
	int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
	return __hiloint2double(hi, lo);
}
// VAR 0: full; 1: chain only (no c>=k+2 updates); 2: updates only with constant pivots (no rsqrt chain)
template<int VAR>
__global__ __launch_bounds__(64) void k_factor(double *T, long ld, long long* cyc)
{
	__shared__ double Lt[LEAF * LP];
	const int lane = threadIdx.x;
	long long t0 = __builtin_amdgcn_s_memtime();
	double a[LEAF];
	const double *rp = T + (long)lane * ld;
#pragma unroll
	for (int k = 0; k < LEAF; k += 2) { d2_t v = *reinterpret_cast<const d2_t *>(rp + k); a[k] = v[0]; a[k + 1] = v[1]; }
	long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
	for (int k = 0; k < LEAF; k++) {
		double rs;
		if (VAR == 2) { rs = 0.5; }
		else {
			const double p = bcast_lane(a[k], k);
			rs = __builtin_amdgcn_rsq(p);
			double t = p * rs; double e = fma(-t, rs, 1.0); rs = fma(rs * 0.5, e, rs); t = p * rs; e = fma(-t, rs, 1.0); rs = fma(rs * 0.5, e, rs);
		}
		const double lik = a[k] * rs;
		a[k] = lik;
		Lt[k * LP + lane] = lik;
		if (k + 1 < LEAF) {
			if (VAR != 2) a[k + 1] = fma(-lik, bcast_lane(lik, k + 1), a[k + 1]);
			if (VAR != 1) {
				int c = (VAR == 2) ? k + 1 : k + 2;
				if (c < LEAF) {
					if (c & 1) { a[c] = fma(-lik, Lt[k * LP + c], a[c]); c++; }
#pragma unroll
					for (; c + 1 < LEAF; c += 2) {
						const d2_t v = *reinterpret_cast<const d2_t *>(&Lt[k * LP + c]);
						a[c] = fma(-lik, v[0], a[c]); a[c + 1] = fma(-lik, v[1], a[c + 1]);
					}
				}
			}
		}
	}
	long long t2 = __builtin_amdgcn_s_memtime();
	double *wp = T + (long)lane * ld;
#pragma unroll
	for (int k = 0; k < LEAF; k++) if (k <= lane) wp[k] = a[k];
	long long t3 = __builtin_amdgcn_s_memtime();
	if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; }
}
int main() {
	const int n = 64; long ld = 8192;
	std::vector<double> h((size_t)n * ld, 0.0);
	for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) h[i * ld + j] = (i == j) ? 64.0 + i : 1.0 / (1 + abs(i - j));
	double *d; long long *c; hipMalloc(&d, h.size() * 8); hipMalloc(&c, 64);
	long long hc[3];
	for (int var = 0; var < 3; var++) {
		for (int rep = 0; rep < 3; rep++) {
			hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
			hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
			hipEventRecord(e0, 0);
			if (var == 0) hipLaunchKernelGGL(k_factor<0>, dim3(1), dim3(64), 0, 0, d, ld, c);
			if (var == 1) hipLaunchKernelGGL(k_factor<1>, dim3(1), dim3(64), 0, 0, d, ld, c);
			if (var == 2) hipLaunchKernelGGL(k_factor<2>, dim3(1), dim3(64), 0, 0, d, ld, c);
			hipEventRecord(e1, 0); hipDeviceSynchronize();
			float ms; hipEventElapsedTime(&ms, e0, e1);
			hipMemcpy(hc, c, 24, hipMemcpyDeviceToHost);
			printf("var %d rep %d: event %.2f us; cycles(100MHz ticks?) load %lld loop %lld store %lld\n", var, rep, ms * 1e3, hc[0], hc[1], hc[2]);
		}
	}
	return 0;
}
