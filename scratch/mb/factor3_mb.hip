#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
typedef double d2_t __attribute__((ext_vector_type(2)));
typedef double d4_t __attribute__((ext_vector_type(4)));
constexpr int LEAF=64, LP=66;
__device__ __forceinline__ double bcast_lane(double v, int srclane)
{
	int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
	int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
	return __hiloint2double(hi, lo);
}
// one 16-column panel (columns 16P..16P+15) of the 64x64 block in LDS, rows 16P..63, row per lane (wave 0)
template <int P>
__device__ __forceinline__ void panel_factor(double *A, int lane, int &bad)
{
	double a[16];
#pragma unroll
	for (int c = 0; c < 16; c += 2) {
		d2_t v = *reinterpret_cast<const d2_t *>(&A[lane * LP + 16 * P + c]);
		a[c] = v[0]; a[c + 1] = v[1];
	}
#pragma unroll
	for (int k = 0; k < 16; k++) {
		const double p = bcast_lane(a[k], 16 * P + k);
		if (!(p > 0.0) && bad == 0) bad = 16 * P + k + 1;
		double rs = __builtin_amdgcn_rsq(p);
		{ double t = p * rs; double e = fma(-t, rs, 1.0); rs = fma(rs * 0.5, e, rs); if (NEWTON2) { t = p * rs; e = fma(-t, rs, 1.0); rs = fma(rs * 0.5, e, rs); } }
		const double lik = (lane == 16 * P + k) ? p * rs : a[k] * rs;
		a[k] = lik;
#pragma unroll
		for (int c = k + 1; c < 16; c++)
			a[c] = fma(-lik, bcast_lane(lik, 16 * P + c), a[c]);
	}
	if (lane >= 16 * P) {
#pragma unroll
		for (int c = 0; c < 16; c += 2) {
			d2_t v = {a[c], a[c + 1]};
			*reinterpret_cast<d2_t *>(&A[lane * LP + 16 * P + c]) = v;
		}
	}
}
// trailing update after panel P: tiles (ti,tj), P < tj <= ti <= 3, C -= Pan_ti Pan_tj^T on the MFMA
template <int P>
__device__ __forceinline__ void panel_update(double *A, int wave, int lane)
{
	const int g = lane >> 4, q = lane & 15;
	int t = 0;
#pragma unroll
	for (int ti = P + 1; ti < 4; ti++)
#pragma unroll
		for (int tj = P + 1; tj <= ti; tj++) {
			if ((t & 3) == wave) {
				d4_t c;
#pragma unroll
				for (int r = 0; r < 4; r++) c[r] = A[(16 * ti + g + 4 * r) * LP + 16 * tj + q];
#pragma unroll
				for (int r = 0; r < 4; r++) {
					const double a = -A[(16 * ti + q) * LP + 16 * P + g + 4 * r];
					const double b = A[(16 * tj + q) * LP + 16 * P + g + 4 * r];
					c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
				}
#pragma unroll
				for (int r = 0; r < 4; r++) A[(16 * ti + g + 4 * r) * LP + 16 * tj + q] = c[r];
			}
			t++;
		}
}
__global__ __launch_bounds__(256) void k_factor(double *T, long ld, int c0, int *info, long long* cyc)
{
	__shared__ double A[LEAF * LP];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	long long t0 = __builtin_amdgcn_s_memtime();
	double *D = T + (long)c0 * ld + c0;
	{
		double v[16];
#pragma unroll
		for (int u = 0; u < 16; u++) v[u] = D[(long)(wave + 4 * u) * ld + lane];
#pragma unroll
		for (int u = 0; u < 16; u++) A[(wave + 4 * u) * LP + lane] = v[u];
	}
	__syncthreads();
	long long t1 = __builtin_amdgcn_s_memtime();
	int bad = 0;
	if (wave == 0) panel_factor<0>(A, lane, bad);
	__syncthreads();
	panel_update<0>(A, wave, lane);
	__syncthreads();
	if (wave == 0) panel_factor<1>(A, lane, bad);
	__syncthreads();
	panel_update<1>(A, wave, lane);
	__syncthreads();
	if (wave == 0) panel_factor<2>(A, lane, bad);
	__syncthreads();
	panel_update<2>(A, wave, lane);
	__syncthreads();
	if (wave == 0) panel_factor<3>(A, lane, bad);
	__syncthreads();
	long long t2 = __builtin_amdgcn_s_memtime();
	if (tid == 0 && bad) atomicMin(info, c0 + bad);
#pragma unroll
	for (int u = 0; u < 16; u++) {
		const int r = wave + 4 * u;
		if (lane <= r) D[(long)r * ld + lane] = A[r * LP + lane];
	}
	long long t3 = __builtin_amdgcn_s_memtime();
	if (tid == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; }
}
int main() {
	const int n = 64; long ld = 8192;
	std::vector<double> h((size_t)n * ld, 0.0), ref(n * n);
	for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) h[i * ld + j] = (i == j) ? 64.0 + i : 1.0 / (1 + abs(i - j));
	// reference cholesky
	for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) ref[i * n + j] = h[i * ld + j];
	for (int j = 0; j < n; j++) { double s = ref[j * n + j]; for (int k = 0; k < j; k++) s -= ref[j * n + k] * ref[j * n + k]; ref[j * n + j] = sqrt(s);
		for (int i = j + 1; i < n; i++) { double t = ref[i * n + j]; for (int k = 0; k < j; k++) t -= ref[i * n + k] * ref[j * n + k]; ref[i * n + j] = t / ref[j * n + j]; } }
	double *d; long long *c; int *info; hipMalloc(&d, h.size() * 8); hipMalloc(&c, 64); hipMalloc(&info, 4);
	long long hc[3];
	for (int rep = 0; rep < 4; rep++) {
		hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
		hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
		hipEventRecord(e0, 0);
		hipLaunchKernelGGL(k_factor, dim3(1), dim3(256), 0, 0, d, ld, 0, info, c);
		hipEventRecord(e1, 0); hipDeviceSynchronize();
		float ms; hipEventElapsedTime(&ms, e0, e1);
		hipMemcpy(hc, c, 24, hipMemcpyDeviceToHost);
		printf("rep %d: event %.2f us; cycles load %lld factor %lld store %lld\n", rep, ms * 1e3, hc[0], hc[1], hc[2]);
	}
	std::vector<double> out((size_t)n * ld); hipMemcpy(out.data(), d, out.size() * 8, hipMemcpyDeviceToHost);
	double err = 0; for (int i = 0; i < n; i++) for (int j = 0; j <= i; j++) err = fmax(err, fabs(out[i * ld + j] - ref[i * n + j]));
	printf("max err %.3e\n", err);
	return 0;
}
