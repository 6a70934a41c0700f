// precision of the hardware v_rsq_f64 / v_rcp_f64 estimates and of one / two Newton refinements
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double *x, double *o0, double *o1, double *o2, double *r0, double *r1, int n)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	double p = x[i];
	double rs = __builtin_amdgcn_rsq(p);
	o0[i] = rs;
	double t = p * rs, e = fma(-t, rs, 1.0); rs = fma(rs * 0.5, e, rs);
	o1[i] = rs;
	t = p * rs; e = fma(-t, rs, 1.0); rs = fma(rs * 0.5, e, rs);
	o2[i] = rs;
	double y = __builtin_amdgcn_rcp(p);
	r0[i] = y;
	y = fma(fma(-p, y, 1.0), y, y);
	r1[i] = y;
}
int main()
{
	const int n = 1 << 20;
	double *hx = new double[n], *h[5];
	for (int i = 0; i < n; i++) hx[i] = ldexp(1.0 + (double)rand() / RAND_MAX, (rand() % 40) - 20);
	double *dx, *d[5];
	hipMalloc(&dx, n * 8); hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
	for (int j = 0; j < 5; j++) { hipMalloc(&d[j], n * 8); h[j] = new double[n]; }
	hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d[0], d[1], d[2], d[3], d[4], n);
	for (int j = 0; j < 5; j++) hipMemcpy(h[j], d[j], n * 8, hipMemcpyDeviceToHost);
	const char *name[5] = {"rsq estimate", "rsq + 1 Newton", "rsq + 2 Newton", "rcp estimate", "rcp + 1 Newton"};
	for (int j = 0; j < 5; j++) {
		double mx = 0;
		for (int i = 0; i < n; i++) {
			long double ref = j < 3 ? 1.0L / sqrtl((long double)hx[i]) : 1.0L / (long double)hx[i];
			double e = fabs((double)(((long double)h[j][i] - ref) / ref));
			if (e > mx) mx = e;
		}
		printf("%-16s max rel err %.3e  (2^%.1f)\n", name[j], mx, log2(mx));
	}
	return 0;
}
