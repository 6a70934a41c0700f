// What does a grid-wide barrier cost on MI355X?  hipLaunchCooperativeKernel (the runtime guarantees that every workgroup is resident;
// the launch fails otherwise) + cooperative_groups::grid_group::sync(), `iters` barriers per launch with a little work between them.
// hipcc --offload-arch=gfx950 -O2 grid_sync.hip -o grid_sync
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;
__global__ __launch_bounds__(256) void k(double *x, int iters)
{
	cg::grid_group g = cg::this_grid();
	const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
	double v = x[i];
	for (int it = 0; it < iters; it++) {
		v = v * 1.0000001 + 1e-9;
		x[i] = v;
		g.sync();
		v += x[(i + 257) % ((long)gridDim.x * blockDim.x)] * 1e-12;     // a value another workgroup wrote before the barrier
	}
	x[i] = v;
}
int main()
{
	int dev = 0, coop = 0;
	hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev);
	printf("cooperative launch supported: %d\n", coop);
	double *x;
	hipMalloc(&x, 2048L * 256 * 8);
	hipMemset(x, 0, 2048L * 256 * 8);
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	for (int nwg : {64, 128, 256, 512, 1024}) {
		for (int iters : {1, 101}) {
			void *args[] = {&x, &iters};
			hipError_t e = hipLaunchCooperativeKernel((void *)k, dim3(nwg), dim3(256), args, 0, 0);   // warm-up
			if (e != hipSuccess) { printf("%d workgroups: launch refused (%s)\n", nwg, hipGetErrorString(e)); break; }
			hipDeviceSynchronize();
			hipEventRecord(e0);
			for (int r = 0; r < 10; r++) hipLaunchCooperativeKernel((void *)k, dim3(nwg), dim3(256), args, 0, 0);
			hipEventRecord(e1);
			hipEventSynchronize(e1);
			float ms = 0;
			hipEventElapsedTime(&ms, e0, e1);
			static float base = 0;
			if (iters == 1) base = ms / 10;
			else printf("%4d workgroups x 256 threads: launch with 1 barrier %.1f us; each further barrier %.2f us\n", nwg, base * 1e3, (ms / 10 - base) * 1e3 / 100);
		}
	}
	return 0;
}
