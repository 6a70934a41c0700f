// A hand-written grid barrier (one atomic arrival per workgroup, a generation word everybody polls) beside the runtime's grid.sync():
// launched cooperatively, so every workgroup is resident and the barrier cannot dead-lock; a spin gives up after 2 ms (exit condition
// every wave reaches).  hipcc --offload-arch=gfx950 -O2 flag_barrier.hip -o flag_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(double *x, unsigned *bar, int iters, int *gave_up)
{
	const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
	double v = x[i];
	for (int it = 0; it < iters; it++) {
		v = v * 1.0000001 + 1e-9;
		x[i] = v;
		__threadfence();                     // this workgroup's stores are visible device-wide before it arrives
		__syncthreads();
		if (threadIdx.x == 0) {
			const unsigned gen = __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (__hip_atomic_fetch_add(bar, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
				__hip_atomic_store(bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_fetch_add(bar + 1, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
			} else {
				const unsigned long long t0 = wall_clock64();
				while (__hip_atomic_load(bar + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == gen) {
					__builtin_amdgcn_s_sleep(2);
					if (wall_clock64() - t0 > 200000ull) { *gave_up = 1; break; }     // 2 ms
				}
			}
		}
		__syncthreads();
		v += __builtin_nontemporal_load(&x[(i + 257) % ((long)gridDim.x * blockDim.x)]) * 1e-12;
	}
	x[i] = v;
}
int main()
{
	double *x; unsigned *bar; int *gu;
	hipMalloc(&x, 2048L * 256 * 8); hipMemset(x, 0, 2048L * 256 * 8);
	hipMalloc(&bar, 64); hipMemset(bar, 0, 64);
	hipMalloc(&gu, 4); hipMemset(gu, 0, 4);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	for (int nwg : {64, 128, 256, 512, 1024}) {
		float base = 0;
		for (int iters : {1, 101}) {
			void *args[] = {&x, &bar, &iters, &gu};
			if (hipLaunchCooperativeKernel((void *)k, dim3(nwg), dim3(256), args, 0, 0) != hipSuccess) { printf("%d: launch refused\n", nwg); return 0; }
			hipDeviceSynchronize();
			hipEventRecord(e0);
			for (int r = 0; r < 10; r++) hipLaunchCooperativeKernel((void *)k, dim3(nwg), dim3(256), args, 0, 0);
			hipEventRecord(e1); hipEventSynchronize(e1);
			float ms = 0; hipEventElapsedTime(&ms, e0, e1);
			if (iters == 1) base = ms / 10;
			else printf("%4d workgroups x 256 threads: each further flag barrier %.2f us\n", nwg, (ms / 10 - base) * 1e3 / 100);
		}
	}
	int h = 0; hipMemcpy(&h, gu, 4, hipMemcpyDeviceToHost);
	printf("a spin gave up: %d\n", h);
	return 0;
}
