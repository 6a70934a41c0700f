// cost of a kernel boundary between dependent launches: 1000 tiny kernels (each reads what the previous one wrote) as plain
// stream launches and as one replayed hipGraph; 64 and 1024 workgroups per launch.  hipcc --offload-arch=gfx950 -O2 kernel_boundary.hip -o kernel_boundary
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void step(double *x, long n)
{
	const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
	x[i] = x[(i + 257) % n] * 1.0000001 + 1e-9;
}
int main()
{
	double *x; hipMalloc(&x, 1024L * 256 * 8); hipMemset(x, 0, 1024L * 256 * 8);
	hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const int K = 1000;
	for (int nwg : {64, 1024}) {
		const long n = (long)nwg * 256;
		for (int i = 0; i < 10; i++) step<<<nwg, 256, 0, s>>>(x, n);
		hipStreamSynchronize(s);
		hipEventRecord(e0, s);
		for (int i = 0; i < K; i++) step<<<nwg, 256, 0, s>>>(x, n);
		hipEventRecord(e1, s); hipEventSynchronize(e1);
		float ms = 0; hipEventElapsedTime(&ms, e0, e1);
		printf("%4d workgroups: %d dependent stream launches: %.2f us each\n", nwg, K, ms * 1e3 / K);
		hipGraph_t g; hipGraphExec_t ge;
		hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
		for (int i = 0; i < K; i++) step<<<nwg, 256, 0, s>>>(x, n);
		hipStreamEndCapture(s, &g);
		hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
		hipGraphLaunch(ge, s); hipStreamSynchronize(s);
		hipEventRecord(e0, s);
		hipGraphLaunch(ge, s);
		hipEventRecord(e1, s); hipEventSynchronize(e1);
		hipEventElapsedTime(&ms, e0, e1);
		printf("%4d workgroups: the same %d launches replayed from a hipGraph: %.2f us each\n", nwg, K, ms * 1e3 / K);
		hipGraphExecDestroy(ge); hipGraphDestroy(g);
	}
	return 0;
}
