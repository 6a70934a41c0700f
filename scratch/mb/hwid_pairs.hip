// Which HW_ID field tells the two workgroups that share a CU apart?  512 threads, 64 KB LDS (two workgroups per CU, as
// the 128x128 GEMM), every workgroup records HW_ID / XCC_ID of its wave 0 and its start time, then sleeps 50 us so that a
// first round of 512 workgroups is co-resident.   hipcc --offload-arch=gfx950 -O2 hwid_pairs.hip -o hwid_pairs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(512, 4) void probe(unsigned *out, unsigned long long *t)
{
	__shared__ double pad[8192];
	pad[threadIdx.x] = threadIdx.x;
	__syncthreads();
	if (threadIdx.x == 0) {
		unsigned hw, xcc;
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
		out[2 * blockIdx.x] = hw;
		out[2 * blockIdx.x + 1] = xcc;
		t[blockIdx.x] = wall_clock64();
	}
	const unsigned long long t0 = wall_clock64();
	while (wall_clock64() - t0 < 5000) __builtin_amdgcn_s_sleep(32);
	if (pad[(threadIdx.x * 7) & 8191] < 0) out[0] = 0;
}
int main()
{
	const int nwg = 1536;
	unsigned *d; unsigned long long *dt;
	hipMalloc(&d, nwg * 8); hipMalloc(&dt, nwg * 8);
	probe<<<nwg, 512>>>(d, dt);
	hipDeviceSynchronize();
	std::vector<unsigned> h(2 * nwg); std::vector<unsigned long long> ht(nwg);
	hipMemcpy(h.data(), d, nwg * 8, hipMemcpyDeviceToHost);
	hipMemcpy(ht.data(), dt, nwg * 8, hipMemcpyDeviceToHost);
	unsigned long long tmin = ~0ull;
	for (auto v : ht) if (v < tmin) tmin = v;
	std::map<unsigned, std::vector<int>> cu;
	for (int i = 0; i < nwg; i++) {
		const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 15;
		const unsigned key = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15);
		cu[key].push_back(i);
	}
	printf("distinct CUs %zu\n", cu.size());
	int shown = 0, tg_differs = 0, wave_differs = 0, pairs = 0;
	for (auto &kv : cu) {
		// first-round pair = the two earliest workgroups of this CU
		std::vector<int> v = kv.second;
		std::sort(v.begin(), v.end(), [&](int a, int b) { return ht[a] < ht[b]; });
		if (v.size() < 2) continue;
		const unsigned a = h[2 * v[0]], b = h[2 * v[1]];
		pairs++;
		tg_differs += (((a >> 16) & 15) & 1) != (((b >> 16) & 15) & 1);
		wave_differs += (((a & 15) >> 1) & 1) != (((b & 15) >> 1) & 1);
		if (shown++ < 12)
			printf("cu %06x: wg %4d (t %6llu tg %u wave %u simd %u) wg %4d (t %6llu tg %u wave %u simd %u)  third: %s\n", kv.first, v[0], ht[v[0]] - tmin,
			       (a >> 16) & 15, a & 15, (a >> 4) & 3, v[1], ht[v[1]] - tmin, (b >> 16) & 15, b & 15, (b >> 4) & 3,
			       v.size() > 2 ? "yes" : "no");
	}
	printf("pairs %d: tg parity differs in %d, wave-slot bit 1 differs in %d\n", pairs, tg_differs, wave_differs);
	// block index difference inside pairs
	std::map<int, int> diff;
	for (auto &kv : cu) { std::vector<int> v = kv.second; std::sort(v.begin(), v.end(), [&](int a, int b) { return ht[a] < ht[b]; }); if (v.size() >= 2) diff[v[1] - v[0]]++; }
	for (auto &d2 : diff) printf("  blockIdx difference %d: %d pairs\n", d2.first, d2.second);
	return 0;
}
