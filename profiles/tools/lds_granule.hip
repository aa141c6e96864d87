// How much LDS does a workgroup really take on gfx950?  (a) asks the runtime how many single-wave workgroups with d bytes
// of dynamic LDS fit a CU; (b) measures it: launches 256 x 40 such workgroups that note when they start and then spin
// for 300 us - the ones that start within the first 150 us are the ones that were resident together.
//   hipcc --offload-arch=gfx950 -O2 profiles/tools/lds_granule.hip -o /tmp/lds_granule && /tmp/lds_granule
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

extern __shared__ float dynamic_lds[];
__global__ void __launch_bounds__(64) probe(unsigned long long* starts, unsigned long long spin_ticks) {
	unsigned long long t0 = wall_clock64();
	dynamic_lds[threadIdx.x] = (float) threadIdx.x;
	if (threadIdx.x == 0) starts[blockIdx.x] = t0;
	while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
	if (dynamic_lds[63 - threadIdx.x] < 0.0f) starts[blockIdx.x] = 0;
}

int main() {
	const int sizes[] = {1024, 8192, 10240, 11264, 12288, 12800, 13056, 13312, 13568, 13824, 14336, 15360, 15361, 15872, 16128, 16384, 16385, 16640, 16720, 16896, 17408, 17920, 18432, 20480};
	int device = 0, cus = 0, rate_khz = 0;
	hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
	hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, device);
	if (rate_khz <= 0) rate_khz = 100000;
	const int blocks = cus * 40;
	unsigned long long* starts = NULL;
	hipMalloc(&starts, sizeof(unsigned long long) * blocks);
	std::vector<unsigned long long> host(blocks);
	const unsigned long long spin = (unsigned long long) rate_khz * 300ull / 1000ull, early = (unsigned long long) rate_khz * 150ull / 1000ull;
	printf("%d CUs, wall clock %d kHz\n", cus, rate_khz);
	for (int bytes : sizes) {
		int predicted = 0;
		hipError_t status = hipOccupancyMaxActiveBlocksPerMultiprocessor(&predicted, probe, 64, (size_t) bytes);
		hipMemset(starts, 0, sizeof(unsigned long long) * blocks);
		hipLaunchKernelGGL(probe, dim3(blocks), dim3(64), (size_t) bytes, 0, starts, spin);
		hipDeviceSynchronize();
		hipMemcpy(host.data(), starts, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
		unsigned long long first = ~0ull;
		for (unsigned long long t : host) if (t && t < first) first = t;
		int together = 0;
		for (unsigned long long t : host) if (t && t - first < early) ++together;
		printf("%6d bytes of dynamic LDS: runtime says %2d workgroups per CU%s, measured %.2f per CU resident together (%d of %d)\n", bytes, predicted, status == hipSuccess ? "" : " (query failed)", (double) together / cus, together, blocks);
	}
	hipFree(starts);
	return 0;
}
