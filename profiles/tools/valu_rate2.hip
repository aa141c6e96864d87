// VALU issue cost per instruction class on gfx950, second measurement (round 4): the kernel
// floors of profiles/summarize.py price selects / compares / min / max at 4.3 clocks per wave64 instruction, the
// guide (MI355X_MICROARCH.md, wave scheduling) says 2 - the two models disagree by 1.6x on a third of the
// instructions of shade_pixels.  What this tool adds to valu_rate.hip:
//   * clocks are counted by the shader itself (s_memtime = shader clock, s_memrealtime = 100 MHz wall clock), so the
//     result does not depend on the clock the driver reports: "real clocks" per instruction and the frequency the
//     shader actually ran at come out of the same kernel
//   * 1 ... 8 waves per SIMD, independent streams (8 chains per wave)
//   * the classes in question: v_cndmask_b32 (mask in an SGPR pair, as the compiler emits it), v_cmp_*_f32 into an
//     SGPR pair, v_min / v_max / v_med3 / v_min3 / v_max3, against v_fma / v_mul / v_add and the transcendental
//   * a compare -> select pair on the same data (the pattern of GLSL's ?: on floats)
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rate2 profiles/tools/valu_rate2.hip && /tmp/valu_rate2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define REPEAT8(s) s(0) s(1) s(2) s(3) s(4) s(5) s(6) s(7)
#define BODY8(s) REPEAT8(s) REPEAT8(s) REPEAT8(s) REPEAT8(s) REPEAT8(s) REPEAT8(s) REPEAT8(s) REPEAT8(s)

template <int OP>
__global__ void __launch_bounds__(256) k_rate(unsigned long long* clocks, int trips, float seed) {
	float a[8], b = seed * 1.0001f, c = seed * 0.5f;
	unsigned long long masks[8];
	for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; masks[i] = __ballot((threadIdx.x + i) & 2); }
	unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
	for (int t = 0; t < trips; ++t) {
#define S_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define S_MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define S_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define S_MIN(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define S_MAX(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define S_MED3(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define S_MIN3(i) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define S_MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define S_CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(masks[i]));
#define S_CMP(i) asm volatile("v_cmp_le_f32 %0, %1, %2" : "=s"(masks[i]) : "v"(a[i]), "v"(b));
#define S_CMPSEL(i) asm volatile("v_cmp_le_f32 %1, %0, %2\n v_cndmask_b32 %0, %0, %3, %1" : "+v"(a[i]), "=&s"(masks[i]) : "v"(b), "v"(c));
#define S_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define S_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define S_AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define S_MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
#define S_FIXUP(i) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define S_CVT(i) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(a[i]));
		if (OP == 0) { BODY8(S_FMA) }
		if (OP == 1) { BODY8(S_MUL) }
		if (OP == 2) { BODY8(S_ADD) }
		if (OP == 3) { BODY8(S_MIN) }
		if (OP == 4) { BODY8(S_MAX) }
		if (OP == 5) { BODY8(S_MED3) }
		if (OP == 6) { BODY8(S_MIN3) }
		if (OP == 7) { BODY8(S_MAX3) }
		if (OP == 8) { BODY8(S_CND) }
		if (OP == 9) { BODY8(S_CMP) }
		if (OP == 10) { BODY8(S_CMPSEL) }
		if (OP == 11) { BODY8(S_RCP) }
		if (OP == 12) { BODY8(S_PERM) }
		if (OP == 13) { BODY8(S_AND) }
		if (OP == 14) { BODY8(S_MOV) }
		if (OP == 15) { BODY8(S_FIXUP) }
		if (OP == 16) { BODY8(S_CVT) }
	}
	unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
	float s = 0.0f;
	for (int i = 0; i < 8; ++i) s += a[i] + (float) (masks[i] & 1);
	if (s == 12345.678f) clocks[4] = (unsigned long long) s;
	// the slowest wave of the grid bounds the kernel: keep the largest span
	if ((threadIdx.x & 63) == 0) {
		atomicMax(clocks + 0, t1 - t0);
		atomicMax(clocks + 1, w1 - w0);
	}
}

template <int OP>
static void run(const char* name, unsigned long long* device_clocks, int waves_per_simd, int instructions_per_statement) {
	const int trips = 2048;
	int blocks = 256 * waves_per_simd;  // one 256-thread workgroup puts one wave on each SIMD of a CU
	k_rate<OP><<<blocks, 256>>>(device_clocks, 16, 1.5f);
	CHECK(hipDeviceSynchronize());
	double best_clocks = 1e30, best_wall = 1e30, best_ms = 1e30;
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	for (int r = 0; r < 3; ++r) {
		CHECK(hipMemset(device_clocks, 0, 64));
		CHECK(hipEventRecord(e0));
		k_rate<OP><<<blocks, 256>>>(device_clocks, trips, 1.5f);
		CHECK(hipEventRecord(e1));
		unsigned long long host[2];
		CHECK(hipMemcpy(host, device_clocks, sizeof(host), hipMemcpyDeviceToHost));
		float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
		if ((double) host[0] < best_clocks) { best_clocks = (double) host[0]; best_wall = (double) host[1]; }
		if (ms < best_ms) best_ms = ms;
	}
	CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
	// per SIMD: waves_per_simd waves x trips x 64 statements
	double instructions = (double) trips * 64.0 * waves_per_simd * instructions_per_statement;
	double mhz = best_clocks / (best_wall / 100.0);  // wall clock ticks at 100 MHz
	// (the span of the slowest wave says what an instruction costs only while all the waves of its SIMD run side by side;
	// the whole kernel between two events does not depend on that: if the two disagree, the waves did not all run together)
	printf("%-28s %d waves/SIMD: %6.2f shader clocks per wave64 instruction (shader clock %.0f MHz, %.2f ns per instruction; whole kernel by events: %.2f ns per instruction)\n", name, waves_per_simd,
		best_clocks / instructions, mhz, best_wall * 10.0 / instructions, best_ms * 1.0e6 / instructions);
}

int main() {
	unsigned long long* device_clocks;
	CHECK(hipMalloc(&device_clocks, 64));
	const int occupancies[] = {1, 2, 3, 4, 6, 8};
	for (int w : occupancies) {
		run<0>("v_fma_f32", device_clocks, w, 1);
		run<1>("v_mul_f32", device_clocks, w, 1);
		run<2>("v_add_f32", device_clocks, w, 1);
		run<3>("v_min_f32", device_clocks, w, 1);
		run<4>("v_max_f32", device_clocks, w, 1);
		run<5>("v_med3_f32", device_clocks, w, 1);
		run<6>("v_min3_f32", device_clocks, w, 1);
		run<7>("v_max3_f32", device_clocks, w, 1);
		run<8>("v_cndmask_b32 (sgpr pair)", device_clocks, w, 1);
		run<9>("v_cmp_le_f32 -> sgpr pair", device_clocks, w, 1);
		run<10>("v_cmp + v_cndmask (pair)", device_clocks, w, 2);
		run<11>("v_rcp_f32", device_clocks, w, 1);
		run<12>("v_perm_b32", device_clocks, w, 1);
		run<13>("v_and_b32", device_clocks, w, 1);
		run<14>("v_mov_b32", device_clocks, w, 1);
		run<15>("v_div_fixup_f32", device_clocks, w, 1);
		run<16>("v_cvt_f32_ubyte1", device_clocks, w, 1);
	}
	return 0;
}
