// Issue rate of the VALU instructions that the shading pass is made of, measured on the GPU it runs
// on (gfx950): how many clocks one wave64 instruction occupies its SIMD for.  The kernels of the pass
// are bound by VALU issue (profiles/*_summary.md), so these numbers are the price list.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rate profiles/tools/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float float2v __attribute__((ext_vector_type(2)));

// 8 independent chains so that dependent-issue latency never shows, 64 instructions per loop trip
#define REPEAT8(s) s(0) s(1) s(2) s(3) s(4) s(5) s(6) s(7)
#define BODY8(s) REPEAT8(s) REPEAT8(s) REPEAT8(s) REPEAT8(s) REPEAT8(s) REPEAT8(s) REPEAT8(s) REPEAT8(s)

template <int OP>
__global__ void __launch_bounds__(256) k_rate(float* out, int trips, float seed) {
	float a[8], b = seed * 1.0001f, c = seed * 0.5f;
	float2v p[8], pb = {b, b}, pc = {c, c};
	unsigned u[8], sel = (threadIdx.x & 1) * 16u;
	unsigned long long mask = __ballot(threadIdx.x & 2);
	__shared__ float lds[4096];
	lds[threadIdx.x] = seed; lds[threadIdx.x + 256] = seed;
	__syncthreads();
	for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; p[i] = float2v{a[i], a[i] + 1.0f}; u[i] = threadIdx.x * 2654435761u + i; }
	for (int t = 0; t < trips; ++t) {
#define S_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define S_MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define S_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
#define S_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
#define S_CVT(i) asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(a[i]) : "v"(u[i]));
#define S_ALIGN(i) asm volatile("v_alignbit_b32 %0, %0, %0, %1" : "+v"(u[i]) : "v"(sel));
#define S_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(sel), "v"(u[(i + 1) & 7]));
#define S_MIN3(i) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define S_MAX(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define S_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define S_RSQ(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
#define S_SQRT(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
#define S_DIVSCALE(i) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a[i]) : "v"(b) : "vcc");
#define S_DIVFMAS(i) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
#define S_DIVFIXUP(i) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define S_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
#define S_MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
#define S_CMP(i) asm volatile("v_cmp_le_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
#define S_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define S_SUB(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define S_FMAC(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define S_FMAAK(i) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f800003" : "+v"(a[i]) : "v"(b));
#define S_AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(sel));
#define S_LSHL(i) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(u[i]));
#define S_XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(sel));
#define S_ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(sel));
#define S_LSHLOR(i) asm volatile("v_lshl_or_b32 %0, %0, 7, %1" : "+v"(u[i]) : "v"(sel));
#define S_BFE(i) asm volatile("v_bfe_u32 %0, %0, 4, 16" : "+v"(u[i]));
#define S_MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define S_CND64(i) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(mask));
#define S_CVTU(i) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a[i]) : "v"(u[i]));
#define S_CVTB(i) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a[i]) : "v"(u[i]));
#define S_LDSR(i) asm volatile("ds_read_b32 %0, %1" : "=v"(a[i]) : "v"(u[i] & 0x3FFCu));
		if (OP == 0) { BODY8(S_FMA) }
		if (OP == 1) { BODY8(S_PKFMA) }
		if (OP == 2) { BODY8(S_CVT) }
		if (OP == 3) { BODY8(S_ALIGN) }
		if (OP == 4) { BODY8(S_PERM) }
		if (OP == 5) { BODY8(S_MIN3) }
		if (OP == 6) { BODY8(S_RCP) }
		if (OP == 7) { BODY8(S_RSQ) }
		if (OP == 8) { BODY8(S_SQRT) }
		if (OP == 9) { BODY8(S_DIVSCALE) }
		if (OP == 10) { BODY8(S_DIVFMAS) }
		if (OP == 11) { BODY8(S_DIVFIXUP) }
		if (OP == 12) { BODY8(S_CNDMASK) }
		if (OP == 13) { BODY8(S_MOV) }
		if (OP == 14) { BODY8(S_CMP) }
		if (OP == 15) { BODY8(S_MUL) }
		if (OP == 16) { BODY8(S_PKMUL) }
		if (OP == 17) { BODY8(S_MAX) }
		if (OP == 18) { BODY8(S_ADD) }
		if (OP == 19) { BODY8(S_SUB) }
		if (OP == 20) { BODY8(S_FMAC) }
		if (OP == 21) { BODY8(S_FMAAK) }
		if (OP == 22) { BODY8(S_AND) }
		if (OP == 23) { BODY8(S_LSHL) }
		if (OP == 24) { BODY8(S_XOR) }
		if (OP == 25) { BODY8(S_ADDU) }
		if (OP == 26) { BODY8(S_LSHLOR) }
		if (OP == 27) { BODY8(S_BFE) }
		if (OP == 28) { BODY8(S_MAX3) }
		if (OP == 29) { BODY8(S_CND64) }
		if (OP == 30) { BODY8(S_CVTU) }
		if (OP == 31) { BODY8(S_CVTB) }
		if (OP == 32) { BODY8(S_LDSR) }
	}
	float s = 0.0f;
	for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y + (float) u[i];
	if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int OP>
static void run(const char* name, float* out, int waves_per_simd) {
	const int trips = 4096;
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	// 256 CUs x 4 SIMDs; one 256-thread workgroup puts one wave on each SIMD of a CU
	int blocks = 256 * waves_per_simd;
	k_rate<OP><<<blocks, 256>>>(out, 64, 1.5f);
	CHECK(hipDeviceSynchronize());
	float best = 1e30f;
	for (int r = 0; r < 5; ++r) {
		CHECK(hipEventRecord(e0));
		k_rate<OP><<<blocks, 256>>>(out, trips, 1.5f);
		CHECK(hipEventRecord(e1));
		CHECK(hipEventSynchronize(e1));
		float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
		if (ms < best) best = ms;
	}
	int clock_khz = 0;
	CHECK(hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeClockRate, 0));
	double instructions_per_simd = (double) trips * 64.0 * waves_per_simd;
	double clocks = best * 1e-3 * clock_khz * 1e3;
	printf("%-22s %d wave(s)/SIMD: %.3f ms, %.2f clocks per wave64 instruction (at %d MHz)\n", name, waves_per_simd, best, clocks / instructions_per_simd, clock_khz / 1000);
}

int main() {
	float* out;
	CHECK(hipMalloc(&out, 4096));
	for (int w = 2; w <= 4; w *= 2) {
		run<0>("v_fma_f32", out, w);
		run<15>("v_mul_f32", out, w);
		run<1>("v_pk_fma_f32", out, w);
		run<16>("v_pk_mul_f32", out, w);
		run<2>("v_cvt_f32_u32_sdwa", out, w);
		run<3>("v_alignbit_b32", out, w);
		run<4>("v_perm_b32", out, w);
		run<5>("v_min3_f32", out, w);
		run<17>("v_max_f32", out, w);
		run<6>("v_rcp_f32", out, w);
		run<7>("v_rsq_f32", out, w);
		run<8>("v_sqrt_f32", out, w);
		run<9>("v_div_scale_f32", out, w);
		run<10>("v_div_fmas_f32", out, w);
		run<11>("v_div_fixup_f32", out, w);
		run<12>("v_cndmask_b32", out, w);
		run<13>("v_mov_b32", out, w);
		run<14>("v_cmp_le_f32", out, w);
		run<18>("v_add_f32", out, w);
		run<19>("v_sub_f32", out, w);
		run<20>("v_fmac_f32", out, w);
		run<21>("v_fmaak_f32", out, w);
		run<22>("v_and_b32", out, w);
		run<23>("v_lshlrev_b32", out, w);
		run<24>("v_xor_b32", out, w);
		run<25>("v_add_u32", out, w);
		run<26>("v_lshl_or_b32", out, w);
		run<27>("v_bfe_u32", out, w);
		run<28>("v_max3_f32", out, w);
		run<29>("v_cndmask_b32 (sgpr mask)", out, w);
		run<30>("v_cvt_f32_u32", out, w);
		run<31>("v_cvt_f32_ubyte1", out, w);
		run<32>("ds_read_b32", out, w);
	}
	return 0;
}
