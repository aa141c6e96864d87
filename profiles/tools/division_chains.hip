// Which FMA chains behind v_rcp_f32 / v_rsq_f32 deliver the correctly rounded quotient / root on gfx950?
// Answered by trying every pair of significands: a quotient's rounding depends on the 2^23 x 2^23 pairs of
// significands only (powers of two scale every intermediate value exactly while nothing leaves the normal
// range - the window divide() in csrc/device_math.h is specified for), and 7e13 pairs at some twenty
// instructions each are half a minute on an MI355X.  The reference is the compiler's own expansion of
// a / b (v_div_scale, v_div_fmas, v_div_fixup: IEEE); `chain A`, the compiler's chain without the rescaling
// (what divide() was until this search), must come out with zero mismatches as a check of the search itself,
// and chain B shows that the search can tell chains apart.  Chain C is what divide() does since.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o profiles/tools/division_chains.bin profiles/tools/division_chains.hip
//   profiles/tools/division_chains.bin [log2 of the number of divisors to try, default 23 = all]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kChains = 3;
struct findings {
	unsigned long long mismatches[kChains];
	unsigned first_a[kChains], first_b[kChains];
	unsigned long long root_mismatches[3];
	unsigned first_root[3];
	unsigned long long scale_mismatches[3];
	unsigned first_scale[3];
};

__device__ __forceinline__ void note(findings* out, int chain, unsigned long long count, unsigned a, unsigned b) {
	if (!count) return;
	if (atomicAdd(&out->mismatches[chain], count) == 0ull) { out->first_a[chain] = a; out->first_b[chain] = b; }
}

// thread = one divisor significand, loop = every dividend significand
__global__ void __launch_bounds__(256) k_quotients(unsigned first_divisor, findings* out) {
	const unsigned b_bits = 0x3F800000u | (first_divisor + blockIdx.x * 256u + threadIdx.x);
	const float b = __uint_as_float(b_bits);
	const float r0 = __builtin_amdgcn_rcpf(b);
	const float r1 = fmaf(fmaf(-b, r0, 1.0f), r0, r0);
	unsigned long long bad[kChains] = {0ull, 0ull, 0ull};
	unsigned bad_a[kChains] = {0u, 0u, 0u};
	for (unsigned m = 0; m < (1u << 23); ++m) {
		const unsigned a_bits = 0x3F800000u | m;
		const float a = __uint_as_float(a_bits);
		const float exact = __fdiv_rn(a, b);
		// A: estimate refined once, quotient, two corrections with exact residuals
		float qa = a * r1;
		qa = fmaf(fmaf(-b, qa, a), r1, qa);
		qa = fmaf(fmaf(-b, qa, a), r1, qa);
		// B: the raw estimate, two corrections (two FMAs fewer)
		float qb = a * r0;
		qb = fmaf(fmaf(-b, qb, a), r0, qb);
		qb = fmaf(fmaf(-b, qb, a), r0, qb);
		// C: the refined estimate, one correction (two FMAs fewer): divide()
		float qc = a * r1;
		qc = fmaf(fmaf(-b, qc, a), r1, qc);
		if (qa != exact) { if (!bad[0]) bad_a[0] = a_bits; ++bad[0]; }
		if (qb != exact) { if (!bad[1]) bad_a[1] = a_bits; ++bad[1]; }
		if (qc != exact) { if (!bad[2]) bad_a[2] = a_bits; ++bad[2]; }
	}
	for (int c = 0; c != kChains; ++c) note(out, c, bad[c], bad_a[c], b_bits);
}

// correctly rounded square root by selection between the 1-ulp hardware result and its two neighbours
// (csrc/device_math.h square_root_unguarded(); pinned against the compiler's sqrtf for every float of 200 binades)
__device__ __forceinline__ float root_by_neighbours(float x) {
	float s = __builtin_amdgcn_sqrtf(x);
	float below = __uint_as_float(__float_as_uint(s) - 1u), above = __uint_as_float(__float_as_uint(s) + 1u);
	float residual_below = fmaf(-below, s, x), residual_above = fmaf(-above, s, x);
	s = (residual_below <= 0.0f) ? below : s;
	s = (residual_above > 0.0f) ? above : s;
	return s;
}

// every significand of two neighbouring binades (the exponent's parity is all that matters)
__global__ void __launch_bounds__(256) k_roots(findings* out) {
	const unsigned x_bits = 0x3F800000u + blockIdx.x * 256u + threadIdx.x;  // [1, 4)
	const float x = __uint_as_float(x_bits);
	const float exact = root_by_neighbours(x);
	// 0: the coupled iteration behind v_rsq_f32 (what the compiler emits for sqrtf when denormals are flushed)
	float r = __builtin_amdgcn_rsqf(x);
	float g = x * r, h = 0.5f * r;
	float e = fmaf(-h, g, 0.5f);
	h = fmaf(h, e, h);
	g = fmaf(g, e, g);
	float d = fmaf(-g, g, x);
	g = fmaf(d, h, g);
	// 1: v_sqrt_f32 and one correction with half the reciprocal of the estimate from v_rsq_f32
	float s = __builtin_amdgcn_sqrtf(x);
	float half_reciprocal = 0.5f * __builtin_amdgcn_rsqf(x);
	float t = fmaf(fmaf(-s, s, x), half_reciprocal, s);
	// 2: inversesqrt = 1 / sqrt by the one-correction chain with v_rsq_f32 as the reciprocal estimate, against
	// the compiler's division of 1 by the exact root
	float y = __builtin_amdgcn_rsqf(x);
	float q = fmaf(fmaf(-exact, y, 1.0f), y, y);
	q = fmaf(fmaf(-exact, q, 1.0f), q, q);
	if (q != __fdiv_rn(1.0f, exact) && atomicAdd(&out->root_mismatches[2], 1ull) == 0ull) out->first_root[2] = x_bits;
	if (g != exact && atomicAdd(&out->root_mismatches[0], 1ull) == 0ull) out->first_root[0] = x_bits;
	if (t != exact && atomicAdd(&out->root_mismatches[1], 1ull) == 0ull) out->first_root[1] = x_bits;
}

// The premise of searching significands only: the estimates depend on the significand (and, for the
// roots, the exponent's parity) alone - v_rcp_f32(m 2^e) = v_rcp_f32(m) 2^-e and so on - for every float
// whose estimate is a normal number.  Checked for all of them.
__global__ void __launch_bounds__(256) k_scale_invariance(findings* out) {
	for (unsigned long long i = (unsigned long long) blockIdx.x * 256u + threadIdx.x; i < (1ull << 31); i += (unsigned long long) gridDim.x * 256u) {
		const unsigned bits = (unsigned) i;
		const int exponent = (int) (bits >> 23) - 127;
		if (exponent < -125 || exponent > 125) continue;
		const float x = __uint_as_float(bits);
		const float m = __uint_as_float((bits & 0x7FFFFFu) | 0x3F800000u);                                 // [1, 2)
		const float m2 = __uint_as_float((bits & 0x7FFFFFu) | ((exponent & 1) ? 0x40000000u : 0x3F800000u));  // [1, 4), same parity
		const int half = (exponent - (exponent & 1)) / 2;
		// (ldexpf of a normal number to a normal number is exact)
		if (__builtin_amdgcn_rcpf(x) != ldexpf(__builtin_amdgcn_rcpf(m), -exponent) && atomicAdd(&out->scale_mismatches[0], 1ull) == 0ull) out->first_scale[0] = bits;
		if (__builtin_amdgcn_sqrtf(x) != ldexpf(__builtin_amdgcn_sqrtf(m2), half) && atomicAdd(&out->scale_mismatches[1], 1ull) == 0ull) out->first_scale[1] = bits;
		if (__builtin_amdgcn_rsqf(x) != ldexpf(__builtin_amdgcn_rsqf(m2), -half) && atomicAdd(&out->scale_mismatches[2], 1ull) == 0ull) out->first_scale[2] = bits;
	}
}

int main(int argc, char** argv) {
	int log2_divisors = argc > 1 ? atoi(argv[1]) : 23;
	if (log2_divisors < 8) log2_divisors = 8;
	if (log2_divisors > 23) log2_divisors = 23;
	findings* device = NULL;
	CHECK(hipMalloc(&device, sizeof(findings)));
	CHECK(hipMemset(device, 0, sizeof(findings)));
	k_roots<<<(1u << 24) / 256u, 256>>>(device);
	CHECK(hipDeviceSynchronize());
	k_scale_invariance<<<8192, 256>>>(device);
	CHECK(hipDeviceSynchronize());
	// divisors: all of them, or evenly spread ones; 2^17 per launch (well below a second each)
	const unsigned divisors = 1u << log2_divisors, per_launch = divisors < (1u << 17) ? divisors : (1u << 17);
	hipEvent_t start, stop;
	CHECK(hipEventCreate(&start)); CHECK(hipEventCreate(&stop));
	CHECK(hipEventRecord(start, 0));
	for (unsigned first = 0; first < divisors; first += per_launch) {
		// (a partial search takes a contiguous block of divisors from every 1 / launches-th of the range)
		unsigned offset = log2_divisors == 23 ? first : (unsigned) (((unsigned long long) first << 23) >> log2_divisors);
		k_quotients<<<per_launch / 256u, 256>>>(offset, device);
		CHECK(hipDeviceSynchronize());
	}
	CHECK(hipEventRecord(stop, 0));
	CHECK(hipEventSynchronize(stop));
	float ms = 0.0f;
	CHECK(hipEventElapsedTime(&ms, start, stop));
	findings host;
	CHECK(hipMemcpy(&host, device, sizeof(host), hipMemcpyDeviceToHost));
	const char* names[kChains] = {"A: rcp refined once, quotient, two corrections", "B: raw rcp, quotient, two corrections", "C: rcp refined once, quotient, one correction (divide())"};
	printf("quotients: 2^%d divisor significands x 2^23 dividend significands in %.1f s\n", log2_divisors, ms * 1.0e-3);
	for (int c = 0; c != kChains; ++c) {
		printf("  %-62s %llu mismatches", names[c], host.mismatches[c]);
		if (host.mismatches[c]) printf(" (e.g. 0x%08x / 0x%08x)", host.first_a[c], host.first_b[c]);
		printf("\n");
	}
	const char* root_names[3] = {"v_rsq_f32, coupled iteration (5 FMAs, 2 products)", "v_sqrt_f32 + one correction by 0.5 v_rsq_f32", "1 / sqrt: v_rsq_f32 refined once, one correction"};
	printf("square roots: 2^24 arguments in [1, 4)\n");
	for (int c = 0; c != 3; ++c) {
		printf("  %-62s %llu mismatches", root_names[c], host.root_mismatches[c]);
		if (host.root_mismatches[c]) printf(" (e.g. 0x%08x)", host.first_root[c]);
		printf("\n");
	}
	const char* scale_names[3] = {"v_rcp_f32", "v_sqrt_f32", "v_rsq_f32"};
	printf("estimates of m 2^e against those of m, every positive float with an exponent in [-125, 125]\n");
	for (int c = 0; c != 3; ++c) {
		printf("  %-62s %llu mismatches", scale_names[c], host.scale_mismatches[c]);
		if (host.scale_mismatches[c]) printf(" (e.g. 0x%08x)", host.first_scale[c]);
		printf("\n");
	}
	return 0;
}
