// Device-side arithmetic for the shading kernels (gfx950).
//
// Three arithmetic modes, selected at compile time per translation unit (VKR_MATH_MODE):
//   2  "libm" (the default of the pass): IEEE division and square roots, no contraction
//      (-ffp-contract=off), inversesqrt as 1 / sqrt, and atan / acos / sin / cos / log2 / pow as
//      glibc 2.35 evaluates them (glibc_math.h) - operation for operation what the CPU oracle does
//      in its math mode 0, the mode that is pinned bit for bit against the reference's shader
//      source compiled as C++.  Frames equal that oracle's in every bit.
//   0  "exact": mode 2 with ONE function replaced: the arctangent - a fifth of the libm kernel's
//      time (two divisions, five argument ranges, a degree-11 polynomial without FMAs) - is a
//      polynomial with explicit FMAs behind a single division of the smaller by the larger magnitude.
//      Every operation is still correctly rounded, so the result is reproducible on any IEEE machine
//      (oracle math mode 1 mirrors it bit for bit).  Measured against mode 2 at BASELINE config 3:
//      RMSE 2.3e-7, no pixel beyond 5e-4, the same NaN-guard pixels (profiles/r03j).  (Until
//      round 3 this mode also had a Newton inversesqrt and polynomial acos / sincos / log2: the
//      0.85-ulp inversesqrt is what moved samples across sliver sectors into or out of the shader's
//      NaN guard - 20 pixels of a config-3 frame - for 5 % of the time.)
//   1  "fast": v_rcp_f32 / v_rsq_f32 / v_sqrt_f32 (1 ulp), -ffp-contract=fast, polynomial
//      transcendentals.
// The reference leaves these precisions to the GLSL driver
// (src/shaders/polygon_sampling.glsl:79-82).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef VKR_MATH_MODE
#define VKR_MATH_MODE 0
#endif
#define VKR_FAST_MATH (VKR_MATH_MODE == 1)
// glibc's functions in both IEEE modes ...
#define VKR_LIBM_MATH (VKR_MATH_MODE != 1)
// ... except those named here, which keep their polynomial forms - bit 0: arctangent (that is what
// mode 0 is), bit 1: inversesqrt, bit 2: acos / sincos (profiling aids, profiles/tools/ab_build.sh:
// pricing the functions one by one is how mode 0 got its definition)
#ifndef VKR_LIBM_EXCEPT
#define VKR_LIBM_EXCEPT (VKR_MATH_MODE == 0 ? 1 : 0)
#endif

#define VKR_DEV __device__ __forceinline__

namespace vkr {

constexpr float kPi = 3.1415926535897932384626433832795f;
constexpr float kInvPi = 0.31830988618379067153776752674503f;
constexpr float kHalfPi = 1.5707963267948966192313216916398f;

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

VKR_DEV f2 mk2(float x, float y) { f2 r; r.x = x; r.y = y; return r; }
VKR_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
VKR_DEV f2 operator+(f2 a, f2 b) { return mk2(a.x + b.x, a.y + b.y); }
VKR_DEV f2 operator-(f2 a, f2 b) { return mk2(a.x - b.x, a.y - b.y); }
VKR_DEV f2 operator*(f2 a, float s) { return mk2(a.x * s, a.y * s); }
VKR_DEV f2 operator-(f2 a) { return mk2(-a.x, -a.y); }
VKR_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
VKR_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
VKR_DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
VKR_DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
VKR_DEV f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
// dot products accumulate left to right
VKR_DEV float dot(f2 a, f2 b) { return a.x * b.x + a.y * b.y; }
VKR_DEV float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
VKR_DEV f3 cross(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
VKR_DEV f3 fma3(float s, f3 a, f3 b) { return mk3(fmaf(s, a.x, b.x), fmaf(s, a.y, b.y), fmaf(s, a.z, b.z)); }
VKR_DEV f2 fma2(float s, f2 a, f2 b) { return mk2(fmaf(s, a.x, b.x), fmaf(s, a.y, b.y)); }
VKR_DEV f2 rot90(f2 a) { return mk2(-a.y, a.x); }

// GLSL min/max/clamp as the specification words them
VKR_DEV float gmax(float x, float y) { return (x < y) ? y : x; }
VKR_DEV float gmin(float x, float y) { return (y < x) ? y : x; }
VKR_DEV float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }
VKR_DEV float positive_part(float x) { return (x > 0.0f) ? x : 0.0f; }
// Identity the optimiser cannot see through.  Used where it would otherwise fold
// select(load a[i], load a[j]) into load a[select(i, j)], which forces register
// arrays into scratch memory.
VKR_DEV float opaque(float x) { asm("" : "+v"(x)); return x; }
// The same for an integer, and pinned where it stands (volatile): what is computed from the result cannot be moved out of
// the loop the call stands in.  For loop-invariant comparisons whose results - execution masks, two scalar registers each -
// the optimiser would otherwise keep across a loop that has no scalar registers to spare.
VKR_DEV uint32_t opaque_here(uint32_t x) { asm volatile("" : "+v"(x)); return x; }

// Division.  IEEE modes: the correctly rounded quotient from v_rcp_f32 - the estimate refined by one
// Newton step, the quotient, ONE correction of it with an exact (FMA) residual, v_div_fixup_f32 for
// zeros, infinities and NaNs.  The compiler's expansion of a / b has a second correction and rescales
// operands near the ends of the exponent range (two v_div_scale_f32, v_div_fmas_f32): 40 clocks of
// VALU issue against 25 here (profiles/tools/valu_rate.hip; a fifth of the shading kernel's
// instructions are divisions).  That one correction is enough on this hardware is not a theorem but
// the result of trying every pair of significands, 2^23 x 2^23 quotients in 70 s on an MI355X
// (profiles/tools/division_chains.hip, profiles/r03w/division_chains.txt: no mismatch against the
// compiler's chain; the raw estimate with two corrections fails 26 621 times); powers of two scale every
// intermediate value exactly, so the result is the IEEE quotient whenever v_div_scale_f32 would not
// have rescaled: |b| in [2^-126, 2^126), a = 0 or |a| >= 2^-103, |a / b| in [2^-126, 2^96)
// (tests/test_gpu_arithmetic.py pins that, and a slice of the search through compare_device_division()).
// The operands here are radiances, densities, areas and lengths of O(1e-10 ... 1e10); like square_root below
// this gives up the last decades of the exponent range, nothing else.
// The IEEE quotient over the whole exponent range: the compiler's own expansion with the two
// v_div_scale_f32 and the v_div_fmas_f32 (40 clocks).  For quotients that may leave the window of
// divide() - the tangent of an angle next to pi / 2, n / d with d -> 0.
VKR_DEV float divide_full_range(float a, float b) {
#if VKR_FAST_MATH
	return a * __builtin_amdgcn_rcpf(b);
#else
	return __fdiv_rn(a, b);
#endif
}
// VKR_IEEE_DIVISION_EVERYWHERE=1 (libvkr_shading_ieee.so, `make ieee`): every quotient and every square root is
// the compiler's full-range IEEE expansion.  The check build of the test-suite: the product kernels must give the
// same frames bit for bit (tests/test_gpu_division_window.py), i.e. no operand of the pass leaves the windows that
// divide() and square_root() document.
#ifndef VKR_IEEE_DIVISION_EVERYWHERE
#define VKR_IEEE_DIVISION_EVERYWHERE 0
#endif
VKR_DEV float divide(float a, float b) {
#if VKR_FAST_MATH
	return a * __builtin_amdgcn_rcpf(b);
#elif VKR_IEEE_DIVISION_EVERYWHERE
	return __fdiv_rn(a, b);
#else
	float r = __builtin_amdgcn_rcpf(b);
	r = fmaf(fmaf(-b, r, 1.0f), r, r);
	float q = a * r;
	q = fmaf(fmaf(-b, q, a), r, q);
	return __builtin_amdgcn_div_fixupf(q, b, a);
#endif
}
VKR_DEV float rcp(float x) {
#if VKR_FAST_MATH
	return __builtin_amdgcn_rcpf(x);
#else
	return divide(1.0f, x);
#endif
}
// Correctly rounded square root (== sqrtf of the oracle).  The compiler's expansion is 17
// instructions, 5 of which rescale denormal inputs; arguments here are sums of products
// of O(1) quantities (exactly 0 or far above 1e-38), so this is the same selection
// between s - 1 ulp, s and s + 1 ulp around the 1-ulp hardware result, without the rescaling.
// The selection between the hardware result and its two neighbours.  For +-0 and +inf the
// "neighbours" are NaN bit patterns (or, above +0, the smallest denormal, whose residual is 0), every
// comparison with their residuals is false and the hardware result - the argument itself - stays.
VKR_DEV float square_root_unguarded(float x) {
	float s = __builtin_amdgcn_sqrtf(x);
	float below = __uint_as_float(__float_as_uint(s) - 1u), above = __uint_as_float(__float_as_uint(s) + 1u);
	float residual_below = fmaf(-below, s, x), residual_above = fmaf(-above, s, x);
	s = (residual_below <= 0.0f) ? below : s;
	s = (residual_above > 0.0f) ? above : s;
	return s;
}
// (Round 3 tried three chains with fewer instructions, each exact for every significand - searched in
// profiles/tools/division_chains.hip - or shown not to be: (a) the hardware root corrected once,
// s + (x - s s) (0.5 v_rsq_f32(x)), and a v_cmp_class_f32 selection for zeros, infinity and NaN: exact,
// 33 against 35 clocks by the price list, but 1 % SLOWER in the kernel (config 3, three runs each on one
// box: 1.603 against 1.587 ms per frame - a second transcendental instruction per root costs more in these
// dependent chains than its issue slot); (b) Markstein's coupled iteration from v_rsq_f32 alone: exact as
// well, the same selection, no cheaper; (c) inversesqrt by divide()'s chain with v_rsq_f32(x) as the
// reciprocal estimate: fails for the two significands whose root has a significand of all ones, 1 - 2^-24
// and 1 - 2^-23 - the squared length of every vector that is normalised a second time, 637 872 pixels of a
// config-3 frame.)
// VKR_SQRT_VARIANT (A/B knob, profiles/r03x/): 0 the selection above; 1 the same with the special cases
// spelled out by a second selection (until round 3); 2 the hardware root corrected once with half of
// v_rsq_f32(x) as the reciprocal and a v_cmp_class_f32 selection for everything that is not a positive normal number
#ifndef VKR_SQRT_VARIANT
#define VKR_SQRT_VARIANT 0
#endif
VKR_DEV float square_root(float x) {
#if VKR_FAST_MATH
	return __builtin_amdgcn_sqrtf(x);
#elif VKR_IEEE_DIVISION_EVERYWHERE
	// (the compiler's correctly rounded root - the default of hipcc; __fsqrt_rn() is NOT: it maps to the native instruction)
	return __builtin_sqrtf(x);
#elif VKR_SQRT_VARIANT == 2
	float estimate = __builtin_amdgcn_sqrtf(x);
	float s = fmaf(fmaf(-estimate, estimate, x), 0.5f * __builtin_amdgcn_rsqf(x), estimate);
	return __builtin_amdgcn_classf(x, 0x100) ? s : estimate;
#elif VKR_SQRT_VARIANT == 1
	float s = square_root_unguarded(x);
	return (x == 0.0f || x == __builtin_inff()) ? x : s;
#else
	// (+-0, +inf, NaN and negative arguments come out as IEEE wants them without a selection:
	// square_root_unguarded says why, tests/test_gpu_arithmetic.py checks it.  Until round 3 a
	// selection spelled it out, variant 1: the same frame time, 1.586 against 1.587 ms.)
	return square_root_unguarded(x);
#endif
}
// inversesqrt as the oracle's math mode 0 (and the reference shader compiled as C++) evaluates it:
// two correctly rounded operations, 1 / sqrt(x)
VKR_DEV float inverse_square_root_ieee(float x) {
#if VKR_IEEE_DIVISION_EVERYWHERE
	return __fdiv_rn(1.0f, __builtin_sqrtf(x));
#elif VKR_SQRT_VARIANT == 2
	return divide(1.0f, square_root(x));
#else
	return divide(1.0f, square_root_unguarded(x));
#endif
}
// GLSL inversesqrt.  Exact mode: integer seed, two Newton steps and one in residual form
// (<= 0.85 ulp), the same single-rounding operations as vkr_rsqrtf in oracle/oracle_math.h;
// 12 instructions instead of the 28 of 1 / sqrt.
VKR_DEV float rsqrt(float x) {
#if VKR_FAST_MATH
	return __builtin_amdgcn_rsqf(x);
#elif VKR_LIBM_MATH && !(VKR_LIBM_EXCEPT & 2)
	return inverse_square_root_ieee(x);
#else
	float hx = 0.5f * x;
	float y = __uint_as_float(0x5F3759DFu - (__float_as_uint(x) >> 1));
	float t = y * y;
	y = y * fmaf(-hx, t, 1.5f);
	t = y * y;
	y = y * fmaf(-hx, t, 1.5f);
	t = y * y;
	y = fmaf(y, fmaf(-hx, t, 0.5f), y);
	// zero, negative numbers, infinity and NaN keep their IEEE results (+-inf, NaN, 0, NaN), which
	// the hardware instruction delivers exactly; the shaders lean on them.  (Denormal arguments
	// would differ from the oracle's 1 / sqrt; sums of squares are zero or far above 1e-38.)
	bool ordinary = x >= 1.17549435e-38f && x < __builtin_inff();
	return ordinary ? y : __builtin_amdgcn_rsqf(x);
#endif
}

// Hybrid arithmetic (VKR_FAST_SHADING, measured once in round 3: profiles/r03_hybrid.md): everything that
// decides WHERE a sample goes - G-buffer decode, LTC matrices, clipping, polygon preparation, sector
// search, the sampled direction, the ray - stays in the translation unit's IEEE arithmetic, so rays and
// NaN-guard pixels are those of the exact frame; only the VALUE of a sample (BRDF, densities, MIS
// weights) uses the approximate reciprocal / root instructions.
#ifndef VKR_FAST_SHADING
#define VKR_FAST_SHADING 0
#endif
VKR_DEV float value_divide(float a, float b) {
#if VKR_FAST_SHADING
	return a * __builtin_amdgcn_rcpf(b);
#else
	return divide(a, b);
#endif
}
VKR_DEV float value_rcp(float x) {
#if VKR_FAST_SHADING
	return __builtin_amdgcn_rcpf(x);
#else
	return rcp(x);
#endif
}
VKR_DEV float value_square_root(float x) {
#if VKR_FAST_SHADING
	return __builtin_amdgcn_sqrtf(x);
#else
	return square_root(x);
#endif
}
VKR_DEV float value_rsqrt(float x) {
#if VKR_FAST_SHADING
	return __builtin_amdgcn_rsqf(x);
#else
	return rsqrt(x);
#endif
}

VKR_DEV f3 normalize(f3 a) { return a * rsqrt(dot(a, a)); }
VKR_DEV f2 normalize(f2 a) { return a * rsqrt(dot(a, a)); }

// 3x3 and 4x3 matrices, column major like GLSL
struct m3 { f3 c[3]; };
struct m43 { f3 c[4]; };
VKR_DEV f3 mul(const m3& m, f3 v) {
	return mk3(
		(m.c[0].x * v.x + m.c[1].x * v.y) + m.c[2].x * v.z,
		(m.c[0].y * v.x + m.c[1].y * v.y) + m.c[2].y * v.z,
		(m.c[0].z * v.x + m.c[1].z * v.y) + m.c[2].z * v.z);
}
VKR_DEV f3 mul_point(const m43& m, f3 p) {
	return mk3(
		((m.c[0].x * p.x + m.c[1].x * p.y) + m.c[2].x * p.z) + m.c[3].x * 1.0f,
		((m.c[0].y * p.x + m.c[1].y * p.y) + m.c[2].y * p.z) + m.c[3].y * 1.0f,
		((m.c[0].z * p.x + m.c[1].z * p.y) + m.c[2].z * p.z) + m.c[3].z * 1.0f);
}
VKR_DEV f3 mul_direction(const m43& m, f3 p) {
	return mk3(
		((m.c[0].x * p.x + m.c[1].x * p.y) + m.c[2].x * p.z) + m.c[3].x * 0.0f,
		((m.c[0].y * p.x + m.c[1].y * p.y) + m.c[2].y * p.z) + m.c[3].y * 0.0f,
		((m.c[0].z * p.x + m.c[1].z * p.y) + m.c[2].z * p.z) + m.c[3].z * 0.0f);
}
VKR_DEV f3 mul_transposed(const m43& m, f3 d) { return mk3(dot(m.c[0], d), dot(m.c[1], d), dot(m.c[2], d)); }

}  // namespace vkr

// glibc's float functions with this file's division and square root (IEEE inside the exponent
// range documented at divide())
#define GM_DIVF(a, b) vkr::divide((a), (b))
#define GM_SQRTF(x) vkr::square_root(x)
#include "glibc_math.h"

namespace vkr {

// VKR_ATAN_TABLE (on wherever glibc's arctangent is used, i.e. in the libm mode): its argument range is looked up in
// an LDS table of 81 rows (1.3 KB per workgroup, fill_atan_rows() at the start of the kernel) instead of found with
// four compares and sixteen selects per call: gm_atanf_rows in glibc_math.h, equal to atanf for all 2^32 arguments
// like gm_atanf.  4.5 % fewer instructions in the config-3 kernel: 1.676 -> 1.610 ms per frame; the V = 7 kernel of
// config 4 loses one of its ten waves per CU to the table and still gains 3 % (profiles/r03k/atan_table.jsonl).
#ifndef VKR_ATAN_TABLE
#define VKR_ATAN_TABLE (VKR_LIBM_MATH && !(VKR_LIBM_EXCEPT & 1))
#endif
#if VKR_ATAN_TABLE
VKR_DEV gm_atan_row_t* atan_rows() {
	__shared__ gm_atan_row_t rows[GM_ATAN_ROW_COUNT];
	return rows;
}
VKR_DEV void fill_atan_rows() {
	for (uint32_t i = threadIdx.x; i < GM_ATAN_ROW_COUNT; i += blockDim.x) atan_rows()[i] = gm_atan_row(i);
	__syncthreads();
}
VKR_DEV float libm_arctangent(float t) { return gm_atanf_rows(t, atan_rows()); }
#else
VKR_DEV void fill_atan_rows() {}
VKR_DEV float libm_arctangent(float t) { return gm_atanf(t); }
#endif

// ---- polynomial transcendentals (coefficients: oracle/tools/fit_math.py) -------

VKR_DEV float atan_unit(float z) {
	float s = z * z;
	float p = -2.508576494e-03f;
	p = fmaf(p, s, 1.399648376e-02f);
	p = fmaf(p, s, -3.667028621e-02f);
	p = fmaf(p, s, 6.318219751e-02f);
	p = fmaf(p, s, -8.689044416e-02f);
	p = fmaf(p, s, 1.104203537e-01f);
	p = fmaf(p, s, -1.427961588e-01f);
	p = fmaf(p, s, 1.999979019e-01f);
	p = fmaf(p, s, -3.333333135e-01f);
	return fmaf(z * s, p, z);
}

VKR_DEV float arctan(float t) {
#if VKR_LIBM_MATH && !(VKR_LIBM_EXCEPT & 1)
	return libm_arctangent(t);
#endif
	float a = fabsf(t);
	bool big = a > 1.0f;
	float z = big ? rcp(a) : a;
	float r = atan_unit(z);
	r = big ? (kHalfPi - r) : r;
	return copysignf(r, t);
}

// atan(n / d) + (n / d < 0 ? pi : 0), positive_atan(n / d) of polygon_sampling.glsl:104-111 with
// ONE division: the range reduction divides the smaller by the larger magnitude directly
// instead of forming the quotient and then its reciprocal.  Operation by operation the mode-1
// o_positive_atan_ratio of oracle/oracle_math.h (which documents the special cases).
VKR_DEV float arctan_ratio_positive(float n, float d) {
#if VKR_LIBM_MATH && !(VKR_LIBM_EXCEPT & 1)
	// as the shader words it: the quotient, its arctangent, pi for a negative quotient
	// (the quotient of an angle next to pi / 2 is as large as floats get)
	float tangent = divide_full_range(n, d);
	return libm_arctangent(tangent) + ((tangent < 0.0f) ? kPi : 0.0f);
#endif
	float a = fabsf(n), b = fabsf(d);
	bool big = a > b;
	float z = divide(big ? b : a, big ? a : b);
	float r = atan_unit(z);
	r = big ? (kHalfPi - r) : r;
	bool differs = ((__float_as_uint(n) ^ __float_as_uint(d)) >> 31) != 0;
	bool negative = differs && (big || z > 0.0f);
	return (differs ? -r : r) + (negative ? kPi : 0.0f);
}

VKR_DEV float asin_tail(float z, float s) {
	float r = 3.392100707e-02f;
	r = fmaf(r, s, 1.700583287e-02f);
	r = fmaf(r, s, 3.113191016e-02f);
	r = fmaf(r, s, 4.459662735e-02f);
	r = fmaf(r, s, 7.500103116e-02f);
	r = fmaf(r, s, 1.666666567e-01f);
	return (z * s) * r;
}

// acos for arguments already clamped to [0, 1]
VKR_DEV float arccos_unit(float x) {
#if VKR_LIBM_MATH && !(VKR_LIBM_EXCEPT & 4)
	return gm_acosf(x);
#endif
	if (x <= 0.5f) {
		float s = x * x;
		return (kHalfPi - x) - asin_tail(x, s);
	}
	float s = (1.0f - x) * 0.5f;
	float z = square_root(s);
	return 2.0f * (z + asin_tail(z, s));
}

// acos on [-1, 1] (o_acos of oracle/oracle_math.h in math mode 1)
VKR_DEV float arccos(float x) {
#if VKR_LIBM_MATH
	return gm_acosf(x);
#endif
	return (x < 0.0f) ? (kPi - arccos_unit(-x)) : arccos_unit(x);
}

// log2 of a positive normal number: exponent + odd series of the mantissa in
// [sqrt(1/2), sqrt(2)]; same operations as vkr_log2f in oracle/oracle_math.h
VKR_DEV float log2_poly(float x) {
#if VKR_FAST_MATH
	return __log2f(x);
#elif VKR_LIBM_MATH
	return gm_log2f(x);
#else
	uint32_t bits = __float_as_uint(x);
	int e = (int) (bits >> 23) - 127;
	float m = __uint_as_float((bits & 0x007FFFFFu) | 0x3F800000u);
	if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
	float s = divide(m - 1.0f, m + 1.0f);
	float z = s * s;
	float r = 2.22222222e-01f;
	r = fmaf(r, z, 2.85714298e-01f);
	r = fmaf(r, z, 4.00000006e-01f);
	r = fmaf(r, z, 6.66666687e-01f);
	r = fmaf(r, z, 2.0f);
	return fmaf(s * r, 1.44269502f, (float) e);
#endif
}

// 2^t, pow(x, 1/3) for x >= 0 and the two-argument arctangent: the same operations as
// vkr_exp2f / vkr_cbrt_positive / vkr_atan2f in oracle/oracle_math.h (exact mode)
VKR_DEV float exp2_poly(float t) {
	float n = rintf(t);
	float r = (t - n) * 0.693147182f;
	float p = 1.98412698e-04f;
	p = fmaf(p, r, 1.38888892e-03f);
	p = fmaf(p, r, 8.33333377e-03f);
	p = fmaf(p, r, 4.16666679e-02f);
	p = fmaf(p, r, 1.66666672e-01f);
	p = fmaf(p, r, 0.5f);
	p = fmaf(p, r, 1.0f);
	p = fmaf(p, r, 1.0f);
	return p * __uint_as_float((uint32_t) ((int) n + 127) << 23);
}
VKR_DEV float cube_root_positive(float x) {
#if VKR_FAST_MATH
	return __powf(x, 1.0f / 3.0f);
#elif VKR_LIBM_MATH
	return gm_powf(x, 1.0f / 3.0f);
#else
	if (!(x > 0.0f)) return x;
	return exp2_poly(log2_poly(x) * (1.0f / 3.0f));
#endif
}
VKR_DEV float arctan(float t);
VKR_DEV float arctan2(float y, float x) {
#if VKR_FAST_MATH
	return atan2f(y, x);
#elif VKR_LIBM_MATH
	return gm_atan2f(y, x);
#else
	float a = arctan(divide(y, x));
	if (x < 0.0f) a += (y >= 0.0f) ? kPi : -kPi;
	return a;
#endif
}

VKR_DEV void sincos_poly(float x, float& out_sin, float& out_cos) {
#if VKR_LIBM_MATH && !(VKR_LIBM_EXCEPT & 4)
	gm_sincosf(x, &out_sin, &out_cos);
	return;
#endif
	const float two_over_pi = 0.63661977236758134308f;
	const float pio2_hi = 1.57079637050628662109375f;
	const float pio2_lo = -4.37113882867379e-8f;
	float k = rintf(x * two_over_pi);
	float r = fmaf(-k, pio2_hi, x);
	r = fmaf(-k, pio2_lo, r);
	float s = r * r;
	float ps = 2.724694696e-06f;
	ps = fmaf(ps, s, -1.984006376e-04f);
	ps = fmaf(ps, s, 8.333331905e-03f);
	ps = fmaf(ps, s, -1.666666716e-01f);
	float sn = fmaf(r * s, ps, r);
	float pc = -2.729846358e-07f;
	pc = fmaf(pc, s, 2.480058174e-05f);
	pc = fmaf(pc, s, -1.388888806e-03f);
	pc = fmaf(pc, s, 4.166666791e-02f);
	float cs = fmaf(s * s, pc, fmaf(-0.5f, s, 1.0f));
	int q = ((int) k) & 3;
	float s_out = (q & 1) ? cs : sn;
	float c_out = (q & 1) ? sn : cs;
	out_sin = (q & 2) ? -s_out : s_out;
	out_cos = ((q + 1) & 2) ? -c_out : c_out;
}

}  // namespace vkr
