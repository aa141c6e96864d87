/* TEST INFRASTRUCTURE — not part of the product.
 *
 * Scalar fp32 helpers for the CPU oracle: GLSL-like vector types with a fixed,
 * documented evaluation order, and the deterministic transcendental functions
 * ("vkr math") whose polynomial form is mirrored by the HIP kernels so that
 * CPU and GPU results can be compared bit for bit.  The oracle can also be
 * switched to libm (math_mode 0) to show that nothing hides behind the
 * polynomials.
 *
 * GLSL leaves the precision of atan/acos/sin/cos/inversesqrt to the driver
 * (reference: src/shaders/polygon_sampling.glsl:79-82 quotes "at most 2 ulps" for
 * native atan on Turing).  Both modes here are within 2-3 ulp of the exact value
 * (tests/test_oracle_properties.py::test_deterministic_math_is_accurate).
 *
 * Everything must be compiled with -ffp-contract=off; fused operations appear
 * only where the GLSL says fma(). */
#ifndef ORACLE_MATH_H
#define ORACLE_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>
/* glibc 2.35's float functions restated operation for operation (the header the kernels' "libm" arithmetic
 * mode is compiled from; plain C99 here).  The oracle evaluates ITS transcendentals through this restatement
 * by default, so that its frames - and every golden fixture made from them - are the same on any IEEE machine,
 * whatever C library and CPU it has.  That the restatement equals the C library the reference shader was
 * compiled against in this image (glibc 2.35, x86-64, FMA / AVX2 IFUNC variants) for every float is a separate
 * claim with its own tests (tests/test_glibc_math.py, oracle/tools/check_glibc_math.c);
 * oracle_set_libm_source(1) switches the oracle to the machine's C library itself. */
#include "../vulkan_renderer_amd/csrc/glibc_math.h"

/* 0 (default): gm_* restatement; 1: the C library of this machine */
extern int g_oracle_system_libm;
static inline float l_atanf(float x) { return g_oracle_system_libm ? atanf(x) : gm_atanf(x); }
static inline float l_acosf(float x) { return g_oracle_system_libm ? acosf(x) : gm_acosf(x); }
static inline float l_sinf(float x) { return g_oracle_system_libm ? sinf(x) : gm_sinf(x); }
static inline float l_cosf(float x) { return g_oracle_system_libm ? cosf(x) : gm_cosf(x); }
static inline float l_log2f(float x) { return g_oracle_system_libm ? log2f(x) : gm_log2f(x); }
static inline float l_powf(float x, float y) { return g_oracle_system_libm ? powf(x, y) : gm_powf(x, y); }
static inline float l_atan2f(float y, float x) { return g_oracle_system_libm ? atan2f(y, x) : gm_atan2f(y, x); }

#define O_PI 3.1415926535897932384626433832795f
#define O_INV_PI 0.31830988618379067153776752674503f
#define O_HALF_PI 1.5707963267948966192313216916398f

typedef struct { float x, y; } v2;
typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;
/* column-major like GLSL: c[i] is column i */
typedef struct { v3 c[3]; } m3;
typedef struct { v3 c[4]; } m43;
typedef struct { v2 c[2]; } m2;

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static inline v2 mk2(float x, float y) { v2 r = {x, y}; return r; }
static inline v3 mk3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v2 add2(v2 a, v2 b) { return mk2(a.x + b.x, a.y + b.y); }
static inline v2 sub2(v2 a, v2 b) { return mk2(a.x - b.x, a.y - b.y); }
static inline v2 scale2(v2 a, float s) { return mk2(a.x * s, a.y * s); }
static inline v2 neg2(v2 a) { return mk2(-a.x, -a.y); }
static inline v3 add3(v3 a, v3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 scale3(v3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
static inline v3 mul3(v3 a, v3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 neg3(v3 a) { return mk3(-a.x, -a.y, -a.z); }
/* dot products accumulate left to right, no fusing */
static inline float dot2(v2 a, v2 b) { return a.x * b.x + a.y * b.y; }
static inline float dot3(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 cross3(v3 a, v3 b) {
	return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* fma(vec3(s), a, b) */
static inline v3 fma3s(float s, v3 a, v3 b) { return mk3(fmaf(s, a.x, b.x), fmaf(s, a.y, b.y), fmaf(s, a.z, b.z)); }
static inline v2 fma2s(float s, v2 a, v2 b) { return mk2(fmaf(s, a.x, b.x), fmaf(s, a.y, b.y)); }
/* GLSL inversesqrt.  Math mode 0: 1 / sqrt (two correctly rounded operations).  Math
 * mode 1 (the form shared with the GPU, where 1 / sqrt costs 28 instructions): integer
 * seed, two Newton steps and a third one in residual form; at most 0.85 ulp off over
 * the whole positive range (oracle/tools/fit_math.py checks that), deterministic because
 * every step is a single IEEE operation. */
extern int g_oracle_math_mode;
static inline float vkr_rsqrtf(float x) {
	/* zero, negative numbers, infinity and NaN keep their IEEE results (+-inf, NaN, 0, NaN):
	 * the shaders lean on them (e.g. 0 * inversesqrt(0) = NaN, then max(0, NaN) = 0) */
	if (!(x >= 1.17549435e-38f && x < INFINITY)) return 1.0f / sqrtf(x);
	float hx = 0.5f * x;
	float y = u2f(0x5F3759DFu - (f2u(x) >> 1));
	float t = y * y;
	y = y * fmaf(-hx, t, 1.5f);
	t = y * y;
	y = y * fmaf(-hx, t, 1.5f);
	t = y * y;
	return fmaf(y, fmaf(-hx, t, 0.5f), y);
}
/* (until round 3 math mode 1 used vkr_rsqrtf; it is kept for tests of the function itself) */
static inline float rsqrt_f(float x) { return 1.0f / sqrtf(x); }
static inline v3 normalize3(v3 a) { return scale3(a, rsqrt_f(dot3(a, a))); }
static inline v2 normalize2(v2 a) { return scale2(a, rsqrt_f(dot2(a, a))); }
static inline float clamp_f(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
static inline v2 rot90(v2 a) { return mk2(-a.y, a.x); }
/* M * v = c0*v.x + c1*v.y + c2*v.z, accumulated left to right */
static inline v3 m3_mul(const m3* m, v3 v) {
	return mk3(
		(m->c[0].x * v.x + m->c[1].x * v.y) + m->c[2].x * v.z,
		(m->c[0].y * v.x + m->c[1].y * v.y) + m->c[2].y * v.z,
		(m->c[0].z * v.x + m->c[1].z * v.y) + m->c[2].z * v.z);
}
/* mat4x3 * vec4(p, w) */
static inline v3 m43_mul(const m43* m, v3 p, float w) {
	return mk3(
		((m->c[0].x * p.x + m->c[1].x * p.y) + m->c[2].x * p.z) + m->c[3].x * w,
		((m->c[0].y * p.x + m->c[1].y * p.y) + m->c[2].y * p.z) + m->c[3].y * w,
		((m->c[0].z * p.x + m->c[1].z * p.y) + m->c[2].z * p.z) + m->c[3].z * w);
}
/* (transpose(mat4x3) * d).xyz */
static inline v3 m43_mul_transposed(const m43* m, v3 d) {
	return mk3(dot3(m->c[0], d), dot3(m->c[1], d), dot3(m->c[2], d));
}

/* ---- deterministic transcendental functions (mirrored in the HIP kernels) -- */

/* 0 = libm, 1 = polynomial "vkr math" */
extern int g_oracle_math_mode;

static inline float poly_atan_core(float z) {
	/* atan(z) for z in [0,1]: z + z*s*P(s); coefficients from oracle/tools/fit_math.py */
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

static inline float vkr_atanf(float t) {
	float a = fabsf(t);
	int big = a > 1.0f;
	float z = big ? (1.0f / a) : a;
	float r = poly_atan_core(z);
	r = big ? (O_HALF_PI - r) : r;
	return copysignf(r, t);
}

static inline float poly_asin_tail(float z, float s) {
	/* asin(z) - z = z*s*R(s), s = z*z <= 0.25 */
	float r = 3.392100707e-02f;
	r = fmaf(r, s, 1.700583287e-02f);
	r = fmaf(r, s, 3.113191016e-02f);
	r = fmaf(r, s, 4.459662735e-02f);
	r = fmaf(r, s, 7.500103116e-02f);
	r = fmaf(r, s, 1.666666567e-01f);
	return (z * s) * r;
}

/* acos for arguments in [0,1] (the shaders always clamp first) */
static inline float vkr_acosf_unit(float x) {
	if (x <= 0.5f) {
		float s = x * x;
		return (O_HALF_PI - x) - poly_asin_tail(x, s);
	}
	else {
		float s = (1.0f - x) * 0.5f;
		float z = sqrtf(s);
		return 2.0f * (z + poly_asin_tail(z, s));
	}
}

static inline void vkr_sincosf(float x, float* out_sin, float* out_cos) {
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
	s_out = (q & 2) ? -s_out : s_out;
	c_out = ((q + 1) & 2) ? -c_out : c_out;
	*out_sin = s_out;
	*out_cos = c_out;
}

/* log2 for positive normal x: exponent + odd series of the mantissa in [sqrt(1/2), sqrt(2)]
 * (s = (m - 1) / (m + 1), log2(m) = 2 / ln 2 (s + s^3/3 + ...)), < 1 ulp off at the sizes
 * the error display feeds it.  Mirrored by log2_poly in csrc/device_math.h. */
static inline float vkr_log2f(float x) {
	uint32_t bits = f2u(x);
	int e = (int) (bits >> 23) - 127;
	float m = u2f((bits & 0x007FFFFFu) | 0x3F800000u);
	if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
	float s = (m - 1.0f) / (m + 1.0f);
	float z = s * s;
	float r = 2.22222222e-01f;
	r = fmaf(r, z, 2.85714298e-01f);
	r = fmaf(r, z, 4.00000006e-01f);
	r = fmaf(r, z, 6.66666687e-01f);
	r = fmaf(r, z, 2.0f);
	return fmaf(s * r, 1.44269502f, (float) e);
}

/* 2^t for |t| < 120: integer part into the exponent, fraction in [-0.5, 0.5] by a Taylor
 * polynomial of degree 7 in t ln 2 (relative error < 2e-7).  Mirrored by exp2_poly. */
static inline float vkr_exp2f(float t) {
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
	return p * u2f((uint32_t) ((int) n + 127) << 23);
}
/* pow(x, 1/3) for x >= 0 as the cubic solver of the reference needs it (cubic_solver.glsl:66) */
static inline float vkr_cbrt_positive(float x) {
	if (!(x > 0.0f)) return x;
	return vkr_exp2f(vkr_log2f(x) * (1.0f / 3.0f));
}
/* two-argument arctangent from the one-argument form */
static inline float vkr_atanf(float t);
static inline float vkr_atan2f(float y, float x) {
	float a = vkr_atanf(y / x);
	if (x < 0.0f) a += (y >= 0.0f) ? O_PI : -O_PI;
	return a;
}

/* Math mode 1 ("deterministic", mirrored bit for bit by the kernels' "exact" arithmetic mode) is
 * math mode 0 with ONE function replaced, the arctangent: polynomial with explicit FMAs, one
 * division per arctangent of a ratio.  Everything else - inversesqrt as 1 / sqrt, acos, sin, cos,
 * log2, pow, atan2 - is the C library in both modes.  (Until round 3 mode 1 replaced all of them;
 * the Newton inversesqrt turned out to be what moved pixels into and out of the shader's NaN guard.) */
static inline float o_log2(float x) { return l_log2f(x); }
static inline float o_atan2(float y, float x) { return l_atan2f(y, x); }
static inline float o_pow_third(float x) { return l_powf(x, 1.0f / 3.0f); }
static inline float o_atan(float t) { return g_oracle_math_mode ? vkr_atanf(t) : l_atanf(t); }
static inline float o_acos(float x) { return l_acosf(x); }
/* atan(n / d) + (n / d < 0 ? pi : 0), i.e. positive_atan(n / d) of polygon_sampling.glsl:104-111.
 * Mode 0 evaluates exactly that with libm.  Mode 1 is the fused form shared with the GPU: the
 * range reduction divides the smaller by the larger magnitude directly, ONE division instead of
 * the quotient followed by a reciprocal; same special cases (0 / 0 = NaN, x / 0 -> pi / 2, a
 * quotient of -0 gives +0).  arctan_ratio in csrc/device_math.h mirrors it operation by operation. */
static inline float o_positive_atan_ratio(float n, float d) {
	if (!g_oracle_math_mode) {
		float tangent = n / d;
		return l_atanf(tangent) + ((tangent < 0.0f) ? O_PI : 0.0f);
	}
	float a = fabsf(n), b = fabsf(d);
	int big = a > b;
	float z = (big ? b : a) / (big ? a : b);
	float r = poly_atan_core(z);
	r = big ? (O_HALF_PI - r) : r;
	int differs = ((f2u(n) ^ f2u(d)) >> 31) != 0;
	int negative = differs && (big || z > 0.0f);
	return (differs ? -r : r) + (negative ? O_PI : 0.0f);
}
static inline float o_acos_unit(float x) { return l_acosf(x); }
static inline void o_sincos(float x, float* s, float* c) { *s = l_sinf(x); *c = l_cosf(x); }

#endif
