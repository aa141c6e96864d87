/* The float functions of glibc 2.35's libm (x86-64), operation by operation.
 *
 * Why this exists: the reference leaves the precision of atan / acos / sin / cos / log2 / pow to
 * the GLSL driver (src/shaders/polygon_sampling.glsl:79-82).  The CPU oracle that is pinned bit for
 * bit against the reference's own shader source (compiled as C++, oracle/ref_stubs) evaluates them
 * with the C library of this image, glibc 2.35.  The "libm" arithmetic mode of the HIP kernels
 * (VKR_MATH_MODE 2, device_math.h) evaluates the very same operations, so that a frame shaded on
 * the GPU equals the frame of that oracle - and with it the reference's shader arithmetic - in
 * every bit, including the pixels where a NaN guard or a shadow-ray silhouette magnifies a
 * last-bit difference into a visible one.
 *
 * What is restated (the functions the oracle calls in math mode 0):
 *   atanf, acosf, atan2f    sysdeps/ieee754/flt-32/{s_atanf,e_acosf,e_atan2f}.c: the Sun fdlibm
 *                           single-precision routines; plain fp32 operations, no contraction
 *                           (these have no FMA build in glibc 2.35).
 *   sinf, cosf              sysdeps/ieee754/flt-32/{s_sinf,s_cosf,sincosf.h}: double-precision
 *                           polynomials after a reduction by pi / 2 (Arm optimized routines).
 *   log2f, powf             sysdeps/ieee754/flt-32/{e_log2f,e_powf}.c with their tables.
 * The last three groups are IFUNCs; on every x86-64 CPU with FMA and AVX2 (this image's host CPUs)
 * the loader picks the build compiled with -mfma -mavx2, in which the compiler fused every
 * a * b + c of the source.  The FMAs below are placed exactly where that build has them (read off
 * the instructions of libm.so.6; oracle/tools/check_glibc_math.c compares every function with the
 * C library over all 2^32 arguments - two-argument functions over random and special pairs).
 * Coefficients and tables are the published ones of fdlibm and of the Arm optimized routines.
 * Results for NaN arguments are some quiet NaN (payloads are not reproduced); errno and floating
 * point exception flags do not exist here.
 *
 * Plain C99 and HIP device code at the same time: the including translation unit must be compiled
 * with -ffp-contract=off.
 *
 * Licence note.  This file restates THIRD-PARTY algorithms, not the reference renderer's: the GNU C Library 2.35
 * (LGPL-2.1-or-later), whose float routines named above come from Sun's fdlibm ("Copyright (C) 1993 by Sun Microsystems,
 * Inc.  Permission to use, copy, modify, and distribute this software is freely granted, provided that this notice is
 * preserved.") and from the Arm Optimized Routines (MIT / Apache-2.0 with LLVM exception).  The polynomial coefficients
 * and the log2 / exp2 tables are necessarily those publications' numbers - bit-equality with the library is the point -
 * while the code around them is written for this file (branch-free range selection, the row table of gm_atanf_rows, FMA
 * placement read off the shipped binary).  NOTICE.md at the repository root repeats this. */
#ifndef VKR_GLIBC_MATH_H
#define VKR_GLIBC_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define GM_FN __device__ __forceinline__
#define GM_TABLE static __device__ const
#define GM_FMA(a, b, c) __builtin_fma((a), (b), (c))
/* GM_DIVF / GM_SQRTF: the correctly rounded quotient and square root.  device_math.h defines them
 * as its divide() / square_root() before it includes this file; the fallbacks are the compiler's
 * expansions of / and sqrtf (IEEE under -ffp-contract=off, HIP's default). */
#ifndef GM_DIVF
#define GM_DIVF(a, b) __fdiv_rn((a), (b))
#endif
#ifndef GM_SQRTF
#define GM_SQRTF(x) __builtin_sqrtf(x) /* (correctly rounded under hipcc's default; __fsqrt_rn() maps to the native instruction) */
#endif
GM_FN uint32_t gm_bits(float f) { return __float_as_uint(f); }
GM_FN float gm_float(uint32_t u) { return __uint_as_float(u); }
GM_FN uint64_t gm_bits64(double d) { return (uint64_t) __double_as_longlong(d); }
GM_FN double gm_double(uint64_t u) { return __longlong_as_double((long long) u); }
#else
#include <math.h>
#include <string.h>
#define GM_FN static inline
#define GM_TABLE static const
#define GM_FMA(a, b, c) fma((a), (b), (c))
#define GM_SQRTF(x) sqrtf(x)
#define GM_DIVF(a, b) ((a) / (b))
GM_FN uint32_t gm_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
GM_FN float gm_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
GM_FN uint64_t gm_bits64(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
GM_FN double gm_double(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
#endif

GM_FN float gm_quiet_nan(void) { return gm_float(0x7FC00000u); }
GM_FN float gm_fabsf(float x) { return gm_float(gm_bits(x) & 0x7FFFFFFFu); }

/* ---- atanf (s_atanf.c) --------------------------------------------------------------------- */

/* Written without branches on the five argument ranges of the source.  The three middle ranges
 * (7/16 <= |x| < 11/16, < 19/16, < 39/16) reduce with t = (|x| - c) / (1 + c |x|), c = 1/2, 1, 3/2 - the
 * source's (2|x| - 1) / (2 + |x|) is that quotient with numerator and denominator doubled, which
 * changes no rounding - and the first range (|x| < 7/16, no reduction) is the same expression with
 * c = 0: t = |x| / 1.  Only the last range, t = -1 / |x|, is selected separately.  Every range ends in
 * hi - ((t P(t^2) - lo) - t) with hi + lo = atan(c) (0 for the first range, where the expression equals
 * the source's x - x P(x^2) in every bit); the sign of x is applied at the end (all operations are
 * odd-symmetric under round to nearest).  The source's shortcut for |x| < 2^-29 (return x) needs no
 * case: there t P(t^2) is below half an ulp of t.  Lanes of a wave never diverge here. */
GM_FN float gm_atanf(float x) {
	const float at0 = 3.3333334327e-01f, at1 = -2.0000000298e-01f, at2 = 1.4285714924e-01f, at3 = -1.1111110449e-01f,
		at4 = 9.0908870101e-02f, at5 = -7.6918758452e-02f, at6 = 6.6610731184e-02f, at7 = -5.8335702866e-02f,
		at8 = 4.9768779427e-02f, at9 = -3.6531571299e-02f, at10 = 1.6285819933e-02f;
	uint32_t hx = gm_bits(x), ix = hx & 0x7FFFFFFFu;
	float ax = gm_float(ix);
	int r1 = ix >= 0x3EE00000u, r2 = ix >= 0x3F300000u, r3 = ix >= 0x3F980000u, r4 = ix >= 0x401C0000u;
	float c = r1 ? 0.5f : 0.0f;
	c = r2 ? 1.0f : c;
	c = r3 ? 1.5f : c;
	float n = ax - c;
	float d = 1.0f + c * ax;
	n = r4 ? -1.0f : n;
	d = r4 ? ax : d;
	float hi = r1 ? 4.6364760399e-01f : 0.0f, lo = r1 ? 5.0121582440e-09f : 0.0f;
	hi = r2 ? 7.8539812565e-01f : hi; lo = r2 ? 3.7748947079e-08f : lo;
	hi = r3 ? 9.8279368877e-01f : hi; lo = r3 ? 3.4473217170e-08f : lo;
	hi = r4 ? 1.5707962513e+00f : hi; lo = r4 ? 7.5497894159e-08f : lo;
	float t = GM_DIVF(n, d);
	float z = t * t;
	float w = z * z;
	float s1 = z * (at0 + w * (at2 + w * (at4 + w * (at6 + w * (at8 + w * at10)))));
	float s2 = w * (at1 + w * (at3 + w * (at5 + w * (at7 + w * at9))));
	float r = hi - ((t * (s1 + s2) - lo) - t);
	/* 2^25 <= |x| <= inf: atanhi[3] + atanlo[3] (a NaN passes through the arithmetic above) */
	r = (ix - 0x4C000000u <= 0x7F800000u - 0x4C000000u) ? (1.5707962513e+00f + 7.5497894159e-08f) : r;
	return gm_float(gm_bits(r) | (hx & 0x80000000u));
}

/* The same function with the argument range looked up instead of compared: `rows` holds, for every
 * value of the top bits of |x| that the five ranges can be told apart by (their bounds 7/16, 11/16,
 * 19/16, 39/16 are multiples of 2^18 as bit patterns), the four numbers of the range:
 * (c, s, atanhi, atanlo) with t = (s |x| - c) / (c |x| + s): s = 1 for the first four ranges (c = 0, 1/2, 1,
 * 3/2), and c = 1, s = 0 for the last one, t = -1 / |x|.  One 16-byte read replaces four compares and
 * sixteen selects; on the GPU the table lives in LDS (device_math.h).  gm_fill_atan_rows() makes it. */
#define GM_ATAN_ROW_COUNT 81
typedef struct __attribute__((aligned(16))) gm_atan_row_s { float c, s, hi, lo; } gm_atan_row_t;
GM_FN gm_atan_row_t gm_atan_row(uint32_t index) {
	gm_atan_row_t r;
	if (index < 1u) { r.c = 0.0f; r.s = 1.0f; r.hi = 0.0f; r.lo = 0.0f; }
	else if (index < 21u) { r.c = 0.5f; r.s = 1.0f; r.hi = 4.6364760399e-01f; r.lo = 5.0121582440e-09f; }
	else if (index < 47u) { r.c = 1.0f; r.s = 1.0f; r.hi = 7.8539812565e-01f; r.lo = 3.7748947079e-08f; }
	else if (index < 80u) { r.c = 1.5f; r.s = 1.0f; r.hi = 9.8279368877e-01f; r.lo = 3.4473217170e-08f; }
	else { r.c = 1.0f; r.s = 0.0f; r.hi = 1.5707962513e+00f; r.lo = 7.5497894159e-08f; }
	return r;
}
GM_FN float gm_atanf_rows(float x, const gm_atan_row_t* rows) {
	const float at0 = 3.3333334327e-01f, at1 = -2.0000000298e-01f, at2 = 1.4285714924e-01f, at3 = -1.1111110449e-01f,
		at4 = 9.0908870101e-02f, at5 = -7.6918758452e-02f, at6 = 6.6610731184e-02f, at7 = -5.8335702866e-02f,
		at8 = 4.9768779427e-02f, at9 = -3.6531571299e-02f, at10 = 1.6285819933e-02f;
	uint32_t hx = gm_bits(x), ix = hx & 0x7FFFFFFFu;
	float ax = gm_float(ix);
	int32_t index = (int32_t) (ix >> 18) - 0xFB7;
	index = index < 0 ? 0 : (index > 80 ? 80 : index);
	gm_atan_row_t row = rows[index];
#if defined(__HIPCC__)
	float n = __builtin_fmaf(row.s, ax, -row.c);
#else
	float n = fmaf(row.s, ax, -row.c);
#endif
	float d = row.c * ax + row.s;
	float t = GM_DIVF(n, d);
	float z = t * t;
	float w = z * z;
	float s1 = z * (at0 + w * (at2 + w * (at4 + w * (at6 + w * (at8 + w * at10)))));
	float s2 = w * (at1 + w * (at3 + w * (at5 + w * (at7 + w * at9))));
	float r = row.hi - ((t * (s1 + s2) - row.lo) - t);
	r = (ix - 0x4C000000u <= 0x7F800000u - 0x4C000000u) ? (1.5707962513e+00f + 7.5497894159e-08f) : r;
	return gm_float(gm_bits(r) | (hx & 0x80000000u));
}

/* ---- acosf (e_acosf.c; the wrapper only adds errno) ------------------------------------------ */

GM_FN float gm_acosf(float x) {
	const float pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f;
	const float ps0 = 1.6666667163e-01f, ps1 = -3.2556581497e-01f, ps2 = 2.0121252537e-01f, ps3 = -4.0055535734e-02f,
		ps4 = 7.9153501429e-04f, ps5 = 3.4793309169e-05f;
	const float qs1 = -2.4033949375e+00f, qs2 = 2.0209457874e+00f, qs3 = -6.8828397989e-01f, qs4 = 7.7038154006e-02f;
	uint32_t hx = gm_bits(x), ix = hx & 0x7FFFFFFFu;
	if (ix == 0x3F800000u) return ((int32_t) hx > 0) ? 0.0f : (pi + 2.0f * pio2_lo);
	if (ix > 0x3F800000u) return gm_quiet_nan();
	if (ix < 0x3F000000u) {
		if (ix <= 0x32800000u) return pio2_hi + pio2_lo;
		float z = x * x;
		float p = z * (ps0 + z * (ps1 + z * (ps2 + z * (ps3 + z * (ps4 + z * ps5)))));
		float q = 1.0f + z * (qs1 + z * (qs2 + z * (qs3 + z * qs4)));
		float r = GM_DIVF(p, q);
		return pio2_hi - (x - (pio2_lo - x * r));
	}
	if ((int32_t) hx < 0) {
		float z = (1.0f + x) * 0.5f;
		float p = z * (ps0 + z * (ps1 + z * (ps2 + z * (ps3 + z * (ps4 + z * ps5)))));
		float q = 1.0f + z * (qs1 + z * (qs2 + z * (qs3 + z * qs4)));
		float s = GM_SQRTF(z);
		float r = GM_DIVF(p, q);
		float w = r * s - pio2_lo;
		return pi - 2.0f * (s + w);
	}
	float z = (1.0f - x) * 0.5f;
	float s = GM_SQRTF(z);
	float df = gm_float(gm_bits(s) & 0xFFFFF000u);
	float c = GM_DIVF(z - df * df, s + df);
	float p = z * (ps0 + z * (ps1 + z * (ps2 + z * (ps3 + z * (ps4 + z * ps5)))));
	float q = 1.0f + z * (qs1 + z * (qs2 + z * (qs3 + z * qs4)));
	float r = GM_DIVF(p, q);
	float w = r * s + c;
	return 2.0f * (df + w);
}

/* ---- atan2f (e_atan2f.c) -------------------------------------------------------------------- */

GM_FN float gm_atan2f(float y, float x) {
	const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
	uint32_t hx = gm_bits(x), hy = gm_bits(y), ix = hx & 0x7FFFFFFFu, iy = hy & 0x7FFFFFFFu;
	if (ix > 0x7F800000u || iy > 0x7F800000u) return x + y;
	if (hx == 0x3F800000u) return gm_atanf(y);
	uint32_t m = ((hy >> 31) & 1u) | ((hx >> 30) & 2u);
	if (iy == 0) {
		if (m < 2) return y;
		return (m == 2) ? (pi + tiny) : (-pi - tiny);
	}
	if (ix == 0) return ((int32_t) hy < 0) ? (-pi_o_2 - tiny) : (pi_o_2 + tiny);
	if (ix == 0x7F800000u) {
		if (iy == 0x7F800000u) {
			switch (m) {
			case 0: return pi_o_4 + tiny;
			case 1: return -pi_o_4 - tiny;
			case 2: return 3.0f * pi_o_4 + tiny;
			default: return -3.0f * pi_o_4 - tiny;
			}
		}
		switch (m) {
		case 0: return 0.0f;
		case 1: return -0.0f;
		case 2: return pi + tiny;
		default: return -pi - tiny;
		}
	}
	if (iy == 0x7F800000u) return ((int32_t) hy < 0) ? (-pi_o_2 - tiny) : (pi_o_2 + tiny);
	int32_t k = ((int32_t) iy - (int32_t) ix) >> 23;
	float z;
	if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
	else if ((int32_t) hx < 0 && k < -60) z = 0.0f;
	else z = gm_atanf(gm_fabsf(GM_DIVF(y, x)));
	switch (m) {
	case 0: return z;
	case 1: return gm_float(gm_bits(z) ^ 0x80000000u);
	case 2: return pi - (z - pi_lo);
	default: return (z - pi_lo) - pi;
	}
}

/* ---- sinf, cosf (s_sinf.c, s_cosf.c, sincosf.h; the -mfma build) ------------------------------ */

/* 4 / pi as a string of bits, 24 overlapping 32-bit windows (__inv_pio4) */
GM_TABLE uint32_t gm_inv_pio4[24] = {
	0xa2u, 0xa2f9u, 0xa2f983u, 0xa2f9836eu, 0xf9836e4eu, 0x836e4e44u, 0x6e4e4415u, 0x4e441529u,
	0x441529fcu, 0x1529fc27u, 0x29fc2757u, 0xfc2757d1u, 0x2757d1f5u, 0x57d1f534u, 0xd1f534ddu, 0xf534ddc0u,
	0x34ddc0dbu, 0xddc0db62u, 0xc0db6295u, 0xdb629599u, 0x6295993cu, 0x95993c43u, 0x993c4390u, 0x3c439041u};

/* sinf_poly of sincosf.h with the first table (the second one negates the cosine coefficients and
 * nothing else, which negates the cosine polynomial exactly and leaves the sine polynomial alone) */
GM_FN double gm_sin_polynomial(double x, double x2) {
	const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
	double x3 = x * x2;
	double t = GM_FMA(s3, x2, s2);
	double x7 = x3 * x2;
	double s = GM_FMA(x3, s1, x);
	return GM_FMA(t, x7, s);
}
GM_FN double gm_cos_polynomial(double x2) {
	const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5, c3 = -0x1.6c087e89a359dp-10, c4 = 0x1.99343027bf8c3p-16;
	double x4 = x2 * x2;
	double t1 = GM_FMA(c1, x2, c0);
	double t2 = GM_FMA(c4, x2, c3);
	double x6 = x4 * x2;
	double c = GM_FMA(x4, c2, t1);
	return GM_FMA(t2, x6, c);
}

/* Reduction of y with 2^-12 <= |y| < inf that is not below pi / 4: returns x in [-pi / 4, pi / 4]
 * and the quadrant in *n (plus the sign of y in *sign for the large path, where |y| is reduced) */
GM_FN double gm_reduce(float y, int* n, int* sign) {
	uint32_t xi = gm_bits(y);
	*sign = 0;
	if (((xi >> 20) & 0x7FFu) < 0x42Fu) {
		/* |y| < 120: reduce_fast without rounding intrinsics */
		double x = (double) y;
		double r = x * 0x1.45F306DC9C883p+23;
		int k = ((int32_t) r + 0x800000) >> 24;
		*n = k;
		return GM_FMA(-(double) k, 0x1.921FB54442D18p0, x);
	}
	/* reduce_large */
	const uint32_t* arr = gm_inv_pio4 + ((xi >> 26) & 15u);
	uint32_t shift = (xi >> 23) & 7u;
	*sign = (int) (xi >> 31);
	xi = (xi & 0xFFFFFFu) | 0x800000u;
	xi <<= shift;
	uint64_t res0 = (uint64_t) (uint32_t) (xi * arr[0]);
	uint64_t res1 = (uint64_t) xi * arr[4];
	uint64_t res2 = (uint64_t) xi * arr[8];
	res0 = (res2 >> 32) | (res0 << 32);
	res0 += res1;
	uint64_t k = (res0 + (1ull << 61)) >> 62;
	res0 -= k << 62;
	*n = (int) k;
	return (double) (int64_t) res0 * 0x1.921FB54442D18p-62;
}

GM_FN float gm_sinf(float y) {
	uint32_t top = (gm_bits(y) >> 20) & 0x7FFu;
	if (top < 0x3F4u) {
		if (top < 0x398u) return y;
		double x = (double) y;
		return (float) gm_sin_polynomial(x, x * x);
	}
	if (top >= 0x7F8u) return gm_quiet_nan();
	int n, sign;
	double x = gm_reduce(y, &n, &sign);
	int q = n + sign;
	double s = ((q & 3) == 1 || (q & 3) == 2) ? -1.0 : 1.0;
	if ((n & 1) == 0) return (float) gm_sin_polynomial(x * s, x * x);
	double c = gm_cos_polynomial(x * x);
	return (float) ((q & 2) ? -c : c);
}

GM_FN float gm_cosf(float y) {
	uint32_t top = (gm_bits(y) >> 20) & 0x7FFu;
	if (top < 0x3F4u) {
		if (top < 0x398u) return 1.0f;
		double x = (double) y;
		return (float) gm_cos_polynomial(x * x);
	}
	if (top >= 0x7F8u) return gm_quiet_nan();
	int n, sign;
	double x = gm_reduce(y, &n, &sign);
	int q = n + sign;
	double s = ((q & 3) == 1 || (q & 3) == 2) ? -1.0 : 1.0;
	if ((n & 1) != 0) return (float) gm_sin_polynomial(x * s, x * x);
	double c = gm_cos_polynomial(x * x);
	return (float) ((q & 2) ? -c : c);
}

/* sinf and cosf of one argument with one reduction (the two results are those of the functions above) */
GM_FN void gm_sincosf(float y, float* out_sin, float* out_cos) {
	uint32_t top = (gm_bits(y) >> 20) & 0x7FFu;
	if (top < 0x3F4u) {
		double x = (double) y, x2 = x * x;
		*out_sin = (top < 0x398u) ? y : (float) gm_sin_polynomial(x, x2);
		*out_cos = (top < 0x398u) ? 1.0f : (float) gm_cos_polynomial(x2);
		return;
	}
	if (top >= 0x7F8u) { *out_sin = *out_cos = gm_quiet_nan(); return; }
	int n, sign;
	double x = gm_reduce(y, &n, &sign);
	int q = n + sign;
	double s = ((q & 3) == 1 || (q & 3) == 2) ? -1.0 : 1.0;
	double x2 = x * x;
	float from_sin = (float) gm_sin_polynomial(x * s, x2);
	double c = gm_cos_polynomial(x2);
	float from_cos = (float) ((q & 2) ? -c : c);
	/* sinf takes the sine polynomial in even quadrants, cosf in odd ones; the cosine polynomial of
	 * cosf belongs to quadrant n ^ 1 ... whose table choice is the same (n + sign) & 2 */
	*out_sin = (n & 1) ? from_cos : from_sin;
	*out_cos = (n & 1) ? from_sin : from_cos;
}

/* ---- log2f, powf (e_log2f.c, e_powf.c, their data; the -mfma builds) --------------------------- */

/* (1 / c, log2(c)) for 16 intervals of the mantissa (__log2f_data.tab == __powf_log2_data.tab) */
GM_TABLE double gm_log2_table[16][2] = {
	{0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
	{0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2}, {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
	{0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
	{0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
	{0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1p+0, 0x0p+0},
	{0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4}, {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
	{0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3}, {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
	{0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2}, {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2}};

/* 2^(i / 32) as bit patterns with i << 47 subtracted (__exp2f_data.tab) */
GM_TABLE uint64_t gm_exp2_table[32] = {
	0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
	0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
	0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
	0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
	0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
	0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
	0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
	0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};

GM_FN float gm_log2f(float x) {
	const double a0 = -0x1.712b6f70a7e4dp-2, a1 = 0x1.ecabf496832ep-2, a2 = -0x1.715479ffae3dep-1, a3 = 0x1.715475f35c8b8p+0;
	uint32_t ix = gm_bits(x);
	if (ix == 0x3F800000u) return 0.0f;
	if (ix - 0x00800000u >= 0x7F800000u - 0x00800000u) {
		if (ix * 2u == 0) return gm_float(0xFF800000u);
		if (ix == 0x7F800000u) return x;
		if ((ix & 0x80000000u) || ix * 2u >= 0xFF000000u) return gm_quiet_nan();
		ix = gm_bits(x * 0x1p23f);
		ix -= 23u << 23;
	}
	uint32_t tmp = ix - 0x3F330000u;
	uint32_t i = (tmp >> 19) & 15u;
	uint32_t top = tmp & 0xFF800000u;
	uint32_t iz = ix - top;
	int32_t k = (int32_t) tmp >> 23;
	double invc = gm_log2_table[i][0], logc = gm_log2_table[i][1];
	double z = (double) gm_float(iz);
	double r = GM_FMA(z, invc, -1.0);
	double y0 = logc + (double) k;
	double y = GM_FMA(a1, r, a2);
	double r2 = r * r;
	double p = GM_FMA(r, a3, y0);
	y = GM_FMA(a0, r2, y);
	return (float) GM_FMA(r2, y, p);
}

GM_FN int gm_checkint(uint32_t iy) {
	int e = (int) ((iy >> 23) & 0xFFu);
	if (e < 0x7F) return 0;
	if (e > 0x7F + 23) return 2;
	if (iy & ((1u << (0x7F + 23 - e)) - 1u)) return 0;
	if (iy & (1u << (0x7F + 23 - e))) return 1;
	return 2;
}

GM_FN float gm_powf(float x, float y) {
	const double a0 = 0x1.27616c9496e0bp-2, a1 = -0x1.71969a075c67ap-2, a2 = 0x1.ec70a6ca7baddp-2, a3 = -0x1.7154748bef6c8p-1, a4 = 0x1.71547652ab82bp+0;
	const double c0 = 0x1.c6af84b912394p-5, c1 = 0x1.ebfce50fac4f3p-3, c2 = 0x1.62e42ff0c52d6p-1;
	uint32_t sign_bias = 0;
	uint32_t ix = gm_bits(x), iy = gm_bits(y);
	int y_special = 2u * iy - 1u >= 2u * 0x7F800000u - 1u;
	if (ix - 0x00800000u >= 0x7F800000u - 0x00800000u || y_special) {
		if (y_special) {
			if (2u * iy == 0) return 1.0f;
			if (ix == 0x3F800000u) return 1.0f;
			if (2u * ix > 2u * 0x7F800000u || 2u * iy > 2u * 0x7F800000u) return x + y;
			if (2u * ix == 2u * 0x3F800000u) return 1.0f;
			if ((2u * ix < 2u * 0x3F800000u) == !(iy & 0x80000000u)) return 0.0f;
			return y * y;
		}
		if (2u * ix - 1u >= 2u * 0x7F800000u - 1u) {
			float x2 = x * x;
			if ((ix & 0x80000000u) && gm_checkint(iy) == 1) {
				x2 = -x2;
				sign_bias = 1;
			}
			if (2u * ix == 0 && (iy & 0x80000000u)) return sign_bias ? gm_float(0xFF800000u) : gm_float(0x7F800000u);
			return (iy & 0x80000000u) ? GM_DIVF(1.0f, x2) : x2;
		}
		if (ix & 0x80000000u) {
			int yint = gm_checkint(iy);
			if (yint == 0) return gm_quiet_nan();
			if (yint == 1) sign_bias = 1u << 16;
			ix &= 0x7FFFFFFFu;
		}
		if (ix < 0x00800000u) {
			ix = gm_bits(x * 0x1p23f);
			ix &= 0x7FFFFFFFu;
			ix -= 23u << 23;
		}
	}
	/* log2_inline */
	uint32_t tmp = ix - 0x3F330000u;
	uint32_t i = (tmp >> 19) & 15u;
	uint32_t top = tmp & 0xFF800000u;
	uint32_t iz = ix - top;
	int32_t k = (int32_t) top >> 23;
	double invc = gm_log2_table[i][0], logc = gm_log2_table[i][1];
	double z = (double) gm_float(iz);
	double r = GM_FMA(z, invc, -1.0);
	double y0 = logc + (double) k;
	double yy = GM_FMA(a0, r, a1);
	double p = GM_FMA(a2, r, a3);
	double r2 = r * r;
	double q = GM_FMA(r, a4, y0);
	double r4 = r2 * r2;
	q = GM_FMA(r2, p, q);
	double logx = GM_FMA(yy, r4, q);
	double ylogx = (double) y * logx;
	if (((gm_bits64(ylogx) >> 47) & 0xFFFFu) >= (0x405F800000000000ull >> 47)) {
		/* |y log2 x| >= 126 */
		if (ylogx > 0x1.fffffffd1d571p+6) return sign_bias ? gm_float(0xFF800000u) : gm_float(0x7F800000u);
		if (ylogx <= -150.0) return sign_bias ? -0.0f : 0.0f;
		if (ylogx < -149.0) return sign_bias ? gm_float(0x80000001u) : gm_float(0x00000001u);
	}
	/* exp2_inline */
	double kd = ylogx + 0x1.8p+47;
	uint64_t ki = gm_bits64(kd);
	kd -= 0x1.8p+47;
	double rr = ylogx - kd;
	uint64_t t = gm_exp2_table[ki & 31u];
	uint64_t ski = ki + sign_bias;
	t += ski << 47;
	double s = gm_double(t);
	double zz = GM_FMA(c0, rr, c1);
	double rr2 = rr * rr;
	double e = GM_FMA(rr, c2, 1.0);
	e = GM_FMA(zz, rr2, e);
	return (float) (e * s);
}

#endif
