/* TEST INFRASTRUCTURE - not part of the product.
 *
 * Compares the restatement of glibc's float functions that the "libm" arithmetic mode of the HIP
 * kernels evaluates (vulkan_renderer_amd/csrc/glibc_math.h) with the C library of this machine:
 * every one of the 2^32 float arguments for the one-argument functions, a few billion random and
 * all special pairs for atan2f and powf.  NaN results compare equal to each other whatever their
 * payload.
 *
 *   gcc -O2 -ffp-contract=off -mfma -fopenmp oracle/tools/check_glibc_math.c -o /tmp/check_glibc_math -lm
 *   /tmp/check_glibc_math [stride] [pairs_in_millions]
 *
 * stride 1 (default) is exhaustive (about a minute per function on 8 cores); the CPU test suite
 * runs the same comparison through liboracle.so on a subset (tests/test_glibc_math.py). */
#include "../../vulkan_renderer_amd/csrc/glibc_math.h"
#include <stdio.h>
#include <stdlib.h>

static int same(float a, float b) {
	if (a != a && b != b) return 1;
	return gm_bits(a) == gm_bits(b);
}

typedef float (*unary_t)(float);

static uint64_t check_unary(const char* name, unary_t ours, unary_t theirs, uint32_t stride) {
	uint64_t mismatches = 0;
	uint32_t first_bad = 0;
	int have_bad = 0;
#pragma omp parallel for schedule(static) reduction(+ : mismatches)
	for (int64_t block = 0; block < 65536; ++block) {
		for (uint32_t low = 0; low < 65536u; low += stride) {
			uint32_t bits = ((uint32_t) block << 16) | low;
			float x = gm_float(bits);
			if (!same(ours(x), theirs(x))) {
				++mismatches;
#pragma omp critical
				if (!have_bad) { have_bad = 1; first_bad = bits; }
			}
		}
	}
	printf("%-8s %llu mismatches over %llu arguments", name, (unsigned long long) mismatches, (unsigned long long) (65536ull * (65536u / stride)));
	if (have_bad) {
		float x = gm_float(first_bad);
		printf("  (e.g. x = %a (0x%08x): ours 0x%08x, libm 0x%08x)", x, first_bad, gm_bits(ours(x)), gm_bits(theirs(x)));
	}
	printf("\n");
	return mismatches;
}

static uint64_t xorshift(uint64_t* s) {
	uint64_t x = *s;
	x ^= x << 13; x ^= x >> 7; x ^= x << 17;
	return *s = x;
}

typedef float (*binary_t)(float, float);

static const uint32_t k_special[] = {
	0x00000000u, 0x80000000u, 0x00000001u, 0x80000001u, 0x007FFFFFu, 0x00800000u, 0x80800000u, 0x3F800000u, 0xBF800000u,
	0x3F000000u, 0x40000000u, 0xC0000000u, 0x40400000u, 0xC0400000u, 0x3EAAAAABu, 0x4019999Au, 0x3ED55555u, 0x41200000u,
	0x7F7FFFFFu, 0xFF7FFFFFu, 0x7F800000u, 0xFF800000u, 0x7FC00000u, 0xFFC00000u, 0x4B800000u, 0x4B800001u, 0xCB800001u,
	0x3F7FFFFFu, 0x3F800001u, 0x42FC0000u, 0xC2FC0000u, 0x43160000u, 0xC3160000u, 0x1E3CE508u, 0x5E800000u};

static uint64_t check_binary(const char* name, binary_t ours, binary_t theirs, uint64_t pairs, int positive_first) {
	uint64_t mismatches = 0;
	uint32_t bad_a = 0, bad_b = 0;
	int have_bad = 0;
	const int special_count = (int) (sizeof(k_special) / sizeof(k_special[0]));
	for (int i = 0; i != special_count; ++i)
		for (int j = 0; j != special_count; ++j) {
			float a = gm_float(k_special[i]), b = gm_float(k_special[j]);
			if (!same(ours(a, b), theirs(a, b))) {
				++mismatches;
				if (!have_bad) { have_bad = 1; bad_a = k_special[i]; bad_b = k_special[j]; }
			}
		}
#pragma omp parallel reduction(+ : mismatches)
	{
		uint64_t state = 0x9E3779B97F4A7C15ull;
#ifdef _OPENMP
		extern int omp_get_thread_num(void);
		state ^= 0xD1B54A32D192ED03ull * (uint64_t) (omp_get_thread_num() + 1);
#endif
#pragma omp for schedule(static)
		for (int64_t i = 0; i < (int64_t) pairs; ++i) {
			uint64_t r = xorshift(&state);
			uint32_t ba = (uint32_t) r, bb = (uint32_t) (r >> 32);
			/* every third pair: moderate magnitudes, where the functions do real work */
			if (i % 3 == 0) {
				ba = (ba & 0x807FFFFFu) | ((110u + ((ba >> 23) & 31u)) << 23);
				bb = (bb & 0x807FFFFFu) | ((118u + ((bb >> 23) & 15u)) << 23);
			}
			if (positive_first && (i & 1)) ba &= 0x7FFFFFFFu;
			float a = gm_float(ba), b = gm_float(bb);
			if (!same(ours(a, b), theirs(a, b))) {
				++mismatches;
#pragma omp critical
				if (!have_bad) { have_bad = 1; bad_a = ba; bad_b = bb; }
			}
		}
	}
	printf("%-8s %llu mismatches over %llu pairs", name, (unsigned long long) mismatches, (unsigned long long) pairs + (uint64_t) special_count * special_count);
	if (have_bad) {
		float a = gm_float(bad_a), b = gm_float(bad_b);
		printf("  (e.g. (%a, %a) = (0x%08x, 0x%08x): ours 0x%08x, libm 0x%08x)", a, b, bad_a, bad_b, gm_bits(ours(a, b)), gm_bits(theirs(a, b)));
	}
	printf("\n");
	return mismatches;
}

static gm_atan_row_t g_rows[GM_ATAN_ROW_COUNT];
static float ours_atan_rows(float x) { return gm_atanf_rows(x, g_rows); }
static float ours_sin_of_sincos(float x) { float s, c; gm_sincosf(x, &s, &c); return s; }
static float ours_cos_of_sincos(float x) { float s, c; gm_sincosf(x, &s, &c); return c; }
static float ours_pow_third(float x) { return gm_powf(x, 1.0f / 3.0f); }
static float libm_pow_third(float x) { return powf(x, 1.0f / 3.0f); }
static float ours_pow_gamma(float x) { return gm_powf(x, 2.4f); }
static float libm_pow_gamma(float x) { return powf(x, 2.4f); }
static float ours_pow_inverse_gamma(float x) { return gm_powf(x, 1.0f / 2.4f); }
static float libm_pow_inverse_gamma(float x) { return powf(x, 1.0f / 2.4f); }

int main(int argc, char** argv) {
	uint32_t stride = argc > 1 ? (uint32_t) strtoul(argv[1], NULL, 10) : 1u;
	uint64_t pairs = (argc > 2 ? strtoull(argv[2], NULL, 10) : 2000ull) * 1000000ull;
	if (stride == 0 || 65536u % stride != 0) stride = 1;
	uint64_t bad = 0;
	for (uint32_t i = 0; i != GM_ATAN_ROW_COUNT; ++i) g_rows[i] = gm_atan_row(i);
	bad += check_unary("atanf", gm_atanf, atanf, stride);
	bad += check_unary("atanf.t", ours_atan_rows, atanf, stride);
	bad += check_unary("acosf", gm_acosf, acosf, stride);
	bad += check_unary("sinf", gm_sinf, sinf, stride);
	bad += check_unary("cosf", gm_cosf, cosf, stride);
	bad += check_unary("sincos.s", ours_sin_of_sincos, sinf, stride);
	bad += check_unary("sincos.c", ours_cos_of_sincos, cosf, stride);
	bad += check_unary("log2f", gm_log2f, log2f, stride);
	bad += check_unary("pow 1/3", ours_pow_third, libm_pow_third, stride);
	bad += check_unary("pow 2.4", ours_pow_gamma, libm_pow_gamma, stride);
	bad += check_unary("pow1/2.4", ours_pow_inverse_gamma, libm_pow_inverse_gamma, stride);
	bad += check_binary("atan2f", gm_atan2f, atan2f, pairs, 0);
	bad += check_binary("powf", gm_powf, powf, pairs, 1);
	printf(bad ? "MISMATCH\n" : "all equal\n");
	return bad != 0;
}
