/* TEST INFRASTRUCTURE - not part of the product.
 *
 * Bulk evaluation of the float functions that the oracle's libm math mode calls, (a) by the C
 * library of this machine and (b) by the restatement of glibc 2.35 that the kernels' "libm"
 * arithmetic mode evaluates (vulkan_renderer_amd/csrc/glibc_math.h, compiled here for the host).
 * tests/test_glibc_math.py compares (a) with (b) on the CPU and (a) with the GPU's results
 * (evaluate_device_arithmetic of the C-ABI). */
#include "../vulkan_renderer_amd/csrc/glibc_math.h"
#include <stddef.h>

/* operation codes shared with evaluate_device_arithmetic (include/vkr_shading_pass.h) */
enum { op_atan = 5, op_acos = 6, op_sin = 7, op_cos = 8, op_log2 = 9, op_pow = 10, op_atan2 = 11, op_inverse_sqrt = 12,
	/* the arctangent with its argument range looked up in a table (what the kernels run, with the table in LDS) */
	op_atan_rows = 13 };

static gm_atan_row_t g_atan_rows[GM_ATAN_ROW_COUNT];
static int g_atan_rows_filled = 0;
static const gm_atan_row_t* atan_rows(void) {
	if (!g_atan_rows_filled) {
		for (uint32_t i = 0; i != GM_ATAN_ROW_COUNT; ++i) g_atan_rows[i] = gm_atan_row(i);
		g_atan_rows_filled = 1;
	}
	return g_atan_rows;
}

static float by_libm(int op, float a, float b) {
	switch (op) {
	case op_atan: case op_atan_rows: return atanf(a);
	case op_acos: return acosf(a);
	case op_sin: return sinf(a);
	case op_cos: return cosf(a);
	case op_log2: return log2f(a);
	case op_pow: return powf(a, b);
	case op_atan2: return atan2f(a, b);
	default: return 1.0f / sqrtf(a);
	}
}

static float by_port(int op, float a, float b) {
	switch (op) {
	case op_atan: return gm_atanf(a);
	case op_atan_rows: return gm_atanf_rows(a, g_atan_rows);
	case op_acos: return gm_acosf(a);
	case op_sin: return gm_sinf(a);
	case op_cos: return gm_cosf(a);
	case op_log2: return gm_log2f(a);
	case op_pow: return gm_powf(a, b);
	case op_atan2: return gm_atan2f(a, b);
	default: return 1.0f / sqrtf(a);
	}
}

/* out[i] = f(a[i], b[i]) (b may be NULL for one-argument functions) */
void oracle_libm_evaluate(int op, int use_port, const float* a, const float* b, float* out, size_t count) {
	(void) atan_rows();
#pragma omp parallel for schedule(static)
	for (ptrdiff_t i = 0; i < (ptrdiff_t) count; ++i)
		out[i] = use_port ? by_port(op, a[i], b ? b[i] : 0.0f) : by_libm(op, a[i], b ? b[i] : 0.0f);
}

/* Number of arguments first_bits + i * stride, i < count, for which port and C library differ
 * (NaNs compare equal); *first_mismatch receives the bit pattern of one of them */
size_t oracle_libm_count_mismatches(int op, uint32_t first_bits, uint32_t stride, size_t count, float second_argument, uint32_t* first_mismatch) {
	size_t mismatches = 0;
	(void) atan_rows();
#pragma omp parallel for schedule(static) reduction(+ : mismatches)
	for (ptrdiff_t i = 0; i < (ptrdiff_t) count; ++i) {
		uint32_t bits = first_bits + (uint32_t) i * stride;
		float x = gm_float(bits);
		float ours = by_port(op, x, second_argument), theirs = by_libm(op, x, second_argument);
		if (!((ours != ours && theirs != theirs) || gm_bits(ours) == gm_bits(theirs))) {
			++mismatches;
#pragma omp critical
			*first_mismatch = bits;
		}
	}
	return mismatches;
}
