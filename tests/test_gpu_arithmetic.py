"""The arithmetic contract of the exact mode (vulkan_renderer_amd/csrc/device_math.h): division and
square root are the correctly rounded IEEE results - the ones the CPU oracle computes with `/` and
sqrtf - for every operand outside the last decades of the exponent range, and for zeros, infinities
and NaNs.  The primitives are evaluated on the device through the C-ABI
(evaluate_device_arithmetic) and compared bit for bit with numpy's float32 arithmetic."""
import ctypes as C

import numpy as np
import pytest

from vulkan_renderer_amd import renderer

pytestmark = pytest.mark.gpu


def evaluate(r, operation, a, b=None):
    a = np.ascontiguousarray(a, np.float32)
    out = np.zeros_like(a)
    fp = C.POINTER(C.c_float)
    b_pointer = None
    if b is not None:
        b = np.ascontiguousarray(b, np.float32)
        b_pointer = b.ctypes.data_as(fp)
    assert r.lib.evaluate_device_arithmetic(C.byref(r.app.device), operation, a.ctypes.data_as(fp), b_pointer, out.ctypes.data_as(fp), a.size) == 0
    return out


def same_bits(x, y):
    x, y = np.asarray(x, np.float32), np.asarray(y, np.float32)
    both_nan = np.isnan(x) & np.isnan(y)
    return (x.view(np.uint32) == y.view(np.uint32)) | both_nan


@pytest.fixture(scope="module")
def device():
    r = renderer.Renderer()
    yield r
    r.close()


def random_floats(rng, count, low_exponent, high_exponent):
    """Signed floats with uniformly random mantissas and exponents in [low, high)"""
    mantissa = rng.integers(0, 1 << 23, count, dtype=np.uint32)
    exponent = rng.integers(low_exponent + 127, high_exponent + 127, count, dtype=np.uint32)
    sign = rng.integers(0, 2, count, dtype=np.uint32)
    return ((sign << 31) | (exponent << 23) | mantissa).view(np.float32)


def test_division_is_the_ieee_quotient_in_the_range_the_kernels_work_in(device):
    rng = np.random.default_rng(7)
    n = 1 << 21
    with np.errstate(all="ignore"):
        # the whole range the shading arithmetic lives in, and far beyond it
        for low, high in ((-20, 20), (-40, 40), (-60, 36)):
            a, b = random_floats(rng, n, low, high), random_floats(rng, n, low, high)
            q = evaluate(device, 0, a, b)
            assert same_bits(q, a / b).all(), (low, high, int((~same_bits(q, a / b)).sum()))
        # mantissa patterns that make rounding hard: quotients next to a tie
        b = random_floats(rng, n, -10, 10)
        q_exact = random_floats(rng, n, -10, 10)
        a = (q_exact.astype(np.float64) * b.astype(np.float64)).astype(np.float32)
        assert same_bits(evaluate(device, 0, a, b), a / b).all()
        # the compiler's own division agrees with numpy everywhere (sanity of the comparison)
        a, b = random_floats(rng, n, -126, 127), random_floats(rng, n, -126, 127)
        assert same_bits(evaluate(device, 3, a, b), a / b).all()


def test_division_with_one_correction_equals_the_ieee_quotient_for_whole_rows_of_significand_pairs(device):
    """divide() corrects its quotient once, the compiler's a / b twice; that once is enough was found by trying all
    2^46 pairs of significands (profiles/tools/division_chains.hip, profiles/r03w/division_chains.txt).  A slice
    of that search stays in the suite: 4096 divisors spread over the significands (and the ones around the
    first counterexample of the chain that does NOT work), each against every one of the 2^23 dividends, at
    several places of the exponent window."""
    out = (C.c_uint64 * 2)()
    pairs = 0
    for first, count, stride, dividend_exponent, divisor_exponent in (
            (0x000000, 4096, 2048, 127, 127), (0x000001, 4096, 2047, 128, 127), (0x57EA09 - 64, 128, 1, 127, 127),
            (0x7FFFFF - 255, 256, 1, 127, 127), (0x000000, 256, 1, 127, 127),
            (0x000123, 512, 16381, 127 + 40, 127 - 30), (0x000456, 512, 16381, 127 - 60, 127 + 20), (0x000789, 512, 16381, 127 - 33, 127 - 80)):
        assert device.lib.compare_device_division(C.byref(device.app.device), first, count, stride, dividend_exponent, divisor_exponent, out) == 0
        assert out[0] == 0, (first, count, stride, dividend_exponent, divisor_exponent, int(out[0]), hex(int(out[1])))
        pairs += count << 23
    assert pairs > 8.0e10


def test_division_keeps_the_ieee_results_for_zeros_infinities_and_nans(device):
    specials = np.array([0.0, -0.0, 1.0, -1.0, 3.5, -2.25e-3, np.inf, -np.inf, np.nan, 1.0e10, -1.0e-10], np.float32)
    a, b = [x.ravel() for x in np.meshgrid(specials, specials)]
    with np.errstate(all="ignore"):
        expected = a / b
    q = evaluate(device, 0, a, b)
    assert same_bits(q, expected).all(), list(zip(a[~same_bits(q, expected)], b[~same_bits(q, expected)], q[~same_bits(q, expected)]))


def test_where_the_short_division_leaves_the_ieee_quotient(device):
    """Documents the edge of the contract: without the operand rescaling of v_div_scale_f32 the
    quotient may be off once an operand or the quotient sits in the last decades of the exponent
    range.  The test pins that the deviation stays out of [2^-100, 2^96]."""
    rng = np.random.default_rng(11)
    n = 1 << 20
    a, b = random_floats(rng, n, -126, 127), random_floats(rng, n, -126, 127)
    with np.errstate(all="ignore"):
        expected = a / b
    differs = ~same_bits(evaluate(device, 0, a, b), expected)
    exponent = lambda x: ((x.view(np.uint32) >> 23) & 0xFF).astype(np.int32) - 127
    inside = (np.abs(exponent(a)) <= 96) & (np.abs(exponent(b)) <= 96) & np.isfinite(expected) & (expected != 0) & (np.abs(exponent(expected)) <= 96)
    assert not (differs & inside).any()


def test_square_root_is_correctly_rounded(device):
    rng = np.random.default_rng(3)
    n = 1 << 21
    x = np.abs(random_floats(rng, n, -100, 100))
    assert same_bits(evaluate(device, 1, x), np.sqrt(x)).all()
    # perfect squares and their neighbours
    k = rng.integers(1, 4096, 4096).astype(np.float32)
    for delta in (0, 1, -1):
        y = ((k * k).view(np.uint32).astype(np.int64) + delta).astype(np.uint32).view(np.float32)
        assert same_bits(evaluate(device, 1, y), np.sqrt(y)).all()
    specials = np.array([0.0, -0.0, np.inf, np.nan, -1.0, 1.0, 4.0], np.float32)
    with np.errstate(all="ignore"):
        assert same_bits(evaluate(device, 1, specials), np.sqrt(specials)).all()


def test_square_root_and_inverse_square_root_are_the_ieee_results_for_every_float_of_the_working_range(device):
    """Not a sample: every float with an exponent in [-100, 100) - 1.7 billion bit patterns - goes through the
    kernels' square_root() and inversesqrt (divide(1, square_root)) and through the compiler's correctly
    rounded sqrtf / 1 / sqrtf on the device (compare_device_arithmetic of the C-ABI); plus zeros, infinity,
    NaN and negative numbers."""
    out = (C.c_uint64 * 2)()
    first, last = (127 - 100) << 23, (127 + 100) << 23
    for mine, theirs in ((1, 4), (12, 16)):
        assert device.lib.compare_device_arithmetic(C.byref(device.app.device), mine, theirs, first, last - first, out) == 0
        assert out[0] == 0, (mine, theirs, int(out[0]), hex(int(out[1])))
        for special in (0x00000000, 0x80000000, 0x7F800000, 0x7FC00000, 0xBF800000, 0xFF800000):
            assert device.lib.compare_device_arithmetic(C.byref(device.app.device), mine, theirs, special, 1, out) == 0
            assert out[0] == 0, (mine, theirs, hex(special))


@pytest.mark.parametrize("extra_lds", [0, 20480, 65536])
def test_hardware_wave_slots_are_unique_among_resident_waves(extra_lds):
    """The two-technique kernels keep one polygon table in device memory, in the region of the hardware slot the wave runs in
    (XCD, shader engine, shader array, CU, SIMD, wave slot: csrc/shading_kernel.h hardware_wave_slot).  Two waves that are resident
    at the same time must never compute the same slot - within a kernel, and across two kernels that run at once."""
    import ctypes as C
    r = renderer.Renderer()
    out = (C.c_uint64 * 2)()
    assert r.lib.check_hardware_wave_slots(C.byref(r.app.device), 200000, extra_lds, out) == 0
    shared, used = int(out[0]), int(out[1])
    r.close()
    print(extra_lds, shared, used)
    assert shared == 0, (shared, used)
    # every SIMD of the chip took part: at least one slot per SIMD of 256 CUs, at most kWaveSlots
    assert 1024 <= used <= (1 << 17), used
