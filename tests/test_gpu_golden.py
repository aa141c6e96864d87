"""The HIP shading pass against (a) the golden frames rendered by the reference's
own shader and (b) the CPU oracle, through the C-ABI."""
import os

import numpy as np
import pytest

import golden_cases
from helpers import DeviceBuffer, compare, oracle_render
from vulkan_renderer_amd import renderer

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# stated tolerance of BASELINE.json: RMSE <= 1e-4 on exposure-scaled linear radiance
RMSE_TOLERANCE = 1.0e-4


@pytest.fixture(scope="module")
def golden_dataset(tmp_path_factory):
    from vulkan_renderer_amd import synthetic
    return synthetic.write_dataset(str(tmp_path_factory.mktemp("golden_dataset")), **golden_cases.DATASET)


@pytest.fixture(scope="module")
def frames():
    return np.load(os.path.join(GOLDEN, "frames.npz"))


def render_case(case, dataset, arithmetic, width=golden_cases.WIDTH, height=golden_cases.HEIGHT, frames_in_flight=1):
    if isinstance(arithmetic, bool):
        arithmetic = "fast" if arithmetic else "libm"
    r = renderer.Renderer(arithmetic=arithmetic, frames_in_flight=frames_in_flight)
    golden_cases.apply_case(r, case, dataset, width, height)
    r.create_targets()
    r.create_pass()
    r.render_visibility()
    r.render()
    return r, r.read_radiance()


LINEAR_CASES = [c for c in golden_cases.FRAME_CASES if c.get("output_linear_rgb", True)]


@pytest.mark.parametrize("case", LINEAR_CASES, ids=[c["key"] for c in LINEAR_CASES])
def test_libm_mode_reproduces_the_frames_of_the_reference_shader_bit_for_bit(case, golden_dataset, frames):
    """The golden frames are outputs of the reference's own GLSL (compiled as C++ by
    oracle/ref_stubs, tests/golden/make_golden.py) with this image's C library behind atan / acos /
    sin / cos / pow.  The libm arithmetic mode of the kernels evaluates the same operations: every
    pixel of every variant - the fragile ones too (Hart's biquadratic warp, Arvo's projected
    solid angle sampler, the error displays) - must come out identical."""
    r, image = render_case(case, golden_dataset, "libm")
    r.close()
    golden = frames[case["key"]].astype(np.float32)
    stats = compare(image, golden)
    print(case["key"], stats)
    assert stats["nan"] == 0
    assert np.array_equal(image[..., :3].view(np.uint32), golden[..., :3].view(np.uint32)), stats


@pytest.mark.parametrize("case", LINEAR_CASES, ids=[c["key"] for c in LINEAR_CASES])
@pytest.mark.parametrize("fast_math", [False, True], ids=["exact", "fast"])
def test_golden_frames_of_the_reference_shader(case, fast_math, golden_dataset, frames):
    """The golden frames were computed with libm transcendentals; the polynomial and the fast
    mode use their own forms, so their comparison is by tolerance, not by bits."""
    r, image = render_case(case, golden_dataset, "fast" if fast_math else "exact")
    r.close()
    stats = compare(image, frames[case["key"]])
    print(case["key"], "fast" if fast_math else "exact", stats)
    assert stats["nan"] == 0
    if "biquadratic" in case.get("technique", ""):
        # one density sample of Hart's biquadratic warp is rounding noise (see
        # tests/test_oracle_properties.py); only the bit-exact comparison below applies
        assert abs(float(image[..., :3].mean()) - float(frames[case["key"]][..., :3].mean())) <= 0.05 * float(frames[case["key"]][..., :3].mean())
        return
    if case.get("error_display"):
        # The displayed quantity is the rounding-level error of the sampler itself, put
        # through a step function (colour bins); it changes with the last bit of atan.
        # (and, in exact mode, with the 0.85-ulp inversesqrt that replaces 1 / sqrt).  Against
        # the libm frames only the distribution can agree: a good part of the pixels
        # identical, the same mean.  (Bit-exactness against the oracle's polynomial mode is
        # the next test.)
        mine, theirs = image[..., :3].reshape(-1, 3), frames[case["key"]][..., :3].reshape(-1, 3)
        same = (mine == theirs).all(axis=-1).mean()
        assert same >= 0.3, same
        assert abs(float(mine.mean()) - float(theirs.mean())) <= 0.05 * max(float(theirs.mean()), 1e-6)
        return
    if "projected_solid_angle_arvo" in case.get("technique", ""):
        # Arvo's projected solid angle sampler is "substantially slower and less stable" in the
        # reference's own words (polygon_sampling_related_work.glsl:516-520): a few pixels hit the
        # shader's NaN guard in one arithmetic and not in the other, the rest agrees more loosely
        difference = np.abs(image[..., :3].astype(np.float64) - frames[case["key"]][..., :3]).max(axis=-1)
        guard = difference > 0.1
        assert guard.sum() <= 8, int(guard.sum())
        assert np.sqrt((difference[~guard] ** 2).mean()) <= 1.0e-3
        return
    assert stats["rmse"] <= RMSE_TOLERANCE, stats
    # single-pixel bound: a sample that lands next to a discontinuity of the estimator
    # (sector boundary, clipping) moves further under approximate reciprocals
    assert stats["max_abs"] <= (1.0e-2 if fast_math else 2.0e-3), stats


@pytest.mark.parametrize("case", LINEAR_CASES, ids=[c["key"] for c in LINEAR_CASES])
@pytest.mark.parametrize("arithmetic", ["libm", "exact"])
def test_ieee_modes_equal_their_oracle_bit_for_bit(case, arithmetic, golden_dataset):
    r, image = render_case(case, golden_dataset, arithmetic, 96, 64)
    cpu, _, _ = oracle_render(r, visibility=r.read_visibility(), math_mode=renderer.ORACLE_MATH_MODE[arithmetic])
    r.close()
    stats = compare(image, cpu)
    assert stats["bit_exact"], stats


def test_srgb_and_half_encodings(golden_dataset, frames):
    import oracle
    case = golden_cases.FRAME_CASES[0]
    r, image = render_case(case, golden_dataset, False)
    srgb = r.read_encoded(output_linear_rgb=False, frame_bits=0)
    low = r.read_encoded(output_linear_rgb=False, frame_bits=1)
    high = r.read_encoded(output_linear_rgb=True, frame_bits=2)
    r.close()
    # the sRGB frame of the reference shader (variant with OUTPUT_LINEAR_RGB=0), stored as UNORM8
    expected = (np.clip(frames["cfg1_srgb_encoded"], 0, 1) * 255.0 + 0.5).astype(np.uint8)
    assert np.abs(srgb[..., :3].astype(int) - expected[..., :3].astype(int)).max() <= 1
    assert (srgb[..., :3] != expected[..., :3]).mean() < 0.01
    assert np.array_equal(low, oracle.encode_half_bits(image, 1, False))
    assert np.array_equal(high, oracle.encode_half_bits(image, 2, True))
    # reassembling the two bytes gives the half-float image (reference main.c:1700-1710)
    halves = (high[..., :3].astype(np.uint16) << 8) | low[..., :3].astype(np.uint16)
    assert np.allclose(halves.view(np.float16).astype(np.float32), image[..., :3], rtol=1e-3, atol=1e-4)


def test_tiles_of_virtual_ranks_reassemble_bit_exactly(golden_dataset):
    """Multi-GPU layout on one GPU: render every rank's slab, gather, scatter back."""
    import ctypes as C
    case = golden_cases.FRAME_CASES[3]
    r, full = render_case(case, golden_dataset, False, 200, 120)  # not a multiple of the tile size
    lib = r.lib
    for tile_size, ranks in ((16, 2), (32, 3), (64, 8)):
        r.set_tiles(tile_size, 0, ranks)
        slab_pixels = r.slab_pixel_count(0)
        gathered = np.zeros((ranks, slab_pixels, 4), np.float32)
        dev = DeviceBuffer(slab_pixels * 16)
        for rank in range(ranks):
            r.set_tiles(tile_size, rank, ranks)
            dev.zero()
            r.render(dev.ptr.value)
            r.sync()
            gathered[rank] = dev.download((slab_pixels, 4), np.float32)
        # host-side scatter using the library's own slab description
        out = np.zeros_like(full)
        for rank in range(ranks):
            r.set_tiles(tile_size, rank, ranks)
            xy = np.zeros((slab_pixels, 2), np.uint32)
            slots = lib.get_slab_pixel_coordinates(C.byref(r.app), rank, xy.ctypes.data, slab_pixels)
            valid = xy[:slots, 0] != 0xFFFFFFFF
            out[xy[:slots][valid, 1], xy[:slots][valid, 0]] = gathered[rank, :slots][valid]
        assert np.array_equal(out.view(np.uint32), full.view(np.uint32)), (tile_size, ranks)
        # device-side scatter
        all_slabs = DeviceBuffer(gathered.nbytes)
        all_slabs.upload(gathered)
        frame = DeviceBuffer(120 * 200 * 16)
        r.set_tiles(tile_size, 0, ranks)
        r.assemble(all_slabs.ptr.value, frame.ptr.value)
        r.sync()
        assert np.array_equal(frame.download((120, 200, 4), np.float32).view(np.uint32), full.view(np.uint32)), (tile_size, ranks)
        # the same exchange in the pass's output format (RGBA8): encode every slab, gather, scatter
        encoded_full = r.read_encoded(False, 0)
        gathered8 = np.zeros((ranks, slab_pixels), np.uint32)
        slab8 = DeviceBuffer(slab_pixels * 4)
        for rank in range(ranks):
            r.set_tiles(tile_size, rank, ranks)
            dev.upload(gathered[rank])
            r.encode_slab(dev.ptr.value, slab8.ptr.value, slab_pixels)
            r.sync()
            gathered8[rank] = slab8.download((slab_pixels,), np.uint32)
        all_slabs8 = DeviceBuffer(gathered8.nbytes)
        all_slabs8.upload(gathered8)
        frame8 = DeviceBuffer(120 * 200 * 4)
        r.set_tiles(tile_size, 0, ranks)
        r.assemble_encoded(all_slabs8.ptr.value, frame8.ptr.value)
        r.sync()
        assert np.array_equal(frame8.download((120, 200, 4), np.uint8), encoded_full), (tile_size, ranks)
        # and as packed RGB8 (what bench.py exchanges by default): the same bytes without the constant alpha
        gathered3 = np.zeros((ranks, slab_pixels, 3), np.uint8)
        slab3 = DeviceBuffer(slab_pixels * 3)
        for rank in range(ranks):
            r.set_tiles(tile_size, rank, ranks)
            dev.upload(gathered[rank])
            r.encode_slab_rgb8(dev.ptr.value, slab3.ptr.value, slab_pixels)
            r.sync()
            gathered3[rank] = slab3.download((slab_pixels, 3), np.uint8)
            assert np.array_equal(gathered3[rank], gathered8[rank].view(np.uint8).reshape(-1, 4)[:, :3])
        all_slabs3 = DeviceBuffer(gathered3.nbytes)
        all_slabs3.upload(gathered3)
        frame8.zero()
        r.set_tiles(tile_size, 0, ranks)
        r.assemble_rgb8(all_slabs3.ptr.value, frame8.ptr.value)
        r.sync()
        assert np.array_equal(frame8.download((120, 200, 4), np.uint8), encoded_full), (tile_size, ranks)
        for b in (dev, all_slabs, frame, slab8, all_slabs8, frame8, slab3, all_slabs3):
            b.free()
    r.set_tiles(16, 0, 1)
    r.close()


def test_missing_variant_and_bad_settings_fail_loudly(golden_dataset):
    import ctypes as C
    r = renderer.Renderer()
    golden_cases.apply_case(r, golden_cases.FRAME_CASES[0], golden_dataset)
    r.create_targets()
    r.app.render_settings.polygon_sampling_technique = 13  # sample_polygon_count: not a technique
    assert r.lib.create_shading_pass(C.byref(r.app.shading_pass), C.byref(r.app)) == 1
    r.app.render_settings.polygon_sampling_technique = 1  # area sampling exists only for the diffuse-only strategy
    r.app.render_settings.sampling_strategies = 1
    assert r.lib.create_shading_pass(C.byref(r.app.shading_pass), C.byref(r.app)) == 1
    r.app.render_settings.polygon_sampling_technique = 4  # solid angle cannot drive LTC strategies
    r.app.render_settings.sampling_strategies = 3
    assert r.lib.create_shading_pass(C.byref(r.app.shading_pass), C.byref(r.app)) == 1
    r.app.render_settings.polygon_sampling_technique = 11
    r.create_pass()
    r.app.render_settings.sampling_strategies = 0  # changing the variant without recreating the pass
    assert r.lib.render_shading_pass(C.byref(r.app), None) == 1
    r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("frames_in_flight", [1, 2, 3, 4])
def test_frames_in_flight_keep_their_own_constants(golden_dataset, frames_in_flight):
    """The host records frames without waiting for the device; every frame must see the
    constants that write_constants produced for it (ring of constant buffers), and a
    frame with unchanged constants must reuse them (no stale or future data)."""
    case = golden_cases.FRAME_CASES[3]
    r, _ = render_case(case, golden_dataset, False, 200, 120, frames_in_flight)
    exposures = [0.5, 0.5, 2.0, 3.0, 3.0, 4.0, 5.0, 6.0, 6.0, 7.0]
    expected = []
    for e in sorted(set(exposures)):
        r.app.render_settings.exposure_factor = e
        r.render()
        r.sync()
        expected.append((e, r.read_radiance()))
    expected = dict(expected)
    buffers = [DeviceBuffer(120 * 200 * 16) for _ in exposures]
    for e, b in zip(exposures, buffers):  # no synchronisation between these launches
        r.app.render_settings.exposure_factor = e
        r.render(b.ptr.value)
    r.sync()  # wait_for_device covers the frame streams
    # frames that share one target arrive in order: the last one wins
    for e in (1.0, 9.0, 2.5):
        r.app.render_settings.exposure_factor = e
        r.render()
    last = r.read_radiance()
    r.app.render_settings.exposure_factor = 2.5
    r.render()
    assert np.array_equal(last.view(np.uint32), r.read_radiance().view(np.uint32))
    for e, b in zip(exposures, buffers):
        got = b.download((120, 200, 4), np.float32)
        assert np.array_equal(got.view(np.uint32), expected[e].view(np.uint32)), e
        b.free()
    r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("frames_in_flight", [1, 2])
def test_render_encoded_equals_render_then_encode(golden_dataset, frames_in_flight):
    """render_shading_pass_encoded(): the frame and, on the same stream, its packed RGB8 form"""
    case = golden_cases.FRAME_CASES[3]
    r, full = render_case(case, golden_dataset, False, 200, 120, frames_in_flight)
    radiance, packed, expected = DeviceBuffer(200 * 120 * 16), DeviceBuffer(200 * 120 * 3), DeviceBuffer(200 * 120 * 3)
    next_stream = r.next_frame_stream()
    assert (next_stream in [int(r.app.device.frame_streams[i]) for i in range(len(r.app.device.frame_streams)) if r.app.device.frame_streams[i]]) == (frames_in_flight > 1)
    r.render_encoded(radiance.ptr.value, packed.ptr.value)
    r.sync()
    assert np.array_equal(radiance.download((120, 200, 4), np.float32).view(np.uint32), full.view(np.uint32))
    r.encode_slab_rgb8(radiance.ptr.value, expected.ptr.value, 200 * 120)
    r.sync()
    rgb = packed.download((120, 200, 3), np.uint8)
    assert np.array_equal(rgb, expected.download((120, 200, 3), np.uint8))
    assert np.array_equal(rgb, r.read_encoded(False, 0)[..., :3])
    for b in (radiance, packed, expected):
        b.free()
    r.close()
