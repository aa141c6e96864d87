"""The BVH builders (HIP kernels: binned SAH, LBVH; host: binned SAH), the two traversal
layouts (binary threaded, four-wide with an LDS stack) and the multi-GPU slab exchange
(RCCL behind the C-ABI, include/vkr_slab_exchange.h) on a real MI355X.

Shadow rays are any-hit queries: whatever tree answers them, the frame must be the oracle's
frame bit for bit (reference contract: ray-query semantics, shading_pass.frag.glsl:120-138)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import golden_cases
from helpers import DeviceBuffer, compare, oracle_render
from vulkan_renderer_amd import experiments, renderer, synthetic

pytestmark = pytest.mark.gpu

BUILDERS = ["sah_device", "lbvh_device", "sah_host"]


def render_config(dataset, config, width, height, builder="sah_device", binary_traversal=False, frames_in_flight=1, arithmetic="libm", **overrides):
    r = renderer.Renderer(binary_traversal=binary_traversal, frames_in_flight=frames_in_flight, arithmetic=arithmetic)
    renderer.setup_config(r, config, dataset, width=width, height=height, acceleration_structure=builder, **overrides)
    r.create_targets()
    r.create_pass()
    r.render_visibility()
    r.render()
    return r, r.read_radiance()


@pytest.mark.parametrize("binary_traversal", [False, True], ids=["wide", "binary"])
@pytest.mark.parametrize("builder", BUILDERS)
def test_every_builder_and_both_trees_give_the_oracle_frame(dataset, builder, binary_traversal):
    r, image = render_config(dataset, 3, 256, 144, builder, binary_traversal)
    structure = r.app.scene.acceleration_structure
    assert structure.builder == renderer.BVH_BUILDER[builder] and structure.build_milliseconds > 0.0
    assert structure.wide_nodes and 0 < structure.wide_node_count < structure.node_count
    assert 3 <= structure.wide_stack_need <= 128
    visibility = r.read_visibility()
    rays = r.last_ray_count()
    cpu, inputs, bvh = oracle_render(r, visibility=visibility, math_mode=renderer.ORACLE_MATH_MODE[r.arithmetic])
    stats = compare(image, cpu)
    # the same rays through both layouts: the wide tree must be a collapse of the binary one
    wide, binary = r.traversal_statistics(True), r.traversal_statistics(False)
    stack_need = int(structure.wide_stack_need)
    r.close()
    assert stats["bit_exact"], (builder, binary_traversal, stats)
    assert rays > 0 and wide["rays"] == binary["rays"] == rays
    assert wide["blocked_rays"] == binary["blocked_rays"]
    assert wide["node_visits"] < binary["node_visits"]
    assert wide["deepest_stack"] <= stack_need


def test_rays_whose_stack_leaves_lds_give_the_same_frame(dataset, monkeypatch):
    """trace_shadow_rays_wide keeps 16 stack entries per lane in LDS and spills deeper ones to global
    memory on a slow path that ordinary scenes hardly reach.  With only four entries in LDS most
    rays go through it; the frame must stay the oracle's, bit for bit."""
    r, image = render_config(dataset, 3, 256, 144)
    deepest = r.traversal_statistics(True)["deepest_stack"]
    visibility = r.read_visibility()
    cpu, _, _ = oracle_render(r, visibility=visibility, math_mode=renderer.ORACLE_MATH_MODE[r.arithmetic])
    r.close()
    monkeypatch.setenv("VKR_WIDE_STACK_LDS", "4")
    r, spilled = render_config(dataset, 3, 256, 144)
    r.close()
    # (the walk pushes all hit children before it takes one off again: one entry more than the statistics' scheme)
    assert deepest + 1 > 4, "the scene is too shallow to leave four LDS entries"
    assert np.array_equal(image.view(np.uint32), cpu.view(np.uint32))
    assert np.array_equal(spilled.view(np.uint32), image.view(np.uint32))


@pytest.mark.parametrize("builder", BUILDERS)
def test_primary_visibility_does_not_depend_on_the_builder(dataset, builder):
    import oracle
    r = renderer.Renderer()
    renderer.setup_config(r, 2, dataset, width=320, height=180, acceleration_structure=builder)
    r.create_targets()
    r.create_pass()
    r.render_visibility()
    gpu = r.read_visibility()
    inputs = r.host_inputs()
    bvh = oracle.Bvh(inputs["quantized_positions"], inputs["dequantization_factor"], inputs["dequantization_summand"])
    cam = r.app.scene_specification.camera
    cpu = oracle.primary_visibility(inputs["constants"], bvh, 320, 180, cam.near, cam.far)
    r.close()
    assert np.array_equal(gpu, cpu), "%d pixels differ" % int((gpu != cpu).sum())


def test_device_sah_tree_is_as_good_as_the_host_tree_and_wide_visits_are_few(big_dataset):
    """Same algorithm, same bins: the trees may differ where triangles tie, not in quality.  And
    the point of the wide layout: a handful of dependent fetches per shadow ray."""
    per_ray = {}
    for builder in ("sah_device", "sah_host", "lbvh_device"):
        r, _ = render_config(big_dataset, 3, 960, 540, builder)
        binary, wide = r.traversal_statistics(False), r.traversal_statistics(True)
        per_ray[builder] = (binary["node_visits"] / binary["rays"], wide["node_visits"] / wide["rays"], wide["boxes_tested"] / wide["rays"],
                            r.app.scene.acceleration_structure.build_milliseconds, r.app.scene.acceleration_structure.wide_stack_need)
        r.close()
    print(per_ray)
    assert abs(per_ray["sah_device"][0] - per_ray["sah_host"][0]) <= 0.05 * per_ray["sah_host"][0], per_ray
    assert per_ray["sah_device"][1] <= 8.0, per_ray
    assert per_ray["lbvh_device"][0] > per_ray["sah_device"][0], per_ray


def test_triangles_in_a_random_order_give_a_tree_of_the_same_quality_and_the_oracle_frame(tmp_path):
    """The device build combines the contributions of a workgroup's 256 consecutive triangles in LDS before
    its device-scope atomics - a 16-slot table keyed by the open node (csrc/lbvh_build.hip).  With the
    triangles along a Morton curve, as the datasets store them, a workgroup meets a handful of nodes; in a
    random order it meets more than sixteen from the first levels on, and most triangles take the direct
    path.  Counts add up and bounds are minima / maxima either way: the tree has to be as good, the walk
    as short, the frame the oracle's."""
    per_ray = {}
    frames = {}
    for name, shuffle_seed in (("morton", None), ("random", 99)):
        dataset = synthetic.write_dataset(str(tmp_path / name), grid=96, box_count=32, seed=77, ltc_resolution=16, fresnel_count=8, shuffle_seed=shuffle_seed)
        for builder in ("sah_device", "sah_host"):
            r, image = render_config(dataset, 3, 320, 180, builder)
            wide = r.traversal_statistics(True)
            per_ray[name, builder] = wide["node_visits"] / wide["rays"]
            if builder == "sah_device":
                cpu, _, _ = oracle_render(r, visibility=r.read_visibility(), math_mode=renderer.ORACLE_MATH_MODE[r.arithmetic])
                assert compare(image, cpu)["bit_exact"], name
                frames[name] = image
            r.close()
    print(per_ray)
    # (the primitive index a pixel sees differs between the two files, what it shades does not - except where a
    # view ray meets an edge of two triangles at exactly the same distance: the smaller index wins)
    differing = int((frames["morton"].view(np.uint32) != frames["random"].view(np.uint32)).any(axis=-1).sum())
    assert differing <= 8, differing
    for name in ("morton", "random"):
        assert abs(per_ray[name, "sah_device"] - per_ray[name, "sah_host"]) <= 0.05 * per_ray[name, "sah_host"], per_ray
    assert abs(per_ray["random", "sah_device"] - per_ray["morton", "sah_device"]) <= 0.05 * per_ray["morton", "sah_device"], per_ray


def test_full_size_config_3_is_bit_exact_in_the_polynomial_mode_too(big_dataset):
    """BASELINE config 3 at its full size (1920x1080, 4 lights, 4 spp per technique, shadow rays);
    tests/test_gpu_full_size.py has the libm mode"""
    r, image = render_config(big_dataset, 3, 1920, 1080, frames_in_flight=2, arithmetic="exact")
    cpu, _, _ = oracle_render(r, visibility=r.read_visibility(), math_mode=renderer.ORACLE_MATH_MODE[r.arithmetic])
    r.close()
    stats = compare(image, cpu)
    assert stats["bit_exact"] and stats["nan"] == 0, stats


def test_the_whole_full_size_config_4_frame_is_bit_exact(big_dataset):
    """BASELINE config 4 at its full size (3840x2160, 8 lights of 3 ... 6 vertices, 8 spp per technique,
    128 shadow rays per pixel at most): the V = 7 kernel.  EVERY pixel of the frame against the oracle (until round 5
    three bands of 16 rows; the whole frame takes the oracle about a minute on the GPU box's host cores), shaded in
    slices of 120 rows so that a failure names where it is."""
    import oracle
    r, image = render_config(big_dataset, 4, 3840, 2160, frames_in_flight=3)
    assert r.app.shading_pass.max_polygon_vertex_count == 7
    inputs = r.host_inputs(r.read_visibility())
    bvh = oracle.Bvh(inputs["quantized_positions"], inputs["dequantization_factor"], inputs["dequantization_summand"])
    frame = oracle.make_frame(inputs, r.oracle_settings(), bvh)
    oracle.set_math_mode(renderer.ORACLE_MATH_MODE[r.arithmetic])
    differing = {}
    try:
        for y0 in range(0, 2160, 120):
            cpu = oracle.shade(frame, y0, y0 + 120)
            count = int((image[y0:y0 + 120].view(np.uint32) != cpu[y0:y0 + 120].view(np.uint32)).any(axis=-1).sum())
            if count:
                differing[y0] = count
    finally:
        oracle.set_math_mode(0)
        r.close()
    assert not differing, "pixels that differ from the oracle, by first row of the slice: %r" % differing
    assert not np.isnan(image).any()


@pytest.mark.parametrize("frames_in_flight", [1, 2, 3])
@pytest.mark.parametrize("band_count", [2, 3, 7])
def test_a_frame_rendered_in_bands_equals_the_frame_rendered_at_once(dataset, band_count, frames_in_flight, monkeypatch):
    """Bands: the frame as several launches over consecutive blocks, each with wavefront buffers sized
    for the band, overlapping on the frame streams (how BASELINE config 4 keeps its buffers at a few
    GB).  Same frame, same ray count, also when frames follow each other without synchronisation.
    (Every shaft pair walked in every frame, VKR_SHAFT_REST=0: which pairs rest in a later frame depends on what the
    frames before found - tests/test_gpu_light_shafts.py - and with them how many rays are traced; here the count is the
    check that bands queue what a whole frame queues.)"""
    monkeypatch.setenv("VKR_SHAFT_REST", "0")
    whole, expected = render_config(dataset, 3, 512, 288)
    rays = whole.last_ray_count()
    whole.close()
    r = renderer.Renderer(frames_in_flight=frames_in_flight, band_count=band_count)
    renderer.setup_config(r, 3, dataset, width=512, height=288, acceleration_structure="sah_device")
    r.create_targets()
    r.create_pass()
    r.render_visibility()
    for _ in range(4):
        r.render()
    image = r.read_radiance()
    assert r.app.shading_pass.last_band_count == band_count
    assert np.array_equal(image.view(np.uint32), expected.view(np.uint32))
    assert r.last_ray_count() == rays
    # the light display's colour goes through a stream that is only allocated when it is needed
    r.app.render_settings.show_polygonal_lights = 1
    r.render()
    with_lights = r.read_radiance()
    r.close()
    whole, _ = render_config(dataset, 3, 512, 288, show_polygonal_lights=True)
    assert np.array_equal(with_lights.view(np.uint32), whole.read_radiance().view(np.uint32))
    whole.close()


def test_error_display_frame_reports_no_rays_and_bad_settings_are_caught_at_render_time(dataset):
    r = renderer.Renderer()
    renderer.setup_config(r, 3, dataset, width=128, height=72, acceleration_structure=True)
    r.create_targets()
    r.create_pass()
    r.render_visibility()
    r.render()
    assert r.last_ray_count() > 0
    r.app.render_settings.error_display = 1  # diffuse backward error: the program returns before it samples
    r.render()
    assert r.last_ray_count() == 0
    # an out-of-range strategy is legal only while an error display is really shown (experiment_list.c:107)
    r.app.render_settings.error_display = 0
    r.render()
    assert r.last_ray_count() > 0
    r.create_pass()
    r.app.render_settings.sampling_strategies = 5
    r.app.render_settings.error_display = 1
    r.create_pass()
    r.render()
    r.app.render_settings.error_display = 0
    assert r.lib.render_shading_pass(C.byref(r.app), None) == 1
    r.app.render_settings.error_display = 1
    r.app.render_settings.sample_count = 0
    assert r.lib.render_shading_pass(C.byref(r.app), None) == 1
    r.close()


@pytest.mark.parametrize("frames_in_flight", [2, 3])
def test_a_reader_of_the_target_is_ordered_between_two_frames_in_flight(dataset, frames_in_flight):
    """render A, encode A on device->stream, render B into the same target: B's resolve must wait
    for the encoding of A (and for A), although finish_frames() has cleared A's pending flag."""
    r, _ = render_config(dataset, 3, 512, 288, frames_in_flight=frames_in_flight)
    expected = {}
    for exposure in (1.0, 7.0):
        r.app.render_settings.exposure_factor = exposure
        r.render()
        expected[exposure] = r.read_encoded(False, 0)
    assert not np.array_equal(expected[1.0], expected[7.0])
    out = np.zeros((288, 512, 4), np.uint8)
    for _ in range(20):
        r.app.render_settings.exposure_factor = 1.0
        r.render()
        assert r.lib.encode_output(C.byref(r.app), 0) == 0  # asynchronous, reads the radiance target
        r.app.render_settings.exposure_factor = 7.0
        r.render()  # another frame stream, same target
        assert r.lib.read_back_encoded(C.byref(r.app), out.ctypes.data) == 0
        assert np.array_equal(out, expected[1.0])
        assert np.array_equal(r.read_encoded(False, 0), expected[7.0])
    r.close()


@pytest.mark.parametrize("frames_in_flight", [2, 3])
def test_tracing_and_resolve_on_a_high_priority_stream_give_the_same_frames(dataset, frames_in_flight, monkeypatch):
    """VKR_TRACE_STREAM_PRIORITY=high moves the tracing and the resolve kernel of every launch to a
    stream of the highest priority behind an event of the shading kernel (an experiment of round 3 that
    did not pay, profiles/r03_trace.md: the knob stays).  The event hand-overs must order the three
    kernels of a frame and the frames among each other exactly as the single stream does: the same
    frames, bit for bit, with frames following each other without synchronisation, also in bands and
    with a reader of the target in between; and the frame period must not fall apart."""
    def frames(r):
        out = []
        for exposure in (1.0, 3.0, 1.0, 5.0, 3.0):
            r.app.render_settings.exposure_factor = exposure
            r.render()
            out.append(r.read_radiance())
        for _ in range(40):
            r.render()
        r.finish_frames()
        periods = r.frame_period_ms(16)
        return out, r.last_ray_count(), float(np.median(periods)) if periods else 0.0

    # (ray counts are compared between runs with different histories: every shaft pair walked in every frame)
    monkeypatch.setenv("VKR_SHAFT_REST", "0")
    r, _ = render_config(dataset, 3, 512, 288, frames_in_flight=frames_in_flight)
    expected, rays, period = frames(r)
    r.close()
    assert not np.array_equal(expected[0], expected[1])
    monkeypatch.setenv("VKR_TRACE_STREAM_PRIORITY", "high")
    for band_count in (0, 3):
        r = renderer.Renderer(frames_in_flight=frames_in_flight, band_count=band_count)
        renderer.setup_config(r, 3, dataset, width=512, height=288, acceleration_structure="sah_device")
        r.create_targets()
        r.create_pass()
        r.render_visibility()
        got, got_rays, got_period = frames(r)
        r.close()
        assert got_rays == rays
        for a, b in zip(got, expected):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), band_count
        # (two more event hand-overs per launch: measured +50 % on frames of 0.13 ms, nothing on long ones)
        if band_count == 0 and period > 0.0:
            assert got_period <= 3.0 * period + 0.2, (period, got_period)


@pytest.mark.parametrize("slab_format", ["rgba32f", "rgb8"])
def test_slab_exchange_with_one_rank_reproduces_the_frame(dataset, slab_format):
    """The whole multi-GPU chain on one GPU: slab layout, ncclAllGather through the C-ABI (a
    communicator of one rank), scatter on the exchange stream, frames overlapping."""
    r, _ = render_config(dataset, 3, 200, 120, frames_in_flight=2)
    expected = {}
    for exposure in (1.0, 2.0, 3.0):
        r.app.render_settings.exposure_factor = exposure
        r.render()
        expected[exposure] = r.read_radiance() if slab_format == "rgba32f" else r.read_encoded(False, 0)
    r.set_tiles(32, 0, 1, slab_layout=True)
    r.create_exchange(r.exchange_id(), slab_format)
    assert r.exchange.rank_count == 1 and r.exchange.slab_pixel_count == r.slab_pixel_count(0)
    e = r.app.swapchain.extent
    frames = [DeviceBuffer(e.width * e.height * (16 if slab_format == "rgba32f" else 4)) for _ in range(5)]
    exposures = [1.0, 2.0, 3.0, 2.0, 1.0]
    for exposure, frame in zip(exposures, frames):  # five frames, no synchronisation in between
        r.app.render_settings.exposure_factor = exposure
        r.render_and_exchange(frame.ptr.value)
    r.finish_exchange()
    r.sync()
    for exposure, frame in zip(exposures, frames):
        if slab_format == "rgba32f":
            got = frame.download((e.height, e.width, 4), np.float32)
            assert np.array_equal(got.view(np.uint32), expected[exposure].view(np.uint32)), exposure
        else:
            assert np.array_equal(frame.download((e.height, e.width, 4), np.uint8), expected[exposure]), exposure
        frame.free()
    # into the render targets, which read_back_* see after finish_slab_exchange()
    r.app.render_settings.exposure_factor = 3.0
    r.render_and_exchange(None)
    r.finish_exchange()
    if slab_format == "rgba32f":
        assert np.array_equal(r.read_radiance().view(np.uint32), expected[3.0].view(np.uint32))
    else:
        out = np.zeros((e.height, e.width, 4), np.uint8)
        assert r.lib.read_back_encoded(C.byref(r.app), out.ctypes.data) == 0
        assert np.array_equal(out, expected[3.0])
    stages = r.exchange_ms()
    assert stages is not None and all(ms >= 0.0 for ms in stages)
    # on demand (round 6): frames stay the gathered slabs, tile-major; the reader - finish_slab_exchange() for the render
    # target, assemble_exchanged_frame() for a buffer of the caller - un-tiles the most recent one
    r.assemble_on_demand(True)
    for exposure in (1.0, 3.0, 2.0):
        r.app.render_settings.exposure_factor = exposure
        r.render_and_exchange(None)
    assert r.exchange.last_frame_assembled == 0
    mine = DeviceBuffer(e.width * e.height * (16 if slab_format == "rgba32f" else 4))
    r.assemble_exchanged(mine.ptr.value)
    r.finish_exchange()
    assert r.exchange.last_frame_assembled == 1
    r.sync()
    if slab_format == "rgba32f":
        assert np.array_equal(r.read_radiance().view(np.uint32), expected[2.0].view(np.uint32))
        assert np.array_equal(mine.download((e.height, e.width, 4), np.float32).view(np.uint32), expected[2.0].view(np.uint32))
    else:
        assert r.lib.read_back_encoded(C.byref(r.app), out.ctypes.data) == 0
        assert np.array_equal(out, expected[2.0])
        assert np.array_equal(mine.download((e.height, e.width, 4), np.uint8), expected[2.0])
    mine.free()
    r.destroy_exchange()
    r.close()


@pytest.mark.parametrize("slab_format", ["rgba32f", "rgb8"])
@pytest.mark.parametrize("rank_count, tile_size, frames_in_flight, band_count", [(2, 32, 3, 0), (2, 16, 2, 2), (4, 64, 3, 0), (3, 32, 1, 0), (2, 32, 4, 0), (2, 16, 7, 0)])
def test_ranks_of_one_process_exchange_their_slabs_end_to_end(dataset, rank_count, tile_size, frames_in_flight, band_count, slab_format):
    """render_and_exchange_frame() with rank_count > 1 on ONE device: every rank is an application_t of its
    own, driven by its own thread; the collective is the local one (device-to-device copies with a
    host-side rendezvous, create_local_slab_exchange) where RCCL would refuse two ranks on one
    device.  Tile schedule, slabs, buffer sets, the overlap with the next frames, event order and
    scatter are the code that runs on N GPUs.  The VERY FIRST frame of every fresh pass goes into a
    buffer of its own and must be complete (the pass picks the stream of a frame; the exchange used to
    guess it and guessed wrong for the first frame), and so must every later one, on every rank."""
    import threading
    width, height = 200, 120
    single, _ = render_config(dataset, 3, width, height)
    exposures = [1.0, 2.0, 3.0, 2.0, 1.0, 3.0, 2.0]
    expected = {}
    for exposure in sorted(set(exposures)):
        single.app.render_settings.exposure_factor = exposure
        single.render()
        expected[exposure] = single.read_radiance() if slab_format == "rgba32f" else single.read_encoded(False, 0)
    single.close()
    lib = renderer.capi.load()
    group = lib.create_local_slab_group(rank_count)
    assert group
    bytes_per_pixel = 16 if slab_format == "rgba32f" else 4
    results, errors = {}, []

    def run_rank(rank):
        try:
            # (band_count 2: every rank renders its slab as two launches on two streams; the exchange waits for the last)
            r = renderer.Renderer(frames_in_flight=frames_in_flight, band_count=band_count)
            renderer.setup_config(r, 3, dataset, width=width, height=height, acceleration_structure="sah_device")
            r.set_tiles(tile_size, rank, rank_count, slab_layout=True)
            r.create_targets()
            r.create_pass()
            r.render_visibility()
            r.create_local_exchange(group, slab_format)
            assert r.exchange.rank == rank and r.exchange.rank_count == rank_count
            frames = [DeviceBuffer(width * height * bytes_per_pixel) for _ in exposures]
            for exposure, frame in zip(exposures, frames):  # no synchronisation in between; the first call is the first frame of the pass
                r.app.render_settings.exposure_factor = exposure
                r.render_and_exchange(frame.ptr.value)
            r.finish_exchange()
            r.sync()
            results[rank] = [frame.download((height, width, 4), np.float32 if slab_format == "rgba32f" else np.uint8) for frame in frames]
            for frame in frames:
                frame.free()
            # the same ranks with frames that stay tile-major until they are read (assemble_on_demand): the last frame of
            # a run of frames, un-tiled by finish_slab_exchange() into the render target
            r.assemble_on_demand(True)
            for exposure in exposures[:frames_in_flight + 2]:
                r.app.render_settings.exposure_factor = exposure
                r.render_and_exchange(None)
            r.finish_exchange()
            if slab_format == "rgba32f":
                results[rank].append(r.read_radiance())
            else:
                encoded = np.zeros((height, width, 4), np.uint8)
                assert r.lib.read_back_encoded(C.byref(r.app), encoded.ctypes.data) == 0
                results[rank].append(encoded)
            r.destroy_exchange()
            r.close()
        except Exception as error:  # (a rank that dies would leave its peers in the rendezvous)
            errors.append((rank, repr(error)))
            raise

    threads = [threading.Thread(target=run_rank, args=(rank,)) for rank in range(rank_count)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors
    assert all(not t.is_alive() for t in threads), "a rank is stuck in the exchange"
    lib.destroy_local_slab_group(group)
    for rank in range(rank_count):
        for index, exposure in enumerate(exposures + [exposures[min(len(exposures), frames_in_flight + 2) - 1]]):
            got, want = results[rank][index], expected[exposure]
            same = np.array_equal(got.view(np.uint32), want.view(np.uint32)) if slab_format == "rgba32f" else np.array_equal(got, want)
            assert same, (rank, index, exposure, int((got != want).any(axis=-1).sum()))


def _local_ranks(rank_count, body):
    """runs body(rank, group) on one thread per rank; -> (results, errors, stuck)"""
    import threading
    lib = renderer.capi.load()
    group = lib.create_local_slab_group(rank_count)
    assert group
    results, errors = {}, {}

    def run(rank):
        try:
            results[rank] = body(rank, group)
        except Exception as error:
            errors[rank] = repr(error)

    threads = [threading.Thread(target=run, args=(rank,)) for rank in range(rank_count)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    stuck = [t for t in threads if t.is_alive()]
    if not stuck:
        lib.destroy_local_slab_group(group)
    return results, errors, stuck


def test_a_rank_that_fails_before_the_gather_releases_its_peers(dataset):
    """the rendezvous of the local gather had no way out - a rank that returned from
    render_and_exchange_frame() with an error before it reached the gather left its peers waiting forever.
    Rank 1 changes its tile schedule after two frames (the exchange then refuses the frame); rank 0 must get an
    error from its third frame instead of hanging."""
    width, height = 200, 120

    def body(rank, group):
        r = renderer.Renderer(frames_in_flight=2)
        renderer.setup_config(r, 3, dataset, width=width, height=height, acceleration_structure="sah_device")
        r.set_tiles(32, rank, 2, slab_layout=True)
        r.create_targets()
        r.create_pass()
        r.render_visibility()
        r.create_local_exchange(group)
        submitted, failed_at = 0, None
        for frame in range(6):
            if rank == 1 and frame == 2:
                r.set_tiles(16, rank, 2, slab_layout=True)
            try:
                r.render_and_exchange(None)
                submitted += 1
            except RuntimeError:
                failed_at = frame
                break
        r.sync()
        r.destroy_exchange()
        r.close()
        return submitted, failed_at

    results, errors, stuck = _local_ranks(2, body)
    assert not stuck, "a rank is stuck in the rendezvous of the local gather"
    assert not errors, errors
    assert results[1] == (2, 2), results
    # (rank 0 is told at the frame whose rendezvous rank 1 never reaches - the third - or, if its thread was ahead, the next)
    assert results[0][1] is not None and 2 <= results[0][1] <= 3, results


def test_ranks_of_a_local_group_need_the_same_number_of_buffer_sets(dataset):
    """sets are addressed by index across the ranks of a local group: a rank whose frames_in_flight gives it another
    number of sets is refused when it joins"""
    def body(rank, group):
        r = renderer.Renderer(frames_in_flight=2 + rank)
        renderer.setup_config(r, 2, dataset, width=128, height=72, acceleration_structure="sah_device")
        r.set_tiles(32, rank, 2, slab_layout=True)
        r.create_targets()
        r.create_pass()
        try:
            r.create_local_exchange(group)
            joined = True
        except RuntimeError:
            joined = False
        r.sync()
        if joined:
            r.destroy_exchange()
        r.close()
        return joined

    results, errors, stuck = _local_ranks(2, body)
    assert not stuck and not errors, (errors, stuck)
    assert sorted(results.values()) == [False, True], results


def test_all_gather_slabs_refuses_a_foreign_buffer_with_a_set_addressed_transport(dataset):
    """all_gather_slabs() fell back to set 0 when `gathered` was not one of the exchange's buffers;
    the local copies then wrote into the peers' set 0 and the caller's buffer stayed empty"""
    r, _ = render_config(dataset, 2, 128, 72, frames_in_flight=2)
    r.set_tiles(32, 0, 1, slab_layout=True)
    lib = renderer.capi.load()
    group = lib.create_local_slab_group(1)
    r.create_local_exchange(group)
    foreign = DeviceBuffer(int(r.exchange.send_bytes))
    assert r.lib.all_gather_slabs(C.byref(r.exchange), r.exchange.send[0], foreign.ptr, r.exchange.stream) == 1
    assert r.lib.all_gather_slabs(C.byref(r.exchange), r.exchange.send[0], r.exchange.gathered[1], r.exchange.stream) == 0
    r.sync()
    foreign.free()
    r.destroy_exchange()
    r.close()
    lib.destroy_local_slab_group(group)


def test_exchange_through_a_gather_function_of_the_caller(dataset):
    """create_slab_exchange_with_gather(): the collective is a function pointer; here a Python callback that
    copies the one rank's slab (hipMemcpyAsync on the stream it is given)"""
    r, _ = render_config(dataset, 3, 200, 120, frames_in_flight=2)
    r.render()
    expected = r.read_radiance()
    r.set_tiles(32, 0, 1, slab_layout=True)
    hip = C.CDLL("libamdhip64.so")
    calls = []
    GATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)

    @GATHER
    def gather(context, rank, buffer_set, send, gathered, send_bytes, stream):
        calls.append((rank, buffer_set, send_bytes))
        return hip.hipMemcpyAsync(C.c_void_p(gathered), C.c_void_p(send), C.c_size_t(send_bytes), 3, C.c_void_p(stream))

    exchange = renderer.capi.SlabExchange()
    assert r.lib.create_slab_exchange_with_gather(C.byref(exchange), C.byref(r.app), C.cast(gather, C.c_void_p), None, 0) == 0
    r.exchange = exchange
    for _ in range(4):
        r.render_and_exchange(None)
    r.finish_exchange()
    assert np.array_equal(r.read_radiance().view(np.uint32), expected.view(np.uint32))
    assert len(calls) == 4 and [c[1] for c in calls] == [0, 1, 0, 1] and calls[0][2] == r.slab_pixel_count(0) * 16
    assert r.lib.create_slab_exchange_with_gather(C.byref(renderer.capi.SlabExchange()), C.byref(r.app), None, None, 0) == 1
    r.destroy_exchange()
    r.close()


def test_exchange_refuses_a_frame_layout_and_a_changed_schedule(dataset):
    r, _ = render_config(dataset, 2, 128, 72)
    token = r.exchange_id()
    with pytest.raises(RuntimeError):
        r.create_exchange(token)  # one rank without slab_layout
    r.set_tiles(32, 0, 1, slab_layout=True)
    r.create_exchange(token)
    r.set_tiles(16, 0, 1, slab_layout=True)
    with pytest.raises(RuntimeError):
        r.render_and_exchange(None)
    r.destroy_exchange()
    r.close()


def test_c_program_tiles_an_experiment_over_the_gpus_of_the_node(tmp_path):
    """vkr_multi_gpu (csrc/examples/vkr_multi_gpu.c): threads + RCCL behind the C-ABI; with one
    GPU the communicator has one rank, the chain is the same"""
    binary = os.path.join(os.path.dirname(renderer.__file__), "vkr_multi_gpu")
    assert os.path.exists(binary), "run make -C vulkan_renderer_amd/csrc (build() does)"
    root = str(tmp_path / "root")
    made = experiments.write_synthetic_data_root(root, grid=64, box_count=16)
    table = experiments.experiment_table()
    index = next(i for i in range(table.count) if table.experiments[i].screenshot_path == b"data/experiments/mis_plane_clamped_optimal_ours_2spp_%.3f.png")
    for slab_format in ("rgba32f", "rgb8"):
        done = subprocess.run([binary, "--gpus", "1", "-e%d" % index, "--frames", "12", "--format", slab_format, "--white-noise",
                               "--fresnel", str(made["fresnel_count"]), root], capture_output=True, text=True, timeout=300)
        assert done.returncode == 0, done.stdout + done.stderr
        assert "0 values differ from the single-GPU frame" in done.stdout, done.stdout
    assert os.path.exists(os.path.join(root, "data", "multi_gpu.png"))


@pytest.mark.parametrize("frames_in_flight", [1, 3])
def test_asynchronous_read_back_delivers_every_frame_while_the_next_ones_render(dataset, frames_in_flight):
    """begin_read_back() / end_read_back() (round 6): copies into pinned staging on a stream of their own, ordered behind the
    frame they read and in front of the next frame that writes the same buffer.  (a) one target, a copy queued behind every
    frame and collected a frame later: each staging buffer holds ITS frame although the next one was already queued;
    (b) a ring of targets, all copies in flight at once."""
    r, _ = render_config(dataset, 3, 200, 120, frames_in_flight=frames_in_flight)
    e = r.app.swapchain.extent
    expected = {}
    for exposure in (1.0, 2.0, 3.0):
        r.app.render_settings.exposure_factor = exposure
        r.render()
        expected[exposure] = r.read_radiance()
    exposures = [1.0, 2.0, 3.0, 1.0, 3.0, 2.0, 1.0]
    # (a) the render target itself, two slots in turn
    got = []
    for index, exposure in enumerate(exposures):
        r.app.render_settings.exposure_factor = exposure
        r.render()
        r.begin_read_back(index % 2)
        if index:
            got.append(r.end_read_back((index - 1) % 2).copy())
    got.append(r.end_read_back((len(exposures) - 1) % 2).copy())
    for exposure, image in zip(exposures, got):
        assert np.array_equal(image.view(np.uint32), expected[exposure].view(np.uint32)), exposure
    # (b) a ring of targets of the caller
    targets = [DeviceBuffer(e.width * e.height * 16) for _ in range(4)]
    for index, exposure in enumerate(exposures[:4]):
        r.app.render_settings.exposure_factor = exposure
        r.render(targets[index].ptr.value)
        r.begin_read_back(index, targets[index].ptr.value, e.width * e.height * 16)
    for index, exposure in enumerate(exposures[:4]):
        assert np.array_equal(r.end_read_back(index).view(np.uint32), expected[exposure].view(np.uint32)), (index, exposure)
    # an encoded frame goes the same way (the source is produced on device->stream)
    r.app.render_settings.exposure_factor = 2.0
    r.render()
    encoded = r.read_encoded(False, 0)
    r.begin_read_back(0, r.app.render_targets.encoded, e.width * e.height * 4)
    assert np.array_equal(r.end_read_back(0, (e.height, e.width, 4), np.uint8), encoded)
    assert r.lib.begin_read_back(C.byref(r.app), 99, None, 0) == 1
    for t in targets:
        t.free()
    r.close()
