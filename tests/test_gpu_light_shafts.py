"""Light shafts (csrc/light_shafts.h): (8x8 pixel patch, light) pairs whose shadow rays cannot be blocked by anything are
found by one conservative walk of the BVH per pair, and their rays are never queued; pairs whose rays can only meet a
handful of triangles get those triangles as an occluder list, and the shading kernel decides their rays itself.  Every
ray query keeps its result, so every frame keeps every bit: rendered with the test (the default) and without it
(VKR_LIGHT_SHAFTS=0), with the lists and without them (VKR_SHAFT_LISTS=0), on both scenes, from random cameras under
random lights, on a slice of the random sweep and on lights that graze, touch or surround the geometry.
(The whole GPU suite runs with shafts and lists on: every bit-parity test against the oracle checks them as well.)"""
import math

import numpy as np
import pytest

import golden_cases
import test_gpu_sweep
from vulkan_renderer_amd import renderer, synthetic

pytestmark = pytest.mark.gpu


def frames_with_and_without(monkeypatch, setup, frames_in_flight=1, arithmetic="libm"):
    """-> (frame with shafts, frame without, rays with, rays without, shaft statistics)"""
    out = {}
    for shafts in (1, 0):
        monkeypatch.setenv("VKR_LIGHT_SHAFTS", str(shafts))
        r = renderer.Renderer(frames_in_flight=frames_in_flight, arithmetic=arithmetic)
        setup(r)
        r.create_targets()
        r.create_pass()
        r.render_visibility()
        for _ in range(frames_in_flight):
            r.render()
        out[shafts] = (r.read_radiance(), r.last_ray_count(), r.light_shaft_statistics())
        r.close()
    monkeypatch.delenv("VKR_LIGHT_SHAFTS")
    return out[1][0], out[0][0], out[1][1], out[0][1], out[1][2], out[0][2]


@pytest.mark.parametrize("config, width, height", [(2, 960, 540), (3, 1920, 1080), ("target", 1280, 720), (4, 1920, 1080)])
def test_benchmark_scene_same_frame_fewer_rays(big_dataset, monkeypatch, config, width, height):
    on, off, rays_on, rays_off, stats, stats_off = frames_with_and_without(
        monkeypatch, lambda r: renderer.setup_config(r, config, big_dataset, width=width, height=height, acceleration_structure="sah_device"), frames_in_flight=2)
    print(config, stats, rays_on, rays_off)
    assert np.array_equal(on.view(np.uint32), off.view(np.uint32)), int((on != off).any(axis=-1).sum())
    assert stats_off["pairs"] == 0 and stats["pairs"] > 0
    # an open scene: a good part of the patches see a good part of the lights freely (40 % of the patches are sky,
    # a quarter of the rest faces away from a given light)
    assert stats["clear_pairs"] > 0.15 * stats["pairs"], stats
    assert stats["clear_pairs"] + stats["list_pairs"] + sum(stats["not_clear"].values()) == stats["pairs"]
    assert 0 < rays_on < 0.8 * rays_off, (rays_on, rays_off)


@pytest.mark.parametrize("config, width, height", [(3, 1280, 720), (4, 960, 540)])
def test_the_fast_mode_keeps_its_frame_too(big_dataset, monkeypatch, config, width, height):
    """Round 6: the fast arithmetic mode runs with shafts, occluder lists and pre-summed final terms like the other two.  Its
    translation units contract a b + c into fused operations, so the two places where the shading kernel repeats what another
    kernel does - the triangle test of an occluder list (the tracing kernel's), the sum of a clear light's terms (the resolve
    kernel's) - are kept free of contraction; the frame is then the same with the test and without it, bit for bit."""
    on, off, rays_on, rays_off, stats, stats_off = frames_with_and_without(
        monkeypatch, lambda r: renderer.setup_config(r, config, big_dataset, width=width, height=height, acceleration_structure="sah_device"), frames_in_flight=2, arithmetic="fast")
    assert np.array_equal(on.view(np.uint32), off.view(np.uint32)), int((on != off).any(axis=-1).sum())
    assert stats["list_pairs"] > 0 and stats["clear_pairs"] > 0 and 0 < rays_on < 0.8 * rays_off, (stats, rays_on, rays_off)


def test_large_scene_same_frame(monkeypatch, tmp_path):
    dataset = synthetic.write_dataset(str(tmp_path / "large"), seed=4321, ltc_resolution=32, fresnel_count=16, large={})
    on, off, rays_on, rays_off, stats, _ = frames_with_and_without(
        monkeypatch, lambda r: renderer.setup_config(r, 3, dataset, width=1920, height=1080, acceleration_structure="sah_device"))
    print(stats, rays_on, rays_off)
    assert np.array_equal(on.view(np.uint32), off.view(np.uint32)), int((on != off).any(axis=-1).sum())
    assert stats["pairs"] > 0 and rays_on <= rays_off


@pytest.fixture(scope="module")
def large_dataset(tmp_path_factory):
    return synthetic.write_dataset(str(tmp_path_factory.mktemp("shafts_large")), seed=4321, ltc_resolution=32, fresnel_count=16, large={})


@pytest.mark.parametrize("frames_in_flight", [1, 3])
def test_pairs_whose_walk_met_too_many_triangles_rest_for_a_few_frames(large_dataset, monkeypatch, frames_in_flight):
    """Round 5 (light_shafts.h, kShaftResting): behind fences and louvres most walks end at "more triangles in the way
    than a list holds".  A verdict is a hint - without one the rays are traced - so the table of a frame context keeps the
    verdicts of its previous frame, and such a pair is not walked again for seven of that context's frames.  Every frame
    must be the same frame, with the same number of traced rays; the failed pairs of the first frame are exactly the
    resting pairs of the following ones, and after seven rests they are walked (and fail) again.  VKR_SHAFT_REST=0
    walks every pair in every frame."""
    def frames(rest, count):
        if rest is None:
            monkeypatch.delenv("VKR_SHAFT_REST", raising=False)
        else:
            monkeypatch.setenv("VKR_SHAFT_REST", str(rest))
        r = renderer.Renderer(frames_in_flight=frames_in_flight)
        renderer.setup_config(r, 3, large_dataset, width=960, height=544, acceleration_structure="sah_device")
        r.create_targets()
        r.create_pass()
        r.render_visibility()
        out = []
        for _ in range(count):
            r.render()
            out.append((r.read_radiance(), r.last_ray_count(), r.light_shaft_statistics()))
        r.close()
        return out
    contexts = frames_in_flight
    resting = frames(None, 10 * contexts)
    walking = frames(0, 2 * contexts)
    first_image, first_rays, first = resting[0]
    failed = first["not_clear"]["triangle_in_the_way"]
    print(first)
    assert failed > 0.2 * first["pairs"] and first["not_clear"]["other"] == 0
    for index, (image, rays, stats) in enumerate(resting):
        assert np.array_equal(image.view(np.uint32), first_image.view(np.uint32)), index
        # The lights of a patch are walked together and share the walk's frontier and step budget: while a hopeless light
        # rests, the walks of the others get further - a few more pairs end clear or with a list, a few more of them find
        # out that they are hopeless too (and rest from the next frame on).  Never fewer, never more rays.
        assert rays <= first_rays and rays > 0.98 * first_rays, (index, rays, first_rays)
        assert stats["clear_pairs"] + stats["list_pairs"] >= first["clear_pairs"] + first["list_pairs"], index
        assert stats["clear_pairs"] + stats["list_pairs"] + sum(stats["not_clear"].values()) == stats["pairs"] == first["pairs"]
        use = index // contexts  # how often this frame's context has rendered before
        if use == 0:
            assert stats["not_clear"] == first["not_clear"], (index, stats)
        elif use % 8 == 0:
            # the pairs that failed in the first frame are walked again, and most of them fail the same way again (not all:
            # the lights that found out later rest now, and walks that share a patch end differently without them)
            assert stats["not_clear"]["triangle_in_the_way"] >= 0.5 * failed, (index, stats)
        else:
            # they rest - all of them; other lights of their patches may find out now that they are hopeless too (a small
            # launch like this one gives a patch twelve steps: without the resting lights the others get to their triangles)
            assert stats["not_clear"]["other"] >= failed, (index, stats)
            if use == 1:
                assert stats["not_clear"]["other"] == failed, (index, stats)
    for image, rays, stats in walking:
        assert np.array_equal(image.view(np.uint32), first_image.view(np.uint32))
        assert rays == first_rays and stats["not_clear"]["triangle_in_the_way"] == failed and stats["not_clear"]["other"] == 0
    monkeypatch.delenv("VKR_SHAFT_REST", raising=False)


def test_a_moved_light_does_not_lean_on_old_verdicts_for_long(large_dataset, monkeypatch):
    """the hint may be stale - lights and camera move - but never wrong: after the lights have changed, frames with resting
    pairs equal the frames of a renderer that walks every pair"""
    rng = np.random.default_rng(5)
    # (four quads like the configuration's own lights: the same kernel variant and constant buffer)
    sets = [[test_gpu_sweep.random_light(rng, 4) for _ in range(4)] for _ in range(3)]

    def frames(rest):
        monkeypatch.setenv("VKR_SHAFT_REST", str(rest))
        r = renderer.Renderer(frames_in_flight=2)
        renderer.setup_config(r, 3, large_dataset, width=640, height=368, acceleration_structure="sah_device")
        r.create_targets()
        r.create_pass()
        r.render_visibility()
        out = []
        for lights in sets:
            r.set_lights(lights)
            for _ in range(3):
                r.render()
                out.append((r.read_radiance(), r.last_ray_count()))
        r.close()
        return out
    with_rests, without = frames(7), frames(0)
    monkeypatch.delenv("VKR_SHAFT_REST", raising=False)
    for index, ((a, rays_a), (b, rays_b)) in enumerate(zip(with_rests, without)):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), index
        # (how many rays are traced differs a little either way: a pair that rests while its new shaft would have been clear
        # has its rays traced, and the lights that share a walk with a resting one get further)
        assert abs(rays_a - rays_b) < 0.05 * rays_b, (index, rays_a, rays_b)


@pytest.mark.parametrize("seed", range(6))
def test_large_scene_from_random_cameras_under_random_lights(large_dataset, monkeypatch, seed):
    """the scene where a wrong "clear" would show (towers, fences, slats between the floor and the lights; 70 % of the
    rays blocked), seen from other places than the benchmark's and lit by other polygons than its four quads"""
    rng = np.random.default_rng(77 + seed)
    lights = [test_gpu_sweep.random_light(rng, int(rng.integers(3, 8))) for _ in range(int(rng.integers(1, 6)))]
    while True:
        position = (float(rng.uniform(-8.0, 8.0)), float(rng.uniform(-8.0, 8.0)), float(rng.uniform(0.3, 4.0)))
        if math.hypot(position[0] - synthetic.DEFAULT_CAMERA["position"][0], position[1] - synthetic.DEFAULT_CAMERA["position"][1]) > 0.5:
            break
    rotation_x, rotation_z, fov = float(rng.uniform(0.25, 0.45) * math.pi), float(rng.uniform(0.0, 2.0) * math.pi), float(rng.uniform(0.2, 0.45) * math.pi)

    def setup(r):
        renderer.setup_config(r, 3, large_dataset, width=960, height=544, acceleration_structure="sah_device")
        r.set_camera(position, rotation_x, rotation_z, fov, synthetic.DEFAULT_CAMERA["near"], synthetic.DEFAULT_CAMERA["far"])
        r.set_lights(lights)
        r.set_settings(sample_count=1 + seed % 3)
    on, off, rays_on, rays_off, stats, _ = frames_with_and_without(monkeypatch, setup, frames_in_flight=1 + seed % 2)
    print(seed, position, stats, rays_on, rays_off)
    assert np.array_equal(on.view(np.uint32), off.view(np.uint32)), (seed, int((on != off).any(axis=-1).sum()), stats)
    # (a camera that looks down always sees the floor: a case without rays would test nothing)
    assert rays_on <= rays_off and rays_off > 0 and stats["pairs"] > stats["not_clear"]["no_shaded_pixel"]


@pytest.mark.parametrize("seed", [s for s in range(60) if test_gpu_sweep.random_case(s)["rays"]][:24])
def test_random_configurations_same_frame(seed, monkeypatch, tmp_path_factory):
    dataset = synthetic.write_dataset(str(tmp_path_factory.mktemp("shafts_sweep")), **golden_cases.DATASET)
    case = test_gpu_sweep.random_case(seed)

    def setup(r):
        golden_cases.apply_case(r, case, dataset, 160, 96)
        cam = dict(synthetic.DEFAULT_CAMERA, **case["camera"])
        r.set_camera(cam["position"], cam["rotation_x"], cam["rotation_z"], cam["vertical_fov"], cam["near"], cam["far"])
        r.set_settings(roughness_factor=case["roughness_factor"], exposure_factor=case["exposure_factor"], mis_visibility_estimate=case["mis_visibility_estimate"])
    on, off, rays_on, rays_off, stats, _ = frames_with_and_without(monkeypatch, setup, frames_in_flight=1 + seed % 3)
    assert np.array_equal(on.view(np.uint32), off.view(np.uint32)), (seed, int((on != off).any(axis=-1).sum()), stats)
    assert rays_on <= rays_off


def test_lights_that_touch_graze_or_surround_the_geometry(monkeypatch, tmp_path):
    """where a shaft has no business being clear: a light lying on the floor, one standing in it, one that the floor
    cuts in two, a ceiling over the whole scene, a light inside a box's footprint, a speck, a light far away"""
    pi = math.pi
    dataset = synthetic.write_dataset(str(tmp_path / "small"), **golden_cases.DATASET)
    quad = synthetic.QUAD
    lights = [
        synthetic.light_spec(quad, (-1.0, 1.0, 2.0e-3), (pi, 0.0, 0.0), (10, 10, 10), (2.0, 2.0)),            # on the floor, facing up
        synthetic.light_spec(quad, (-2.0, 0.0, 0.0), (0.5 * pi, 0.0, 0.7), (8, 8, 8), (1.5, 1.5)),               # upright, its lower edge in the floor
        synthetic.light_spec(quad, (0.5, 2.0, -0.4), (0.5 * pi, 0.0, 2.0), (8, 8, 8), (1.0, 1.2)),               # cut in two by the floor
        synthetic.light_spec(quad, (-12.0, 12.0, 4.0), (pi, 0.0, 0.0), (40, 40, 40), (24.0, 24.0)),             # a ceiling over everything
        synthetic.light_spec(synthetic.regular_polygon(7, 0.5, 0.1), (-1.5, 1.5, 0.3), (0.6 * pi, 0.15, 1.1), (14, 12, 9), (1.8, 1.3)),  # between the boxes
        synthetic.light_spec([(0, 0), (1, 0), (0, 1)], (-2.0, 0.5, 0.8), (0.7 * pi, 0.2, 1.0), (1e3, 1e3, 1e3), (2e-3, 5e-3)),          # a speck
        synthetic.light_spec(synthetic.regular_polygon(5), (300.0, 500.0, 800.0), (pi, 0.0, 0.0), (1e7, 1e7, 1e7), (40.0, 40.0)),      # far away
    ]
    for strategy, heuristic, technique in ((3, 3, "projected_solid_angle"), (3, 4, "projected_solid_angle"), (1, 0, "projected_solid_angle"),
                                           (0, 0, "solid_angle"), (1, 0, "clipped_solid_angle"), (2, 0, "projected_solid_angle_biased")):
        case = dict(lights=lights, strategy=strategy, heuristic=heuristic, samples=2, rays=True, technique=technique)
        on, off, rays_on, rays_off, stats, _ = frames_with_and_without(monkeypatch, lambda r: golden_cases.apply_case(r, case, dataset, 192, 108))
        assert np.array_equal(on.view(np.uint32), off.view(np.uint32)), (strategy, heuristic, technique, int((on != off).any(axis=-1).sum()), stats)
        assert rays_on <= rays_off


def lists_on_and_off(monkeypatch, setup, frames_in_flight=1):
    """-> {lists: (frame, rays, statistics)} with the shafts on and the occluder lists on / off"""
    out = {}
    monkeypatch.setenv("VKR_LIGHT_SHAFTS", "1")
    for lists in (1, 0):
        monkeypatch.setenv("VKR_SHAFT_LISTS", str(lists))
        r = renderer.Renderer(frames_in_flight=frames_in_flight)
        setup(r)
        r.create_targets()
        r.create_pass()
        r.render_visibility()
        for _ in range(frames_in_flight):
            r.render()
        out[lists] = (r.read_radiance(), r.last_ray_count(), r.light_shaft_statistics())
        r.close()
    monkeypatch.delenv("VKR_SHAFT_LISTS")
    return out


@pytest.mark.parametrize("config, width, height", [(3, 1920, 1080), ("target", 1280, 720), (4, 1920, 1080), (2, 960, 540)])
def test_occluder_lists_same_frame_far_fewer_rays(big_dataset, monkeypatch, config, width, height):
    """Where a shaft walk finds a handful of triangles in the way and nothing else, the shading kernel tests the rays of
    the pair against those triangles itself (the tracing kernel's triangle test, on the spot) instead of queueing them:
    most of the rays that were left after the clear pairs, and every bit of the frame as it was."""
    out = lists_on_and_off(monkeypatch, lambda r: renderer.setup_config(r, config, big_dataset, width=width, height=height, acceleration_structure="sah_device"), frames_in_flight=2)
    (on, rays_on, stats), (off, rays_off, stats_off) = out[1], out[0]
    print(config, stats, rays_on, rays_off)
    assert np.array_equal(on.view(np.uint32), off.view(np.uint32)), int((on != off).any(axis=-1).sum())
    assert stats_off["list_pairs"] == 0 and stats["list_pairs"] > 0
    assert stats["list_pairs"] <= stats["listed_triangles"] <= 12 * stats["list_pairs"]
    # (a pair with a list was a pair with "a triangle in the way"; the lights of a patch are walked together, and where
    # one of them now goes on collecting triangles until the walk is called off as too long, a pair that used to come out
    # clear may be traced instead: a tenth of a per cent of them)
    walked = lambda t: t["clear_pairs"] + t["list_pairs"] + t["not_clear"]["triangle_in_the_way"] + t["not_clear"]["walk_too_long"] + t["not_clear"]["queue_full"]
    assert walked(stats) == walked(stats_off)
    assert 0.99 * stats_off["clear_pairs"] <= stats["clear_pairs"] <= stats_off["clear_pairs"]
    assert 0 < rays_on < (0.5 if config != 2 else 1.0) * rays_off, (rays_on, rays_off)


@pytest.mark.parametrize("heuristic", ["optimal", "optimal_clamped", "balance"])
def test_occluder_lists_with_terms_that_survive_a_blocked_ray(big_dataset, monkeypatch, heuristic):
    """the optimal MIS heuristic is the one estimator whose term has a value when its ray is blocked: a ray that the list
    blocks leaves that value (not nothing) in the sum"""
    out = lists_on_and_off(monkeypatch, lambda r: renderer.setup_config(r, 3, big_dataset, width=960, height=540, acceleration_structure="sah_device", mis_heuristic=heuristic))
    assert np.array_equal(out[1][0].view(np.uint32), out[0][0].view(np.uint32)), heuristic
    assert out[1][2]["list_pairs"] > 0 and out[1][1] < out[0][1]


def test_shafts_are_automatic_where_they_pay(big_dataset, monkeypatch):
    """without VKR_LIGHT_SHAFTS in the environment the pass decides: on for config 3 (32 rays per pixel at most) and the
    target shape (8; since the occluder lists 0.44 instead of 0.50 ms per frame), off for config 2 (2 rays per pixel: a
    walk per patch costs more than its rays)"""
    monkeypatch.delenv("VKR_LIGHT_SHAFTS", raising=False)
    for config, expected in ((3, True), (2, False), ("target", True)):
        r = renderer.Renderer()
        renderer.setup_config(r, config, big_dataset, width=640, height=360, acceleration_structure="sah_device")
        r.create_targets()
        r.create_pass()
        r.render_visibility()
        r.render()
        stats = r.light_shaft_statistics()
        r.close()
        assert (stats["pairs"] > 0) == expected, (config, stats)


def test_techniques_whose_samples_may_miss_the_polygon_are_left_alone(big_dataset, monkeypatch):
    """the shaft holds the rays of techniques that aim at the light polygon; the related-work samplers are not on the list"""
    monkeypatch.setenv("VKR_LIGHT_SHAFTS", "1")
    r = renderer.Renderer()
    renderer.setup_config(r, 3, big_dataset, width=640, height=360, acceleration_structure="sah_device", sampling_strategies="diffuse_only", polygon_technique="area_turk")
    r.create_targets()
    r.create_pass()
    r.render_visibility()
    r.render()
    stats = r.light_shaft_statistics()
    r.close()
    assert stats["pairs"] == 0
