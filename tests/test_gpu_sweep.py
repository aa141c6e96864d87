"""Randomised parity sweep: every axis of the kernel variant space (strategy, MIS heuristic,
technique, light count, polygon sizes 3..7, sample count, shadow rays, light display, error
display, roughness factor, camera) is drawn from a seeded generator and the frame must equal
the oracle's bit for bit - in the libm arithmetic mode (the default of the pass) against the oracle's
libm mode, which is the arithmetic pinned against the reference shader, and in the polynomial
("exact") mode against the oracle's polynomial mode.  Plus the edge cases of
the domain: no lights, lights below the horizon of every pixel, grazing and huge lights,
many lights, many samples, a frame smaller than a workgroup."""
import ctypes as C
import math
import os

import numpy as np
import pytest

import golden_cases
from helpers import compare, oracle_render
from vulkan_renderer_amd import renderer, synthetic

pytestmark = pytest.mark.gpu

WIDTH, HEIGHT = 80, 48


@pytest.fixture(scope="module")
def dataset(tmp_path_factory):
    return synthetic.write_dataset(str(tmp_path_factory.mktemp("sweep_dataset")), **golden_cases.DATASET)


def random_convex_polygon(rng, n):
    angles = np.sort(rng.uniform(0.0, 2.0 * math.pi, n))
    # keep the polygon from degenerating: spread the angles a little
    angles = angles + np.linspace(0.0, 0.3, n)
    radii = rng.uniform(0.35, 0.6)
    return [(0.5 + radii * math.cos(a), 0.5 + radii * math.sin(a)) for a in angles]


def random_light(rng, n):
    return synthetic.light_spec(random_convex_polygon(rng, n),
                                (rng.uniform(-4.0, 2.0), rng.uniform(-1.0, 5.0), rng.uniform(0.3, 3.5)),
                                (rng.uniform(0.3, 1.0) * math.pi, rng.uniform(-0.4, 0.4), rng.uniform(0.0, 2.0 * math.pi)),
                                tuple(rng.uniform(2.0, 14.0, 3)), (rng.uniform(0.4, 2.5), rng.uniform(0.4, 2.5)))


def random_case(seed):
    rng = np.random.default_rng(1000 + seed)
    strategy = int(rng.integers(0, 5))
    if strategy == 0 and rng.random() < 0.35:
        technique = ["baseline", "area_turk", "bilinear_cosine_warp_hart", "bilinear_cosine_warp_clipping_hart",
                     "biquadratic_cosine_warp_hart", "biquadratic_cosine_warp_clipping_hart"][int(rng.integers(0, 6))]
    elif strategy <= 1 and rng.random() < 0.25:
        technique = ["rectangle_solid_angle_urena", "solid_angle_arvo", "projected_solid_angle_arvo"][int(rng.integers(0, 3))]
    elif strategy <= 1:
        technique = ["projected_solid_angle", "projected_solid_angle_biased", "solid_angle", "clipped_solid_angle"][int(rng.integers(0, 4))]
    else:
        technique = ["projected_solid_angle", "projected_solid_angle_biased"][int(rng.integers(0, 2))]
    heuristic = int(rng.integers(0, 5)) if strategy == 3 else int(rng.integers(0, 2))
    light_count = int(rng.integers(1, 5))
    lights = [random_light(rng, int(rng.integers(3, 8))) for _ in range(light_count)]
    error_display = 0
    if technique.startswith("projected") and rng.random() < 0.15:
        error_display = int(rng.integers(1, 4)) if strategy <= 1 else int(rng.integers(1, 7))
        if technique.endswith("arvo"):
            error_display = int(rng.integers(1, 3))  # Arvo's error function has no forward error
    return dict(strategy=strategy, technique=technique, heuristic=heuristic, lights=lights, samples=int(rng.integers(1, 4)),
                rays=bool(rng.random() < 0.6), show_lights=bool(rng.random() < 0.3), error_display=error_display,
                roughness_factor=float(rng.uniform(0.3, 1.5)), exposure_factor=float(rng.uniform(1.0, 10.0)),
                mis_visibility_estimate=float(rng.uniform(0.1, 0.9)),
                camera=dict(position=(rng.uniform(-4.0, 0.0), rng.uniform(-3.0, 0.0), rng.uniform(0.8, 2.5)),
                            rotation_x=rng.uniform(0.35, 0.5) * math.pi, rotation_z=rng.uniform(1.1, 1.5) * math.pi,
                            vertical_fov=rng.uniform(0.25, 0.4) * math.pi))


def render_and_compare(case, dataset, width=WIDTH, height=HEIGHT, inline_rays=False, frames_in_flight=1, arithmetic="libm"):
    r = renderer.Renderer(inline_rays=inline_rays, frames_in_flight=frames_in_flight, arithmetic=arithmetic)
    r.load_scene(dataset["scene"], dataset["textures"], acceleration_structure=True)
    r.load_ltc_table(dataset["ltc"], dataset["fresnel_count"])
    r.load_noise_table("white")
    cam = dict(synthetic.DEFAULT_CAMERA)
    cam.update(case.get("camera", {}))
    r.set_camera(cam["position"], cam["rotation_x"], cam["rotation_z"], cam["vertical_fov"], cam["near"], cam["far"])
    r.set_lights(case["lights"])
    r.set_settings(width=width, height=height, sample_count=case.get("samples", 1), sampling_strategies=case.get("strategy", 0),
                   mis_heuristic=case.get("heuristic", 0), polygon_technique=case.get("technique", "projected_solid_angle"),
                   trace_shadow_rays=case.get("rays", False), show_polygonal_lights=case.get("show_lights", False),
                   error_display=case.get("error_display", 0), error_min_exponent=-7.0,
                   roughness_factor=case.get("roughness_factor", 1.0), exposure_factor=case.get("exposure_factor", 8.0),
                   mis_visibility_estimate=case.get("mis_visibility_estimate", 0.5))
    r.create_targets()
    r.create_pass()
    r.render_visibility()
    for _ in range(frames_in_flight + 1 if frames_in_flight > 1 else 1):  # several frames so that all contexts are used
        r.render()
    image = r.read_radiance()
    cpu, _, _ = oracle_render(r, visibility=r.read_visibility(), math_mode=renderer.ORACLE_MATH_MODE[arithmetic])
    rays = r.last_ray_count()
    r.close()
    return compare(image, cpu), image, rays


# VKR_SWEEP_SEEDS=n widens the sweeps (round 1 ran 600 seeds here and 200 with textures once, round 2 - LDS polygon
# tables, the short exact division, the four-wide tree - 1200 and 400 with the final kernels: all 1608 bit-exact)
@pytest.mark.parametrize("arithmetic", ["libm", "exact"])
@pytest.mark.parametrize("seed", range(int(os.environ.get("VKR_SWEEP_SEEDS", "48"))))
def test_random_configuration_is_bit_exact(seed, arithmetic, dataset):
    case = random_case(seed)
    stats, image, _ = render_and_compare(case, dataset, inline_rays=(seed % 5 == 4), frames_in_flight=1 + seed % 4, arithmetic=arithmetic)
    summary = {k: case[k] for k in ("strategy", "technique", "heuristic", "samples", "rays", "show_lights", "error_display")}
    summary["vertex_counts"] = [len(l["vertices_plane_space"]) for l in case["lights"]]
    assert stats["nan"] == 0 and stats["bit_exact"], (summary, stats)
    assert np.isfinite(image).all()


def test_no_lights_renders_black(dataset):
    stats, image, rays = render_and_compare(dict(lights=[], strategy=3, heuristic=3, rays=True), dataset)
    assert stats["bit_exact"] and rays == 0
    assert not image[..., :3].any() and (image[..., 3] == 1.0).all()


def test_light_below_every_horizon_contributes_nothing(dataset):
    # a light under the floor, facing down: clipped away for every shading point above it
    light = synthetic.light_spec(synthetic.QUAD, (-1.0, 1.0, -3.0), (0.0, 0.0, 0.0), (10, 10, 10), (2.0, 2.0))
    for strategy, technique in ((0, "projected_solid_angle"), (3, "projected_solid_angle"), (1, "clipped_solid_angle")):
        stats, image, rays = render_and_compare(dict(lights=[light], strategy=strategy, technique=technique, heuristic=0, rays=True), dataset)
        assert stats["bit_exact"], (strategy, technique, stats)


def test_grazing_and_huge_lights(dataset):
    lights = [
        # almost in the floor plane: every pixel clips it
        synthetic.light_spec(synthetic.regular_polygon(7, 0.5, 0.2), (-1.0, 1.0, 0.02), (0.5 * math.pi, 0.0, 0.3), (8, 8, 8), (3.0, 0.05)),
        # a ceiling that covers the scene: central case almost everywhere
        synthetic.light_spec(synthetic.QUAD, (-12.0, 12.0, 4.0), (math.pi, 0.0, 0.0), (40, 40, 40), (24.0, 24.0)),
    ]
    for strategy, heuristic in ((0, 0), (3, 3), (3, 4), (2, 0), (4, 0)):
        for arithmetic in ("libm", "exact"):
            stats, _, _ = render_and_compare(dict(lights=lights, strategy=strategy, heuristic=heuristic, samples=2, rays=True), dataset, arithmetic=arithmetic)
            assert stats["nan"] == 0 and stats["bit_exact"], (strategy, heuristic, arithmetic, stats)


def test_many_lights_and_many_samples(dataset):
    rng = np.random.default_rng(5)
    many_lights = [random_light(rng, 3 + i % 5) for i in range(24)]
    stats, _, rays = render_and_compare(dict(lights=many_lights, strategy=3, heuristic=3, samples=1, rays=True), dataset, 48, 32)
    assert stats["bit_exact"] and rays > 0, stats
    stats, _, _ = render_and_compare(dict(lights=many_lights[:1], strategy=3, heuristic=2, samples=64, rays=True), dataset, 48, 32)
    assert stats["bit_exact"], stats
    stats, _, _ = render_and_compare(dict(lights=many_lights[:2], strategy=0, samples=128, technique="solid_angle"), dataset, 32, 16)
    assert stats["bit_exact"], stats


@pytest.mark.parametrize("size", [(1, 1), (7, 3), (17, 33)])
def test_frames_smaller_than_a_workgroup(size, dataset):
    stats, image, _ = render_and_compare(dict(lights=golden_cases.QUAD, strategy=1, heuristic=0, rays=True), dataset, size[0], size[1])
    assert image.shape == (size[1], size[0], 4) and stats["bit_exact"], stats


@pytest.fixture(scope="module")
def textured_dataset(tmp_path_factory):
    return synthetic.write_dataset(str(tmp_path_factory.mktemp("sweep_textured")), **golden_cases.TEXTURED_DATASET)


@pytest.mark.parametrize("arithmetic", ["libm", "exact"])
@pytest.mark.parametrize("seed", range(100, 100 + max(16, int(os.environ.get("VKR_SWEEP_SEEDS", "48")) // 3)))
def test_random_configuration_with_textures_is_bit_exact(seed, arithmetic, dataset, textured_dataset):
    """The sweep once more with a random texturing technique per light (area, light probe, IES
    profile or none) and, for every second seed, textured materials."""
    rng = np.random.default_rng(seed)
    data = textured_dataset if seed % 2 else dataset
    case = random_case(seed)
    names = {"area": "area", "portal": "portal", "ies_profile": "ies"}
    for light in case["lights"]:
        technique = ["none", "area", "portal", "ies_profile"][int(rng.integers(0, 4))]
        if technique != "none":
            light["texturing_technique"] = technique
            light["texture_file_path"] = data["light_textures"][names[technique] if rng.random() < 0.8 else "portal_rgb16"]
    stats, image, _ = render_and_compare(case, data, frames_in_flight=1 + seed % 3, arithmetic=arithmetic)
    summary = {k: case[k] for k in ("strategy", "technique", "heuristic", "samples", "rays", "show_lights", "error_display")}
    summary["texturing"] = [l.get("texturing_technique", "none") for l in case["lights"]]
    assert stats["nan"] == 0 and stats["bit_exact"], (summary, stats)


def test_block_wise_ray_queue_counts_the_rays_it_traces(dataset):
    """Eight rays per lane and more switch the two-technique strategies to block-wise slot
    reservation (null rays pad the blocks): the count that is reported must be the real rays,
    i.e. what a walk over the queues finds when it skips the null rays."""
    for samples, lights in ((2, golden_cases.QUADS), (1, golden_cases.QUADS[:1])):  # 16 rays per lane: blocks; 2: exact
        r = renderer.Renderer()
        golden_cases.apply_case(r, dict(lights=lights, strategy=3, heuristic=3, samples=samples, rays=True), dataset, 160, 96)
        r.create_targets()
        r.create_pass()
        r.render_visibility()
        r.render()
        reported = r.last_ray_count()
        walked = r.traversal_statistics()
        r.close()
        assert reported > 0 and walked["rays"] == reported, (samples, reported, walked)
