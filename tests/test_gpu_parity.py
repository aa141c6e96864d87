"""Parity of the HIP shading pass with the CPU oracle on identical inputs.

The libm arithmetic mode (the default) must agree bit for bit with the oracle's libm
mode - the arithmetic that tests/test_reference_live.py pins against the reference's shader
source -, the polynomial "exact" mode with the oracle's polynomial mode; fast mode must stay
within the tolerance BASELINE.json states (RMSE <= 1e-4 on exposure-scaled linear radiance)
wherever no pixel sits on a discontinuity of the shader (tests/test_gpu_full_size.py has the
rule for the full-size frames, where such pixels exist)."""
import numpy as np
import pytest

from helpers import compare, oracle_render
from vulkan_renderer_amd import renderer

pytestmark = pytest.mark.gpu

RMSE_TOLERANCE = 1.0e-4


def gpu_render(dataset, config, width, height, arithmetic, inline_rays=False, **overrides):
    r = renderer.Renderer(arithmetic=arithmetic, inline_rays=inline_rays)
    renderer.setup_config(r, config, dataset, width=width, height=height, acceleration_structure=True, **overrides)
    r.create_targets()
    r.create_pass()
    r.render_visibility()
    visibility = r.read_visibility()
    r.render()
    image = r.read_radiance()
    return r, image, visibility


@pytest.mark.parametrize("inline_rays", [False, True], ids=["wavefront", "inline"])
@pytest.mark.parametrize("arithmetic", ["libm", "exact"])
@pytest.mark.parametrize("config", [1, 2, 3])
def test_ieee_modes_match_their_oracle_bitwise(dataset, config, arithmetic, inline_rays):
    r, image, visibility = gpu_render(dataset, config, 256, 144, arithmetic, inline_rays=inline_rays)
    cpu, inputs, bvh = oracle_render(r, visibility=visibility, math_mode=renderer.ORACLE_MATH_MODE[arithmetic])
    stats = compare(image, cpu)
    r.close()
    print(config, arithmetic, stats)
    assert stats["nan"] == 0
    assert stats["bit_exact"], stats


@pytest.mark.parametrize("arithmetic", ["libm", "exact"])
def test_config_1_at_its_stated_size_matches_its_oracle_bitwise(dataset, arithmetic):
    """BASELINE configs[0] as it is worded: 512x512, 1 spp, one triangle light, diffuse-only LTC, no ray visibility"""
    r, image, visibility = gpu_render(dataset, 1, 512, 512, arithmetic)
    assert not r.app.shading_pass.use_ray_tracing and r.app.scene_specification.polygonal_light_count == 1
    cpu, inputs, bvh = oracle_render(r, visibility=visibility, math_mode=renderer.ORACLE_MATH_MODE[arithmetic])
    stats = compare(image, cpu)
    r.close()
    assert (visibility != 0xFFFFFFFF).mean() > 0.3
    assert stats["nan"] == 0 and stats["bit_exact"], stats


@pytest.mark.parametrize("config", [1, 2, 3])
def test_fast_mode_within_tolerance(dataset, config):
    r, image, visibility = gpu_render(dataset, config, 256, 144, "fast")
    cpu, inputs, bvh = oracle_render(r, visibility=visibility, math_mode=0)
    stats = compare(image, cpu)
    r.close()
    print(config, stats)
    assert stats["nan"] == 0
    assert stats["rmse"] <= RMSE_TOLERANCE, stats


def test_primary_visibility_matches_oracle(dataset):
    import oracle
    r = renderer.Renderer()
    renderer.setup_config(r, 2, dataset, width=320, height=180, acceleration_structure=True)
    r.create_targets()
    r.create_pass()
    r.render_visibility()
    gpu = r.read_visibility()
    inputs = r.host_inputs()
    bvh = oracle.Bvh(inputs["quantized_positions"], inputs["dequantization_factor"], inputs["dequantization_summand"])
    cam = r.app.scene_specification.camera
    cpu = oracle.primary_visibility(inputs["constants"], bvh, 320, 180, cam.near, cam.far)
    r.close()
    assert (gpu != 0xFFFFFFFF).mean() > 0.3
    assert np.array_equal(gpu, cpu), "%d pixels differ" % int((gpu != cpu).sum())


@pytest.mark.parametrize("rays", [True, False], ids=["rays", "no_rays"])
def test_polygon_tables_in_device_memory_by_workgroup_and_by_hardware_wave_slot(dataset, monkeypatch, rays):
    """From V = 6 on the two-technique kernels keep one of their two polygon tables in device memory: in the
    region of the workgroup (default) or of the hardware slot the wave runs in (VKR_PSA_TABLE_INDEX=slot).  Config 4's lights
    (3 ... 6 vertices: V = 7) with and without wavefront rays: both schemes give the oracle's frame."""
    frames = {}
    for index in ("block", "slot"):
        monkeypatch.setenv("VKR_PSA_TABLE_INDEX", index)
        r, image, visibility = gpu_render(dataset, 4, 320, 180, "libm", sample_count=2, trace_shadow_rays=rays)
        assert r.app.shading_pass.max_polygon_vertex_count == 7
        if index == "block":
            cpu, _, _ = oracle_render(r, visibility=visibility, math_mode=0)
        frames[index] = image
        r.close()
    monkeypatch.delenv("VKR_PSA_TABLE_INDEX")
    assert compare(frames["block"], cpu)["bit_exact"]
    assert np.array_equal(frames["block"].view(np.uint32), frames["slot"].view(np.uint32))


@pytest.mark.parametrize("config", [2, 3])
def test_the_order_of_the_blocks_changes_no_pixel(dataset, config):
    """With one rank that renders in place the tile size only decides which 16x16 blocks are consecutive in the launch (0 = the
    automatic choice, 64 or 128 by frame size): same frame, same rays, for a frame whose size is no multiple of
    any tile."""
    frames = {}
    for tile in (0, 16, 32, 64, 128):
        r = renderer.Renderer(frames_in_flight=2)
        renderer.setup_config(r, config, dataset, width=456, height=200, acceleration_structure=True)
        r.set_tiles(tile, 0, 1)
        r.create_targets()
        r.create_pass()
        r.render_visibility()
        r.render()
        r.render()
        frames[tile] = (r.read_radiance(), r.last_ray_count())
        r.close()
    for tile, (image, rays) in frames.items():
        assert np.array_equal(image.view(np.uint32), frames[16][0].view(np.uint32)) and rays == frames[16][1], tile
