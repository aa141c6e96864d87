"""Definition of the golden frame cases shared by tests/golden/make_golden.py (which
renders them with the reference's shader) and the tests (which render them with
the oracle and the HIP kernels).  Everything is derived from seeds, so the inputs
are reproduced byte for byte wherever the tests run."""
import math

import numpy as np

import oracle
from oracle import reference
from vulkan_renderer_amd import renderer, synthetic

DATASET = dict(grid=48, box_count=12, seed=1234, ltc_resolution=16, fresnel_count=8)
WIDTH, HEIGHT = 64, 36

PI = math.pi
TRIANGLE = [synthetic.light_spec([(0, 0), (1, 0), (0, 1)], (-1.0, 0.5, 3.0), (PI, 0.0, 0.0), (10, 10, 10), (1.5, 1.5))]
PENTAGON = [synthetic.light_spec(synthetic.regular_polygon(5), (-0.5, 1.0, 2.5), (0.85 * PI, 0.1, 0.3), (12, 11, 10), (1.6, 1.6))]
QUADS = synthetic.config_lights(3)
MIXED = [
    synthetic.light_spec([(0, 0), (1, 0), (0.3, 1)], (-2.0, 0.0, 1.2), (0.55 * PI, 0.2, 0.9), (7, 5, 4), (1.2, 1.0)),
    synthetic.light_spec(synthetic.regular_polygon(6, 0.5, 0.3), (0.5, 2.5, 2.8), (0.95 * PI, -0.1, 0.0), (9, 9, 12), (1.4, 1.1)),
    synthetic.light_spec(synthetic.regular_polygon(5, 0.5, 1.0), (-3.5, 3.0, 0.6), (0.5 * PI, 0.0, 2.2), (5, 8, 5), (0.9, 0.9)),
]
HEPTAGON = [synthetic.light_spec(synthetic.regular_polygon(7, 0.5, 0.1), (-1.5, 1.5, 1.4), (0.6 * PI, 0.15, 1.1), (14, 12, 9), (1.8, 1.3))]
QUAD = [synthetic.light_spec(synthetic.QUAD, (-1.0, 1.0, 2.0), (0.8 * PI, 0.1, 0.5), (10, 9, 8), (1.3, 1.0))]

# strategy / heuristic numbers: reference src/main.h:45-89
FRAME_CASES = [
    dict(key="cfg1_diffuse_only", lights=TRIANGLE, strategy=0, heuristic=0, samples=1),
    dict(key="cfg2_ggx_mis_rays", lights=PENTAGON, strategy=1, heuristic=0, samples=1, rays=True),
    dict(key="ggx_mis_power", lights=PENTAGON, strategy=1, heuristic=1, samples=2),
    dict(key="cfg3_mis_clamped_rays", lights=QUADS, strategy=3, heuristic=3, samples=2, rays=True),
    dict(key="mis_balance_mixed", lights=MIXED, strategy=3, heuristic=0, samples=1),
    dict(key="mis_power_mixed", lights=MIXED, strategy=3, heuristic=1, samples=1),
    dict(key="mis_weighted_mixed", lights=MIXED, strategy=3, heuristic=2, samples=1),
    dict(key="mis_optimal_mixed", lights=MIXED, strategy=3, heuristic=4, samples=1),
    dict(key="separately_mixed", lights=MIXED, strategy=2, heuristic=0, samples=1),
    dict(key="random_mixed", lights=MIXED, strategy=4, heuristic=0, samples=2),
    dict(key="heptagon_show_lights", lights=HEPTAGON, strategy=3, heuristic=3, samples=1, show_lights=True),
    dict(key="biased_psa", lights=QUAD, strategy=0, heuristic=0, samples=1, technique="projected_solid_angle_biased"),
    dict(key="solid_angle", lights=QUAD, strategy=0, heuristic=0, samples=1, technique="solid_angle"),
    dict(key="clipped_solid_angle_ggx", lights=QUAD, strategy=1, heuristic=0, samples=1, technique="clipped_solid_angle"),
    dict(key="baseline_technique", lights=MIXED, strategy=0, heuristic=0, samples=2, technique="baseline"),
    dict(key="area_turk_rays", lights=MIXED, strategy=0, heuristic=0, samples=2, technique="area_turk", rays=True),
    dict(key="urena_rectangle_ggx", lights=QUAD, strategy=1, heuristic=0, samples=2, technique="rectangle_solid_angle_urena"),
    dict(key="arvo_solid_angle_mixed", lights=MIXED, strategy=0, heuristic=0, samples=1, technique="solid_angle_arvo"),
    dict(key="arvo_solid_angle_ggx", lights=PENTAGON, strategy=1, heuristic=1, samples=2, technique="solid_angle_arvo"),
    dict(key="hart_bilinear", lights=MIXED, strategy=0, heuristic=0, samples=2, technique="bilinear_cosine_warp_hart"),
    dict(key="hart_bilinear_clipping_rays", lights=MIXED, strategy=0, heuristic=0, samples=2, technique="bilinear_cosine_warp_clipping_hart", rays=True),
    dict(key="hart_biquadratic", lights=MIXED, strategy=0, heuristic=0, samples=2, technique="biquadratic_cosine_warp_hart"),
    dict(key="hart_biquadratic_clipping_rays", lights=MIXED, strategy=0, heuristic=0, samples=2, technique="biquadratic_cosine_warp_clipping_hart", rays=True),
    dict(key="arvo_psa_mixed", lights=MIXED, strategy=0, heuristic=0, samples=2, technique="projected_solid_angle_arvo"),
    dict(key="arvo_psa_ggx_rays", lights=PENTAGON, strategy=1, heuristic=0, samples=1, technique="projected_solid_angle_arvo", rays=True),
    dict(key="arvo_psa_error_backward", lights=MIXED, strategy=0, heuristic=0, samples=1, technique="projected_solid_angle_arvo", error_display=1),
    dict(key="error_backward_diffuse_only", lights=MIXED, strategy=0, heuristic=0, samples=1, error_display=1),
    dict(key="error_backward_scaled_mis", lights=MIXED, strategy=3, heuristic=3, samples=1, error_display=2),
    dict(key="error_forward_specular_mis", lights=MIXED, strategy=3, heuristic=3, samples=1, error_display=6),
    dict(key="cfg1_srgb_encoded", lights=TRIANGLE, strategy=0, heuristic=0, samples=1, output_linear_rgb=False),
]


# Light textures: every texturing technique of
# get_polygon_radiance, shading_pass.frag.glsl:151-185.  "texture" names a file of
# synthetic.write_light_textures; the texel filter is this build's (unpinned by the reference, whose
# driver filters), everything around it -- plane-space, probe and IES coordinates, the cosine
# division, the light display -- goes through the reference's shader.
def textured_light(light, technique, texture):
    return dict(light, texturing_technique=technique, texture=texture)


TEXTURED_MIXED = [textured_light(MIXED[0], "area", "area"), textured_light(MIXED[1], "portal", "portal"), textured_light(MIXED[2], "ies_profile", "ies")]
LIGHT_TEXTURE_CASES = [
    dict(key="light_textures_mis_show_lights", lights=TEXTURED_MIXED, strategy=3, heuristic=3, samples=1, show_lights=True),
    dict(key="light_textures_mis_balance", lights=TEXTURED_MIXED, strategy=3, heuristic=0, samples=1),
    dict(key="light_textures_ggx_rays", lights=[textured_light(PENTAGON[0], "area", "area")], strategy=1, heuristic=0, samples=1, rays=True),
    dict(key="light_textures_separately", lights=TEXTURED_MIXED, strategy=2, heuristic=0, samples=1),
    dict(key="light_textures_random", lights=TEXTURED_MIXED, strategy=4, heuristic=0, samples=2),
    dict(key="light_textures_probe_rgb16", strategy=0, heuristic=0, samples=1,
         lights=[textured_light(TRIANGLE[0], "portal", "portal_rgb16")]),
]


def apply_case(scene, case, dataset, width=WIDTH, height=HEIGHT):
    """Loads the dataset into a HostScene / Renderer and applies the case."""
    rays = bool(case.get("rays", False))
    if any("texture" in light for light in case["lights"]):
        case = dict(case, lights=[dict(light, texture_file_path=dataset["light_textures"][light["texture"]]) if "texture" in light else light for light in case["lights"]])
    scene.load_scene(dataset["scene"], dataset["textures"], acceleration_structure=True if scene._device else False)
    scene.load_ltc_table(dataset["ltc"], dataset["fresnel_count"])
    scene.load_noise_table("white")
    cam = synthetic.DEFAULT_CAMERA
    scene.set_camera(cam["position"], cam["rotation_x"], cam["rotation_z"], cam["vertical_fov"], cam["near"], cam["far"])
    scene.set_lights(case["lights"])
    scene.set_settings(width=width, height=height, sample_count=case["samples"], sampling_strategies=case["strategy"],
                       mis_heuristic=case["heuristic"], polygon_technique=case.get("technique", "projected_solid_angle"),
                       trace_shadow_rays=rays, show_polygonal_lights=bool(case.get("show_lights", False)),
                       error_display=case.get("error_display", 0), error_min_exponent=-7.0)


def reference_variant(case):
    counts = [len(l["vertices_plane_space"]) for l in case["lights"]]
    return reference.variant_name(strategy=case["strategy"], heuristic=case["heuristic"],
                                  technique=case.get("technique", "projected_solid_angle"), lights=len(counts),
                                  min_light_vertices=min(counts), max_light_vertices=max(counts), samples=case["samples"],
                                  rays=case.get("rays", False), show_lights=case.get("show_lights", False),
                                  output_linear_rgb=case.get("output_linear_rgb", True), error_display=case.get("error_display", 0))


def build_frame(case, dataset, width=WIDTH, height=HEIGHT):
    """Returns (host scene, oracle frame incl. visibility buffer and BVH, reference variant name)."""
    hs = renderer.HostScene()
    apply_case(hs, case, dataset, width, height)
    inputs = hs.host_inputs()
    bvh = oracle.Bvh(inputs["quantized_positions"], inputs["dequantization_factor"], inputs["dequantization_summand"])
    cam = synthetic.DEFAULT_CAMERA
    inputs["visibility"] = oracle.primary_visibility(inputs["constants"], bvh, width, height, cam["near"], cam["far"])
    frame = oracle.make_frame(inputs, hs.oracle_settings(), bvh)
    return hs, frame, reference_variant(case)


def capacity_variant(n):
    """A built reference variant whose MAX_POLYGON_VERTEX_COUNT is n + 1 (for sub-function vectors)."""
    table = {3: reference.variant_name(strategy=0, lights=1, max_light_vertices=3, samples=1),
             4: reference.variant_name(strategy=3, heuristic=3, lights=4, max_light_vertices=4, samples=2, rays=True),
             5: reference.variant_name(strategy=1, heuristic=1, lights=1, max_light_vertices=5, samples=2),
             6: reference.variant_name(strategy=3, heuristic=0, lights=3, min_light_vertices=3, max_light_vertices=6, samples=1),
             7: reference.variant_name(strategy=3, heuristic=3, lights=1, max_light_vertices=7, samples=1, show_lights=True)}
    return table[n]


def random_polygon(rng, n):
    """A convex planar n-gon in front of / around the horizon of the origin."""
    normal = rng.normal(size=3)
    normal /= np.linalg.norm(normal)
    t = np.cross(normal, [0.31, 0.52, 0.79])
    t /= np.linalg.norm(t)
    b = np.cross(normal, t)
    while True:
        ang = np.sort(rng.uniform(0, 2 * np.pi, n))
        gaps = np.diff(np.concatenate([ang, [ang[0] + 2 * np.pi]]))
        if gaps.min() > 0.35 and gaps.max() < np.pi * 0.95:
            break
    center = rng.normal(size=3) * 1.2
    center[2] = abs(center[2]) + rng.uniform(-0.6, 1.2)
    radius = rng.uniform(0.3, 1.8)
    return np.array([center + radius * (np.cos(a) * t + np.sin(a) * b) for a in ang], np.float32)


# Textured scenes: the same reference variants, material textures as real images (BC1 sRGB base
# colour, RGBA8 specular, BC5 normal with mip chains).  The texture filter itself is the
# driver's in the reference; the compiled reference shader calls the oracle's sampler for it
# (oracle.h), so these frames pin everything around the filter: derivatives, coordinates,
# how the texels enter the BRDF parameters.
TEXTURED_DATASET = dict(DATASET, textured=True, texture_size=32)
TEXTURED_CASES = [
    dict(key="textured_cfg2_ggx_mis_rays", lights=PENTAGON, strategy=1, heuristic=0, samples=1, rays=True),
    dict(key="textured_mis_clamped", lights=QUADS, strategy=3, heuristic=3, samples=2, rays=True),
    dict(key="textured_diffuse_only", lights=TRIANGLE, strategy=0, heuristic=0, samples=1),
]
