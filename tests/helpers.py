"""Shared helpers of the test-suite (not collected by pytest)."""
import numpy as np

import oracle
from vulkan_renderer_amd import renderer, synthetic


def oracle_render(host_scene, visibility=None, math_mode=1, bvh=None, width=None, height=None):
    """Shades the host scene with the CPU oracle.  Returns (image, inputs, bvh)."""
    inputs = host_scene.host_inputs()
    e = host_scene.app.swapchain.extent
    if bvh is None:
        bvh = oracle.Bvh(inputs["quantized_positions"], inputs["dequantization_factor"], inputs["dequantization_summand"])
    if visibility is None:
        cam = host_scene.app.scene_specification.camera
        visibility = oracle.primary_visibility(inputs["constants"], bvh, e.width, e.height, cam.near, cam.far)
    inputs["visibility"] = visibility
    frame = oracle.make_frame(inputs, host_scene.oracle_settings(), bvh)
    oracle.set_math_mode(math_mode)
    try:
        image = oracle.shade(frame)
    finally:
        oracle.set_math_mode(0)
    return image, inputs, bvh


def compare(gpu, cpu):
    """Error statistics over RGB of exposure-scaled radiance."""
    a, b = gpu[..., :3].astype(np.float64), cpu[..., :3].astype(np.float64)
    diff = np.abs(a - b)
    per_pixel = diff.max(axis=-1)
    return {
        "rmse": float(np.sqrt(((a - b) ** 2).mean())),
        "max_abs": float(diff.max()),
        "mismatched_pixels": int((per_pixel > 0).sum()),
        "pixels_over_1e-3": int((per_pixel > 1e-3).sum()),
        "nan": int(np.isnan(gpu).sum()),
        "bit_exact": bool(np.array_equal(gpu.view(np.uint32), cpu.view(np.uint32))),
    }
