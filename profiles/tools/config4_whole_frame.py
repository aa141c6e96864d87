#!/usr/bin/env python3
"""BASELINE config 4 (3840x2160, 8 spp per technique, 8 polygonal lights, shadow rays) - the WHOLE frame of the
kernels against the whole frame of the reference-pinned oracle (libm mode), every pixel, bit for bit.
The oracle needs a few minutes on the box's host cores; the record goes to gpurun_out/<name>/config4_whole_frame.json.

  gpurun --timeout 900 -- 'python profiles/tools/config4_whole_frame.py r05a'"""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
from vulkan_renderer_amd import renderer, synthetic  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "config4"
    out_dir = os.path.join(ROOT, "gpurun_out", name)
    os.makedirs(out_dir, exist_ok=True)
    with tempfile.TemporaryDirectory() as d:
        dataset = synthetic.write_dataset(d, grid=256, box_count=64, seed=1234, ltc_resolution=64, fresnel_count=51)
        r = renderer.Renderer(frames_in_flight=3)
        renderer.setup_config(r, 4, dataset, acceleration_structure="sah_device")
        r.create_targets()
        r.create_pass()
        r.render_visibility()
        r.render()
        r.render()
        gpu = r.read_radiance()
        visibility = r.read_visibility()
        rays = r.last_ray_count()
        bands = int(r.app.shading_pass.last_band_count)
        inputs = r.host_inputs(visibility)
        settings = r.oracle_settings()
        r.close()
    height, width = gpu.shape[:2]
    bvh = oracle.Bvh(inputs["quantized_positions"], inputs["dequantization_factor"], inputs["dequantization_summand"])
    frame = oracle.make_frame(inputs, settings, bvh)
    oracle.set_math_mode(0)
    cores = len(os.sched_getaffinity(0))
    t = time.perf_counter()
    cpu = np.zeros_like(gpu)
    step = 120
    for y0 in range(0, height, step):
        cpu[y0:y0 + step] = oracle.shade(frame, y0, min(height, y0 + step))[y0:y0 + step]
    seconds = time.perf_counter() - t
    differing = int((gpu[..., :3].view(np.uint32) != cpu[..., :3].view(np.uint32)).any(axis=-1).sum())
    record = {"workload": "BASELINE config 4: %dx%d, 8 spp per technique, 8 polygonal lights, clamped optimal MIS, shadow rays, libm arithmetic" % (width, height),
              "pixels": int(width * height), "pixels_compared": int(width * height), "pixels_differing_in_bits": differing,
              "bit_exact_rgba": bool(np.array_equal(gpu.view(np.uint32), cpu.view(np.uint32))),
              "shaded_fraction": float((visibility != 0xFFFFFFFF).mean()), "nan": int(np.isnan(gpu).sum()),
              "shadow_rays": int(rays), "bands_per_frame": bands, "oracle_seconds": round(seconds, 1), "oracle_threads": cores,
              "oracle": oracle.libm_description()}
    json.dump(record, open(os.path.join(out_dir, "config4_whole_frame.json"), "w"), indent=1)
    print(json.dumps(record))
    return 0 if differing == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
