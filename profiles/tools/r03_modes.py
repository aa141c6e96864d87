#!/usr/bin/env python3
"""Round 3, first GPU run: the three arithmetic modes of the pass at BASELINE size.
For every mode: frame period (three frames in flight), kernels alone, and the frame against the
CPU oracle in both of its math modes (0 = libm, the one pinned against the reference shader; 1 =
polynomial).  Writes one JSON object per (config, mode) line.
    python profiles/tools/r03_modes.py [configs ...]"""
import json
import os
import sys
import tempfile
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import oracle
from vulkan_renderer_amd import renderer, synthetic


def compare(gpu, cpu):
    d = gpu[..., :3].astype(np.float64) - cpu[..., :3].astype(np.float64)
    d = np.nan_to_num(d, nan=1.0e3)
    per_pixel = np.abs(d).max(axis=-1)
    bits = (gpu[..., :3].view(np.uint32) != cpu[..., :3].astype(np.float32).view(np.uint32)).any(axis=-1)
    return {"rmse": float(np.sqrt((d ** 2).mean())), "pixels_differing_in_bits": int(bits.sum()), "pixels_over_1e-2": int((per_pixel > 1e-2).sum()),
            "rmse_without_those": float(np.sqrt((d[per_pixel <= 1e-2] ** 2).sum() / d.size)), "max_abs": float(per_pixel.max())}


def main():
    modes = os.environ.get("VKR_MODES", "libm,exact,fast").split(",")
    configs = [int(c) if c != "target" else c for c in sys.argv[1:]] or [2, 3]
    with tempfile.TemporaryDirectory() as tmp:
        dataset = synthetic.write_dataset(tmp, grid=256, box_count=64, seed=1234, ltc_resolution=64, fresnel_count=51)
        for config in configs:
            oracle_frames = {}
            for mode in modes:
                r = renderer.Renderer(arithmetic=mode, frames_in_flight=3, timing_stride=1)
                renderer.setup_config(r, config, dataset)
                r.create_targets()
                r.create_pass()
                r.render_visibility()
                for _ in range(60):
                    r.render()
                r.finish_frames(); r.sync()
                steps = 200 if config != 4 else 20
                t0 = time.perf_counter()
                for _ in range(steps):
                    r.render()
                r.finish_frames(); r.sync()
                period = (time.perf_counter() - t0) / steps * 1e3
                image = r.read_radiance()
                visibility = r.read_visibility()
                out = {"config": config, "mode": mode, "library": os.path.basename(os.environ.get("VKR_SHADING_LIBRARY", "libvkr_shading.so")), "ms_per_frame": round(period, 4), "rays": r.last_ray_count()}
                r.frames_in_flight = 1
                r.create_pass()
                for _ in range(12):
                    r.render()
                r.sync()
                out["shade_alone_ms"] = round(float(np.mean(r.shading_kernel_ms(8))), 4)
                out["pass_alone_ms"] = round(float(np.mean(r.dispatch_ms(8))), 4)
                if not oracle_frames and config != 4:
                    inputs = r.host_inputs(visibility)
                    bvh = oracle.Bvh(inputs["quantized_positions"], inputs["dequantization_factor"], inputs["dequantization_summand"])
                    frame = oracle.make_frame(inputs, r.oracle_settings(), bvh)
                    for m in (0, 1):
                        oracle.set_math_mode(m)
                        oracle_frames[m] = oracle.shade(frame)
                    oracle.set_math_mode(0)
                    out["oracle_libm_vs_polynomial"] = compare(oracle_frames[0].astype(np.float32), oracle_frames[1])
                for m, name in ((0, "vs_oracle_libm"), (1, "vs_oracle_polynomial")):
                    if m in oracle_frames:
                        out[name] = compare(image, oracle_frames[m])
                r.close()
                print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
