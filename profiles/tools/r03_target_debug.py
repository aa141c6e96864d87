#!/usr/bin/env python3
"""Which pixels of a full-size frame differ between the GPU (libm mode) and the oracle (libm mode)?
    [VKR_SHADING_LIBRARY=...] python profiles/tools/r03_target_debug.py [config ...]"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import oracle
from vulkan_renderer_amd import renderer, synthetic

configs = [c if c == "target" else int(c) for c in sys.argv[1:]] or ["target"]
with tempfile.TemporaryDirectory() as tmp:
    dataset = synthetic.write_dataset(tmp, grid=256, box_count=64, seed=1234, ltc_resolution=64, fresnel_count=51)
    for config in configs:
        r = renderer.Renderer(arithmetic="libm", timing_stride=1)
        renderer.setup_config(r, config, dataset)
        r.create_targets(); r.create_pass(); r.render_visibility()
        for _ in range(20):
            r.render()
        r.sync()
        image = r.read_radiance()
        shade_ms = float(np.mean(r.shading_kernel_ms(8)))
        inputs = r.host_inputs(r.read_visibility())
        bvh = oracle.Bvh(inputs["quantized_positions"], inputs["dequantization_factor"], inputs["dequantization_summand"])
        frame = oracle.make_frame(inputs, r.oracle_settings(), bvh)
        cpu = oracle.shade(frame)
        r.close()
        differ = (image[..., :3].view(np.uint32) != cpu[..., :3].astype(np.float32).view(np.uint32)).any(axis=-1)
        yx = np.argwhere(differ)
        print(json.dumps({"config": config, "library": os.environ.get("VKR_SHADING_LIBRARY", "default"), "shade_alone_ms": round(shade_ms, 4), "pixels_differing": int(differ.sum()),
                          "examples": [{"yx": [int(y), int(x)], "gpu": [float(v) for v in image[y, x, :3]], "cpu": [float(v) for v in cpu[y, x, :3]]} for y, x in yx[:6]]}), flush=True)
