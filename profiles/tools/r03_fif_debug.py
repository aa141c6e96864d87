#!/usr/bin/env python3
"""Frames in flight x number of frames x arithmetic: pixels differing from the oracle (debugging aid)."""
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

config = sys.argv[1] if len(sys.argv) > 1 else "target"
config = config if config == "target" else int(config)
with tempfile.TemporaryDirectory() as tmp:
    dataset = synthetic.write_dataset(tmp, grid=256, box_count=64, seed=1234, ltc_resolution=64, fresnel_count=51)
    cpu = {}
    for arithmetic in ("libm", "exact"):
        for fif in (1, 2, 3):
            for frames in (1, 2, 3, 7):
                r = renderer.Renderer(arithmetic=arithmetic, frames_in_flight=fif)
                renderer.setup_config(r, config, dataset, acceleration_structure="sah_device")
                r.create_targets(); r.create_pass(); r.render_visibility()
                for _ in range(frames):
                    r.render()
                image = r.read_radiance()
                if arithmetic not in cpu:
                    inputs = r.host_inputs(r.read_visibility())
                    bvh = oracle.Bvh(inputs["quantized_positions"], inputs["dequantization_factor"], inputs["dequantization_summand"])
                    frame = oracle.make_frame(inputs, r.oracle_settings(), bvh)
                    oracle.set_math_mode(renderer.ORACLE_MATH_MODE[arithmetic])
                    cpu[arithmetic] = oracle.shade(frame)
                    oracle.set_math_mode(0)
                r.close()
                differ = (image[..., :3].view(np.uint32) != cpu[arithmetic][..., :3].astype(np.float32).view(np.uint32)).any(axis=-1)
                yx = np.argwhere(differ)
                print(json.dumps({"config": config, "arithmetic": arithmetic, "frames_in_flight": fif, "frames": frames, "pixels_differing": int(differ.sum()),
                                  "rows": [int(yx[:, 0].min()), int(yx[:, 0].max())] if len(yx) else None, "columns": [int(yx[:, 1].min()), int(yx[:, 1].max())] if len(yx) else None}), flush=True)
