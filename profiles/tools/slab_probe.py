#!/usr/bin/env python3
"""One rank's slab of an N-rank tiling, rendered alone on one GPU (what profiles/tools/predict_scaling.py times), as a
command that rocprofv3 can wrap:

    python profiles/tools/slab_probe.py --config 3 --ranks 8 --rank 0 --tile 32 --frames-in-flight 3 --steps 200
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/x -o trace -- python profiles/tools/slab_probe.py ...

Prints one JSON line: ms per frame of the slab, of the whole frame, and the ratio."""
import argparse
import json
import os
import sys
import tempfile
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import DeviceBuffer
from vulkan_renderer_amd import renderer, synthetic


def time_frames(r, target, steps):
    for _ in range(max(8, steps // 4)):
        r.render(target)
    r.finish_frames(); r.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        r.render(target)
    r.finish_frames(); r.sync()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="3")
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--tile", type=int, default=32)
    ap.add_argument("--frames-in-flight", type=int, default=3)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--mode", default="libm")
    ap.add_argument("--whole", action="store_true", help="also time the whole frame")
    args = ap.parse_args()
    config = args.config if args.config == "target" else int(args.config)
    with tempfile.TemporaryDirectory() as tmp:
        dataset = synthetic.write_dataset(tmp, grid=256, box_count=64, seed=1234, ltc_resolution=64, fresnel_count=51)
        r = renderer.Renderer(arithmetic=args.mode, frames_in_flight=args.frames_in_flight, timing_stride=64)
        renderer.setup_config(r, config, dataset)
        r.set_tiles(16, 0, 1, slab_layout=False)
        r.create_targets(); r.create_pass(); r.render_visibility()
        whole_ms = time_frames(r, None, args.steps) if args.whole else None
        out = {"config": args.config, "ranks": args.ranks, "tile": args.tile, "frames_in_flight": args.frames_in_flight, "whole_frame_ms": whole_ms and round(whole_ms, 4)}
        r.set_tiles(args.tile, args.rank, args.ranks, slab_layout=args.ranks > 1)
        slab = DeviceBuffer(r.slab_pixel_count(args.rank) * 16)
        slab_ms = time_frames(r, slab.ptr.value, args.steps)
        out["slab_ms"] = round(slab_ms, 4)
        out["rays"] = r.last_ray_count()
        if whole_ms:
            out["speedup_if_all_ranks_alike"] = round(whole_ms / slab_ms, 3)
        r.sync()
        slab.free()
        r.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
