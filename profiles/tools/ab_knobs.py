#!/usr/bin/env python3
"""A/B of run-time knobs of the shading pass on ONE workload in ONE process (one box, one dataset, one BVH): every
setting is timed `--rounds` times in turn (A B C A B C ...), frames pipelined as in bench.py.

    python profiles/tools/ab_knobs.py --scene large --config 3 --set VKR_WIDE_REFILL=0 --set VKR_WIDE_REFILL=16 --set VKR_WIDE_REFILL=16,VKR_SHAFT_REST=0
    python profiles/tools/ab_knobs.py --config 3 --ranks 8 --tile 32 --set fif=3 --set fif=6 --set fif=6,VKR_SHAFT_MAX_STEPS=16

A setting is a comma-separated list of ENV=value pairs; `fif=n` sets the frames in flight of the pass.  With --ranks N
the slab of rank --rank is rendered alone (profiles/tools/predict_scaling.py does that for every rank).
Prints one JSON line per setting: ms per frame of every round, their minimum and median, rays per frame."""
import argparse
import json
import os
import statistics
import sys
import tempfile
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import DeviceBuffer
from vulkan_renderer_amd import renderer, synthetic


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="3")
    ap.add_argument("--scene", default="bench", choices=["bench", "large"])
    ap.add_argument("--ranks", type=int, default=1)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--tile", type=int, default=32)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--mode", default="libm")
    ap.add_argument("--set", action="append", default=[], dest="settings")
    args = ap.parse_args()
    config = args.config if args.config == "target" else int(args.config)
    settings = args.settings or [""]
    touched = sorted({pair.split("=")[0] for s in settings for pair in s.split(",") if pair and not pair.startswith("fif=")})
    with tempfile.TemporaryDirectory() as tmp:
        if args.scene == "large":
            dataset = synthetic.write_dataset(tmp, seed=4321, ltc_resolution=64, fresnel_count=51, large={})
        else:
            dataset = synthetic.write_dataset(tmp, grid=256, box_count=64, seed=1234, ltc_resolution=64, fresnel_count=51)
        r = renderer.Renderer(arithmetic=args.mode, frames_in_flight=3, timing_stride=64)
        renderer.setup_config(r, config, dataset)
        r.set_tiles(args.tile if args.ranks > 1 else 0, args.rank, args.ranks, slab_layout=args.ranks > 1)
        r.create_targets()
        target = None
        slab = None
        if args.ranks > 1:
            slab = DeviceBuffer(r.slab_pixel_count(args.rank) * 16)
            target = slab.ptr.value
        times = {s: [] for s in settings}
        rays = {}
        first = True
        for _ in range(args.rounds):
            for s in settings:
                for name in touched:
                    os.environ.pop(name, None)
                fif = 3
                for pair in s.split(","):
                    if not pair:
                        continue
                    key, value = pair.split("=", 1)
                    if key == "fif":
                        fif = int(value)
                    else:
                        os.environ[key] = value
                r.frames_in_flight = fif
                r.create_pass()
                if first:
                    r.render_visibility()
                    first = False
                for _ in range(max(24, args.steps // 4)):
                    r.render(target)
                r.finish_frames(); r.sync()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    r.render(target)
                r.finish_frames(); r.sync()
                times[s].append((time.perf_counter() - t0) / args.steps * 1e3)
                rays[s] = r.last_ray_count()
        for s in settings:
            print(json.dumps({"config": args.config, "scene": args.scene, "ranks": args.ranks, "setting": s or "(defaults)", "ms_per_frame": [round(t, 4) for t in times[s]],
                              "min": round(min(times[s]), 4), "median": round(statistics.median(times[s]), 4), "rays": rays[s]}), flush=True)
        if slab is not None:
            r.sync()
            slab.free()
        r.close()


if __name__ == "__main__":
    main()
