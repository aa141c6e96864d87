#!/usr/bin/env python3
"""Where the gap between the wall-clock mean and the median of event-bracketed frame periods comes from (VERDICT round 5,
weak #9).  Renders one workload with frames in flight as bench.py does, every frame bracketed by events, and prints the
distribution of the periods between the ends of consecutive frames, of k-frame windows for several k, and the wall clock.

    python profiles/tools/frame_periods.py --config 3 --frames 240 --fif 3

If frames in flight finish in bursts of `fif`, the period between the ends of two frames that are k frames apart is
quantised to whole bursts: the median over k-frame windows is then biased unless k is a multiple of `fif`."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from vulkan_renderer_amd import renderer, synthetic


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="3")
    ap.add_argument("--frames", type=int, default=240)
    ap.add_argument("--fif", type=int, default=3)
    ap.add_argument("--mode", default="libm")
    args = ap.parse_args()
    config = args.config if args.config == "target" else int(args.config)
    with tempfile.TemporaryDirectory() as tmp:
        dataset = synthetic.write_dataset(tmp, grid=256, box_count=64, seed=1234, ltc_resolution=64, fresnel_count=51)
        r = renderer.Renderer(arithmetic=args.mode, frames_in_flight=args.fif, timing_stride=1)
        renderer.setup_config(r, config, dataset)
        r.create_targets()
        r.create_pass()
        r.render_visibility()
        for _ in range(300):
            r.render()
        r.finish_frames()
        r.sync()
        t0 = time.perf_counter()
        for _ in range(args.frames):
            r.render()
        r.finish_frames()
        r.sync()
        wall_ms = (time.perf_counter() - t0) * 1e3 / args.frames
        periods = np.array(r.frame_period_ms(args.frames - 1), np.float64)
        latency = np.array(r.dispatch_ms(args.frames), np.float64)
        r.close()
    ends = np.concatenate([[0.0], np.cumsum(periods)])
    out = {"config": str(config), "frames_in_flight": args.fif, "frames": args.frames, "wall_ms_per_frame": round(wall_ms, 4),
           "mean_period_ms": round(float(periods.mean()), 4), "median_period_ms": round(float(np.median(periods)), 4),
           "period_percentiles_ms": {str(q): round(float(np.percentile(periods, q)), 4) for q in (1, 10, 25, 50, 75, 90, 99)},
           "first_24_periods_ms": [round(float(v), 3) for v in periods[:24]],
           "latency_ms": {"mean": round(float(latency.mean()), 4), "median": round(float(np.median(latency)), 4)},
           "windows": {}}
    for k in (2, 3, 4, 6, 8, 9, 12, 16, 24):
        # non-overlapping windows of k frames, as a timing stride of k brackets them
        spans = (ends[k::k] - ends[:-k:k])[: (len(ends) - 1) // k] / k
        out["windows"][str(k)] = {"mean": round(float(spans.mean()), 4), "median": round(float(np.median(spans)), 4), "min": round(float(spans.min()), 4), "max": round(float(spans.max()), 4), "count": int(len(spans))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
