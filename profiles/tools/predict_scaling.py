#!/usr/bin/env python3
"""Strong scaling of the tiled frame, predicted from ONE GPU (no N > 1 hardware was available to the
builder: gpurun boxes have one GPU).

For BASELINE configs 3 and 4, N in {2, 4, 8} ranks and tile sizes 16 / 32 / 64 this renders every rank's
slab ALONE - the real tile schedule (tile t -> rank t mod N), the whole pass (shade, trace, resolve),
frames in flight as in bench.py - and reports per-rank milliseconds, their balance (sum / max / N) and
the predicted speed-up t(whole frame on one GPU) / max_r t(rank r's slab), next to the time the
all-gather of the slabs needs on xGMI (7 links x 153 GB/s per GPU: one slab per link for a direct
all-gather, N - 1 slabs over one link for a ring), which overlaps the next frame.
What the prediction leaves out: the exchange itself (it runs on its own stream; its kernels take a few
CUs for ~0.1-0.8 ms per frame), host launch overheads of 8 processes, and clock differences between GPUs.

    python profiles/tools/predict_scaling.py [--configs 3 4] [--out gpurun_out/r03/predicted_scaling]
"""
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
import numpy as np

from helpers import DeviceBuffer
from vulkan_renderer_amd import renderer, synthetic

XGMI_LINK_GBPS = 153.0


def time_frames(r, target, steps):
    for _ in range(max(3, steps // 4)):
        r.render(target)
    r.finish_frames(); r.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        r.render(target)
    r.finish_frames(); r.sync()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", nargs="+", default=["3", "4"])
    ap.add_argument("--ranks", nargs="+", type=int, default=[2, 4, 8])
    ap.add_argument("--tiles", nargs="+", type=int, default=[16, 32, 64])
    ap.add_argument("--mode", default="libm")
    ap.add_argument("--frames-in-flight", type=int, default=0, help="0: what bench.py uses for that many ranks (renderer.frames_in_flight_for)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "predicted_scaling"))
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    results = []
    lines = ["# Predicted strong scaling of the tiled frame (one GPU, every rank's slab rendered alone)", "",
             "`profiles/tools/predict_scaling.py` on one MI355X, arithmetic mode %s, frames in flight as in bench.py (three; four from eight ranks on).  **No N > 1 run exists**: this is a prediction from measured per-rank work." % args.mode, ""]
    with tempfile.TemporaryDirectory() as tmp:
        dataset = synthetic.write_dataset(tmp, grid=256, box_count=64, seed=1234, ltc_resolution=64, fresnel_count=51)
        for config in [c if c == "target" else int(c) for c in args.configs]:
            r = renderer.Renderer(arithmetic=args.mode, frames_in_flight=3, timing_stride=64)
            settings = renderer.setup_config(r, config, dataset)
            width, height = settings["width"], settings["height"]
            steps = 6 if config == 4 else 40
            r.set_tiles(16, 0, 1, slab_layout=False)
            r.create_targets(); r.create_pass(); r.render_visibility()
            whole_ms = time_frames(r, None, steps if config == 4 else 400)
            lines += ["## BASELINE config %s (%dx%d): whole frame on one GPU %.3f ms" % (config, width, height, whole_ms), "",
                      "| ranks | tile | per-rank ms (rank 0 ... N-1) | balance = sum / (N max) | predicted speed-up = whole / max | slab MB | all-gather ms direct / ring |", "|---|---|---|---|---|---|---|"]
            for ranks in args.ranks:
                depth = args.frames_in_flight or renderer.frames_in_flight_for(ranks)
                if depth != r.frames_in_flight:
                    r.frames_in_flight = depth
                    r.create_pass()
                for tile in args.tiles:
                    per_rank = []
                    slab_pixels = 0
                    for rank in range(ranks):
                        r.set_tiles(tile, rank, ranks, slab_layout=True)
                        slab_pixels = r.slab_pixel_count(0)
                        slab = DeviceBuffer(slab_pixels * 16)
                        per_rank.append(time_frames(r, slab.ptr.value, max(4, steps if config == 4 else steps * 4)))
                        r.sync()
                        slab.free()
                    slab_mb = slab_pixels * 16 / 1e6
                    direct_ms, ring_ms = slab_mb / XGMI_LINK_GBPS, (ranks - 1) * slab_mb / XGMI_LINK_GBPS
                    entry = {"config": config, "ranks": ranks, "tile": tile, "whole_frame_ms": round(whole_ms, 4), "per_rank_ms": [round(v, 4) for v in per_rank],
                             "balance": round(sum(per_rank) / (ranks * max(per_rank)), 4), "predicted_speedup": round(whole_ms / max(per_rank), 3),
                             "sum_over_max": round(sum(per_rank) / max(per_rank), 3), "frames_in_flight": depth, "slab_mb": round(slab_mb, 2), "all_gather_ms_direct": round(direct_ms, 4), "all_gather_ms_ring": round(ring_ms, 4),
                             "bands_per_frame": int(r.app.shading_pass.last_band_count)}
                    results.append(entry)
                    print(json.dumps(entry), flush=True)
                    lines.append("| %d | %d | %s | %.3f | **%.2f** | %.1f | %.3f / %.3f |" % (ranks, tile, " ".join("%.3f" % v for v in per_rank), entry["balance"], entry["predicted_speedup"], slab_mb, direct_ms, ring_ms))
            lines.append("")
            r.close()
    open(args.out + ".md", "w").write("\n".join(lines) + "\n")
    json.dump(results, open(args.out + ".json", "w"), indent=1)


if __name__ == "__main__":
    main()
