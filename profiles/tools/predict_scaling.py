#!/usr/bin/env python3
"""Strong scaling of the tiled frame, predicted from ONE GPU (no N > 1 hardware was available to the
builder: gpurun boxes have one GPU).

For BASELINE configs 3 and 4 and the north_star target shape, N in {2, 4, 8} ranks and tile size 32 this renders every rank's
slab - the real tile schedule (tile t -> rank t mod N), the whole pass (shafts, shade, trace, resolve), frames in flight
as in bench.py - three ways:
  alone      render_shading_pass() into a slab, nothing else (what rounds 3 - 5 reported)
  exchange   render_and_exchange_frame() with the exchange machinery of include/vkr_slab_exchange.h - buffer sets, events
             between frame streams and exchange stream, in-place gather, frames un-tiled on demand - and a STAND-IN
             collective that moves what a ring all-gather moves through this GPU's memory: N - 1 slabs copied into the
             gathered buffer (device-to-device, from a buffer that plays the peers), on the exchange stream
  scatter    the same with the scatter kernel behind every gather (--assemble every-frame of bench.py)
and combines the slowest rank's time with a MODEL of the collective on xGMI (7 links x 153 GB/s per GPU), which overlaps the
next frame - a rank's frame period is max(its shading side, the collective):
  direct     every peer's slab arrives over that peer's own link:              slab / 153 GB/s
  rings      RCCL spreads its rings over the links; half the aggregate rate:   (N - 1) slabs / (0.5 x 7 x 153 GB/s)
  one ring   all traffic over one link (the pessimistic end):                  (N - 1) slabs / 153 GB/s
Predicted speed-up = t(whole frame on one GPU) / max(max_r t(rank r), collective).
What the prediction still cannot see: RCCL's own kernels (they hold some CUs while they wait for the links), host launch
overheads of 8 processes, clock differences between GPUs.

    python profiles/tools/predict_scaling.py [--configs 3 target 4] [--out gpurun_out/r10/predicted_scaling]
"""
import argparse
import ctypes as C
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
    # (at least two rounds of the frame contexts: a context allocates its buffers when it first renders)
    for _ in range(max(9, steps // 4)):
        r.render(target)
    r.finish_frames(); r.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        r.render(target)
    time_frames.issue_ms = (time.perf_counter() - t0) / steps * 1e3
    r.finish_frames(); r.sync()
    return (time.perf_counter() - t0) / steps * 1e3


def time_exchanged_frames(r, steps):
    for _ in range(max(9, steps // 4)):
        r.render_and_exchange(None)
    r.finish_exchange(); r.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        r.render_and_exchange(None)
    # (host time to queue a frame: a bound when the host cannot keep up with slabs of 0.1 ms)
    time_exchanged_frames.issue_ms = (time.perf_counter() - t0) / steps * 1e3
    r.finish_exchange(); r.sync()
    return (time.perf_counter() - t0) / steps * 1e3


def stand_in_collective(hip, peers, rank, ranks, workgroups=0):
    """slab_gather_function_t: the N - 1 slabs of the peers land in their slots of `gathered` (two device-to-device copies
    around the rank's own slot, which the frame was shaded into).  workgroups > 0: copied by that many workgroups
    (copy_with_workgroups of the C-ABI) - a few compute units busy for a while, like the channels of a collective; 0: by
    hipMemcpyAsync, whose blit kernel takes the whole GPU for a moment (what round 6 first measured: 0.03 - 0.05 ms per frame of
    a rank's shading side at N = 8, profiles/r10k)."""
    from vulkan_renderer_amd import capi
    lib = capi.load()

    def copy(destination, source, nbytes, stream):
        if workgroups:
            return lib.copy_with_workgroups(destination, source, nbytes, workgroups, stream)
        return hip.hipMemcpyAsync(C.c_void_p(destination), C.c_void_p(source), C.c_size_t(nbytes), 3, C.c_void_p(stream))

    def gather(my_rank, buffer_set, send, gathered, send_bytes, stream):
        failed = 0
        if rank > 0:
            failed |= copy(gathered, peers, rank * send_bytes, stream)
        if rank + 1 < ranks:
            offset = (rank + 1) * send_bytes
            failed |= copy(gathered + offset, peers + offset, (ranks - 1 - rank) * send_bytes, stream)
        return int(failed != 0)
    return gather


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", nargs="+", default=["3", "target", "4"])
    ap.add_argument("--ranks", nargs="+", type=int, default=[2, 4, 8])
    ap.add_argument("--tiles", nargs="+", type=int, default=[32])
    ap.add_argument("--mode", default="libm")
    ap.add_argument("--frames-in-flight", type=int, default=0, help="0: what bench.py uses for that many ranks (renderer.frames_in_flight_for)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "predicted_scaling"))
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    hip = C.CDLL("libamdhip64.so")
    results = []
    lines = ["# Predicted strong scaling of the tiled frame (one GPU, every rank's slab rendered by itself)", "",
             "`profiles/tools/predict_scaling.py` on one MI355X, arithmetic mode %s, frames in flight as in bench.py (three; four from eight ranks on), RGBA32F slabs.  **No N > 1 run exists**: this is a prediction from measured per-rank work and a model of the links." % args.mode,
             "", "Per-rank ms: `alone` = the pass into a slab; `exchange` = through render_and_exchange_frame() with a stand-in collective that copies N - 1 slabs into the gathered buffer (frames un-tiled on demand); `scatter` = the same with the scatter kernel behind every gather.",
             "Collective on xGMI (modelled, overlaps the next frame): direct = slab / 153 GB/s; rings = (N - 1) slabs / 535 GB/s; one ring = (N - 1) slabs / 153 GB/s.  Speed-up = whole frame / max(slowest rank, collective).", ""]
    with tempfile.TemporaryDirectory() as tmp:
        dataset = synthetic.write_dataset(tmp, grid=256, box_count=64, seed=1234, ltc_resolution=64, fresnel_count=51)
        for config in [c if c == "target" else int(c) for c in args.configs]:
            r = renderer.Renderer(arithmetic=args.mode, frames_in_flight=3, timing_stride=63)
            settings = renderer.setup_config(r, config, dataset)
            width, height = settings["width"], settings["height"]
            steps = 6 if config == 4 else 40
            r.set_tiles(0, 0, 1, slab_layout=False)
            r.create_targets(); r.create_pass(); r.render_visibility()
            whole_ms = time_frames(r, None, steps if config == 4 else 400)
            lines += ["## BASELINE config %s (%dx%d): whole frame on one GPU %.3f ms" % (config, width, height, whole_ms), "",
                      "| ranks | tile | slowest rank ms: alone / exchange / scatter | balance (exchange) | collective ms: direct / rings / one ring | speed-up, shading side only (alone) | **speed-up with the exchange: direct / rings / one ring** | ... with a scatter per frame (rings) | exchanging the encoded output (rgb8): slowest rank ms, speed-up (one ring) |", "|---|---|---|---|---|---|---|---|---|"]
            for ranks in args.ranks:
                depth = args.frames_in_flight or renderer.frames_in_flight_for(ranks)
                if depth != r.frames_in_flight:
                    r.frames_in_flight = depth
                    r.create_pass()
                for tile in args.tiles:
                    alone, exchanged, scattered, encoded = [], [], [], []
                    slab_pixels = 0
                    rank_steps = max(4, steps if config == 4 else steps * 4)
                    for rank in range(ranks):
                        r.set_tiles(tile, rank, ranks, slab_layout=True)
                        slab_pixels = r.slab_pixel_count(0)
                        slab = DeviceBuffer(slab_pixels * 16)
                        alone.append(time_frames(r, slab.ptr.value, rank_steps))
                        r.sync()
                        slab.free()
                        peers = DeviceBuffer(slab_pixels * 16 * ranks)
                        for on_demand, slab_format, out in ((True, "rgba32f", exchanged), (False, "rgba32f", scattered), (True, "rgb8", encoded)):
                            r.create_exchange_with_gather(stand_in_collective(hip, peers.ptr.value, rank, ranks), slab_format)
                            r.assemble_on_demand(on_demand)
                            out.append(time_exchanged_frames(r, rank_steps))
                            r.destroy_exchange()
                        peers.free()
                    slab_mb = slab_pixels * 16 / 1e6
                    collective = {"direct": slab_mb / XGMI_LINK_GBPS, "rings": (ranks - 1) * slab_mb / (0.5 * 7 * XGMI_LINK_GBPS), "one_ring": (ranks - 1) * slab_mb / XGMI_LINK_GBPS}
                    with_exchange = {k: whole_ms / max(max(exchanged), v) for k, v in collective.items()}
                    with_scatter = {k: whole_ms / max(max(scattered), v) for k, v in collective.items()}
                    # the encoded output (packed RGB8, 3 of 16 bytes per pixel) as the exchanged format: --exchange rgb8 of bench.py
                    with_rgb8 = {k: whole_ms / max(max(encoded), v * 3.0 / 16.0) for k, v in collective.items()}
                    entry = {"config": config, "ranks": ranks, "tile": tile, "whole_frame_ms": round(whole_ms, 4), "per_rank_ms": [round(v, 4) for v in alone],
                             "per_rank_ms_exchange": [round(v, 4) for v in exchanged], "per_rank_ms_scatter": [round(v, 4) for v in scattered], "per_rank_ms_exchange_rgb8": [round(v, 4) for v in encoded],
                             "balance": round(sum(exchanged) / (ranks * max(exchanged)), 4), "predicted_speedup_shading_only": round(whole_ms / max(alone), 3),
                             "predicted_speedup": {k: round(v, 3) for k, v in with_exchange.items()}, "predicted_speedup_scatter_every_frame": {k: round(v, 3) for k, v in with_scatter.items()}, "predicted_speedup_rgb8": {k: round(v, 3) for k, v in with_rgb8.items()},
                             "collective_ms": {k: round(v, 4) for k, v in collective.items()},
                             "frames_in_flight": depth, "slab_mb": round(slab_mb, 2), "bands_per_frame": int(r.app.shading_pass.last_band_count)}
                    results.append(entry)
                    print(json.dumps(entry), flush=True)
                    lines.append("| %d | %d | %.3f / %.3f / %.3f | %.3f | %.3f / %.3f / %.3f | %.2f | **%.2f / %.2f / %.2f** | %.2f | %.3f, %.2f |" % (
                        ranks, tile, max(alone), max(exchanged), max(scattered), entry["balance"], collective["direct"], collective["rings"], collective["one_ring"],
                        entry["predicted_speedup_shading_only"], with_exchange["direct"], with_exchange["rings"], with_exchange["one_ring"], with_scatter["rings"], max(encoded), with_rgb8["one_ring"]))
            lines.append("")
            r.close()
    open(args.out + ".md", "w").write("\n".join(lines) + "\n")
    json.dump(results, open(args.out + ".json", "w"), indent=1)


if __name__ == "__main__":
    main()
