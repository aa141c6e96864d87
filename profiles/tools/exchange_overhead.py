#!/usr/bin/env python3
"""Where the exchange path loses time on a small slab (round 6): one rank's slab of an N-rank schedule rendered
  alone        render_shading_pass() into a buffer
  machinery    render_and_exchange_frame() with a collective that does nothing (events, streams, buffer sets only)
  copies       ... with the stand-in collective of predict_scaling.py (N - 1 slabs copied device to device)
  local        ... with the local transport of the library (create_local_slab_exchange, a group of ONE rank: no Python in the frame loop)
each with frames un-tiled on demand and with a scatter per frame.  Prints one JSON line."""
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
sys.path.insert(0, os.path.join(ROOT, "profiles", "tools"))
from helpers import DeviceBuffer
from predict_scaling import stand_in_collective, time_exchanged_frames, time_frames
from vulkan_renderer_amd import renderer, synthetic


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="3")
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--rank", type=int, default=3)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--fif", type=int, default=0)
    args = ap.parse_args()
    config = args.config if args.config == "target" else int(args.config)
    hip = C.CDLL("libamdhip64.so")
    out = {"config": str(config), "ranks": args.ranks, "rank": args.rank}
    with tempfile.TemporaryDirectory() as tmp:
        dataset = synthetic.write_dataset(tmp, grid=256, box_count=64, seed=1234, ltc_resolution=64, fresnel_count=51)
        depth = args.fif or renderer.frames_in_flight_for(args.ranks)
        r = renderer.Renderer(frames_in_flight=depth, timing_stride=63)
        renderer.setup_config(r, config, dataset)
        r.set_tiles(32, args.rank, args.ranks, slab_layout=True)
        r.create_targets(); r.create_pass(); r.render_visibility()
        pixels = r.slab_pixel_count(0)
        slab = DeviceBuffer(pixels * 16)
        peers = DeviceBuffer(pixels * 16 * args.ranks)
        out["frames_in_flight"] = depth
        for round_index in range(2):
            out.setdefault("alone", []).append(round(time_frames(r, slab.ptr.value, args.steps), 4))
            out.setdefault("alone_issue", []).append(round(time_frames.issue_ms, 4))
            # "c_noop": a C function that takes no argument and returns 0 (hipGetLastError) stands in for the collective: no Python in the loop
            noop = C.cast(hip.hipGetLastError, C.c_void_p).value
            r.exchange = renderer.capi.SlabExchange()
            assert r.lib.create_slab_exchange_with_gather(C.byref(r.exchange), C.byref(r.app), noop, None, 0) == 0
            r.assemble_on_demand(True)
            out.setdefault("c_noop", []).append(round(time_exchanged_frames(r, args.steps), 4))
            out.setdefault("c_noop_issue", []).append(round(time_exchanged_frames.issue_ms, 4))
            r.destroy_exchange()
            for name, gather in (("machinery", lambda *a: 0), ("copies", stand_in_collective(hip, peers.ptr.value, args.rank, args.ranks, 0)),
                                 ("copies16", stand_in_collective(hip, peers.ptr.value, args.rank, args.ranks, 16)), ("copies32", stand_in_collective(hip, peers.ptr.value, args.rank, args.ranks, 32))):
                for on_demand in (True, False):
                    r.create_exchange_with_gather(gather, "rgba32f")
                    r.assemble_on_demand(on_demand)
                    out.setdefault(name + ("" if on_demand else "+scatter"), []).append(round(time_exchanged_frames(r, args.steps), 4))
                    out.setdefault(name + ("" if on_demand else "+scatter") + "_issue", []).append(round(time_exchanged_frames.issue_ms, 4))
                    r.destroy_exchange()
        r.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
