#!/usr/bin/env python3
"""One rank's slab through one path - alone / machinery (collective that does nothing) / copies (stand-in collective) - as a
command that rocprofv3 can wrap (profiles/tools/exchange_overhead.py runs them all in turn).
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/x -o trace -- python profiles/tools/exchange_probe.py --path machinery"""
import argparse
import ctypes as C
import json
import os
import sys
import tempfile

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
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--path", default="alone", choices=["alone", "machinery", "copies"])
    ap.add_argument("--scatter", action="store_true")
    args = ap.parse_args()
    config = args.config if args.config == "target" else int(args.config)
    hip = C.CDLL("libamdhip64.so")
    with tempfile.TemporaryDirectory() as tmp:
        dataset = synthetic.write_dataset(tmp, grid=256, box_count=64, seed=1234, ltc_resolution=64, fresnel_count=51)
        r = renderer.Renderer(frames_in_flight=renderer.frames_in_flight_for(args.ranks), timing_stride=63)
        renderer.setup_config(r, config, dataset)
        r.set_tiles(32, args.rank, args.ranks, slab_layout=True)
        r.create_targets(); r.create_pass(); r.render_visibility()
        pixels = r.slab_pixel_count(0)
        slab = DeviceBuffer(pixels * 16)
        peers = DeviceBuffer(pixels * 16 * args.ranks)
        if args.path == "alone":
            ms = time_frames(r, slab.ptr.value, args.steps)
        else:
            r.create_exchange_with_gather((lambda *a: 0) if args.path == "machinery" else stand_in_collective(hip, peers.ptr.value, args.rank, args.ranks), "rgba32f")
            r.assemble_on_demand(not args.scatter)
            ms = time_exchanged_frames(r, args.steps)
            r.destroy_exchange()
        r.close()
    print(json.dumps({"path": args.path, "ms": round(ms, 4)}))


if __name__ == "__main__":
    main()
