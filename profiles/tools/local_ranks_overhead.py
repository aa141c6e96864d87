#!/usr/bin/env python3
"""What the N-rank pipeline costs when nothing is gained: N ranks of ONE process share ONE GPU.

Every rank is a thread with its own application_t, renders its tiles into a slab, the slabs are
exchanged with device-to-device copies (create_local_slab_exchange: the collective that stands in for
ncclAllGather where RCCL refuses several ranks on a device) and every rank scatters the whole frame.
The GPU does the work of one frame plus N gathers and N scatters per frame, so frames per second fall
short of the single-rank figure by exactly the overhead of tiling, exchange and N host threads - a
number a single-GPU box CAN measure.  (It is not a scaling measurement.)

    python profiles/tools/local_ranks_overhead.py [--config 3] [--ranks 1 2 4 8] [--tile 32]
"""
import argparse
import json
import os
import sys
import tempfile
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vulkan_renderer_amd import capi, renderer, synthetic


def run(config, dataset, ranks, tile, steps, mode):
    lib = capi.load()
    group = lib.create_local_slab_group(ranks)
    renderers, elapsed, errors = [None] * ranks, [0.0] * ranks, []
    ready, go = threading.Barrier(ranks + 1), threading.Barrier(ranks + 1)

    def rank_thread(rank):
        try:
            r = renderer.Renderer(arithmetic=mode, frames_in_flight=3, timing_stride=64)
            renderer.setup_config(r, config, dataset, acceleration_structure="sah_device")
            r.set_tiles(tile, rank, ranks, slab_layout=True)
            r.create_targets(); r.create_pass(); r.render_visibility()
            r.create_local_exchange(group, "rgba32f")
            renderers[rank] = r
            for _ in range(max(8, steps // 5)):
                r.render_and_exchange(None)
            r.finish_exchange(); r.sync()
            ready.wait(); go.wait()
            t0 = time.perf_counter()
            for _ in range(steps):
                r.render_and_exchange(None)
            r.finish_exchange(); r.sync()
            elapsed[rank] = time.perf_counter() - t0
        except Exception as error:
            errors.append(repr(error))
            raise

    threads = [threading.Thread(target=rank_thread, args=(k,)) for k in range(ranks)]
    for t in threads:
        t.start()
    ready.wait(); go.wait()
    for t in threads:
        t.join()
    assert not errors, errors
    for r in renderers:
        r.destroy_exchange(); r.close()
    lib.destroy_local_slab_group(group)
    return max(elapsed) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--ranks", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--tile", type=int, default=32)
    ap.add_argument("--mode", default="libm")
    args = ap.parse_args()
    steps = 200 if args.config != 4 else 12
    with tempfile.TemporaryDirectory() as tmp:
        dataset = synthetic.write_dataset(tmp, grid=256, box_count=64, seed=1234, ltc_resolution=64, fresnel_count=51)
        r = renderer.Renderer(arithmetic=args.mode, frames_in_flight=3, timing_stride=64)
        settings = renderer.setup_config(r, args.config, dataset, acceleration_structure="sah_device")
        r.create_targets(); r.create_pass(); r.render_visibility()
        for _ in range(20):
            r.render()
        r.finish_frames(); r.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            r.render()
        r.finish_frames(); r.sync()
        plain = (time.perf_counter() - t0) / steps * 1e3
        r.close()
        print(json.dumps({"config": args.config, "pipeline": "plain single-GPU pass, no tiles, no exchange", "ms_per_frame": round(plain, 4)}), flush=True)
        for ranks in args.ranks:
            ms = run(args.config, dataset, ranks, args.tile, steps, args.mode)
            print(json.dumps({"config": args.config, "ranks_sharing_one_gpu": ranks, "tile": args.tile, "ms_per_frame": round(ms, 4), "overhead_vs_plain": round(ms / plain - 1.0, 4),
                              "frame": "%dx%d" % (settings["width"], settings["height"])}), flush=True)


if __name__ == "__main__":
    main()
