#!/usr/bin/env python3
"""Benchmark of the shading pass (BASELINE.json: Msamples/s = pixels x spp / s).

  python bench.py --gpus 1 --steps K --warmup W
  python bench.py --gpus N ...                      (starts its N ranks itself: launch_ranks() below)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
  python bench.py --gpus N --dry-launch             (no GPU: the ranks only rendezvous over gloo)

A step is one pass of the shading kernels over one frame: write_constants -> upload ->
shade / trace / resolve over the rank's tiles [-> all-gather of the tile slabs -> scatter
into the frame].  Inputs (scene, BVH, LTC and noise tables, visibility buffer) are resident
in HBM before the timed region; data is synthetic (seeded generators,
vulkan_renderer_amd/synthetic.py).

Workload.  N = 1 runs BASELINE config 3 (1920x1080, 4 spp per technique, 4 polygonal lights,
diffuse + specular MIS with the clamped optimal heuristic, shadow rays), the heaviest 1080p
configuration; --config 1 | 2 | 4 | target select the others (target = north_star's
"1920x1080, 4 spp, 1 light").  N > 1 scales STRONGLY: the same fixed frame is cut into tiles,
tile t goes to rank t mod N, every rank shades its tiles into a dense slab, one RCCL
all-gather (ncclAllGather called from C behind the C-ABI, include/vkr_slab_exchange.h) per
frame delivers all slabs to all ranks and a scatter kernel reassembles the frame - all inside
the timed region, overlapped with the shading of the next frame.  The value at N = 1 is the
plain single-GPU pass (no exchange), so the per-N values of one workload are comparable.
Every run also measures BASELINE config 4 (3840x2160, 8 spp, 8 lights: the configuration
BASELINE.json tiles over 8 GPUs) as a second workload and attaches it as "secondary" to the
JSON line (--no-secondary to skip).  --scaling weak / --exchange none keep the round-1 mode
(frame height x N, slabs stay where they are) as an option.

PyTorch is plumbing here: device selection, the stream, the process group used for the
rendezvous token, barriers and the max over ranks of the timing.
"""
import argparse
import ctypes
import json
import os

# The frame streams of the shading pass and the exchange stream need hardware queues of their own
# next to torch's and RCCL's streams; the HIP runtime multiplexes all streams onto 4 queues unless
# told otherwise.  Must be set before the runtime initialises (i.e. before torch is imported).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The benchmark lives in vulkan_renderer_amd/bench/ (launching, timing, parity, roofline and the shaping of the line are modules
# of their own since round 6); the names below are what this file has always offered to tests and tools.
from vulkan_renderer_amd.bench.common import (FP32_VECTOR_PEAK_TFLOPS, HBM_PEAK_GBPS, algorithmic_bytes_per_pixel, available_cpus, cpu_model,  # noqa: E402,F401
                                              kernel_source_hash, libm_identity, parse_config, pmc_entry_for)
from vulkan_renderer_amd.bench.launch import Job, dry_launch, free_port, launch_ranks  # noqa: E402,F401
from vulkan_renderer_amd.bench.line import LINE_LIMIT, short_line, write_details  # noqa: E402,F401
from vulkan_renderer_amd.bench.roofline import build_roofline, live_traffic  # noqa: E402,F401
from vulkan_renderer_amd.bench.workload import mode_companion, protocol_window, run_workload, timing_matrix_cells  # noqa: E402,F401


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed frames (default 2000 / 2000 / 500 / 100 for configs 1-4)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed frames right before the timed ones (default: a tenth of the steps); --prewarm-frames come before them")
    ap.add_argument("--config", type=parse_config, default=3, choices=[1, 2, 3, 4, "target"], help="BASELINE.json configuration (default 3, the heaviest 1080p one); target = 1920x1080, 4 spp, 1 light")
    ap.add_argument("--scene", default="bench", choices=["bench", "large"], help="bench: the scene of SURVEY.md 8(d) (131 840 triangles); large: 2.6 M triangles with stacked occluders, thin triangles, deep occlusion, eight materials")
    ap.add_argument("--no-secondary", action="store_true", help="do not also measure BASELINE config 4 (3840x2160, 8 spp, 8 lights)")
    ap.add_argument("--mode", default="libm", choices=["libm", "exact", "fast"],
                    help="libm: IEEE arithmetic with glibc's transcendentals, bit-identical to the CPU oracle in the mode that is pinned against the reference shader (default); "
                         "exact: the libm mode with a polynomial arctangent, bit-identical to the oracle's math mode 1 (arithmetic_mode_polynomial of the C-ABI); fast: approximate reciprocals + contraction")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"], help="N > 1: strong = the fixed frame is cut into tiles (default); weak = the frame height grows with N")
    ap.add_argument("--exchange", choices=("rgba32f", "rgb8", "none"), default="rgba32f",
                    help="N > 1: all-gather of the tile slabs per frame as float radiance (default) or as packed RGB8 of the encoded output, then the scatter into the frame on every rank; none leaves every rank's slab in its HBM")
    ap.add_argument("--assemble", choices=("on-demand", "every-frame"), default="on-demand",
                    help="N > 1: on-demand (default) = a frame stays the gathered slabs (tile-major) and is un-tiled when it is read - once per fence of this run; every-frame = a scatter kernel behind every all-gather (SURVEY.md 8e names both)")
    ap.add_argument("--tile-size", type=int, default=32)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--bvh", default="sah_device", choices=["sah_device", "lbvh_device", "sah_host"], help="who builds the BVH (default: binned SAH by HIP kernels)")
    ap.add_argument("--binary-traversal", action="store_true", help="walk the binary tree in the wavefront kernel instead of the four-wide one")
    ap.add_argument("--no-rays", action="store_true", help="disable shadow rays (TRACE_SHADOW_RAYS=0) for experiments")
    ap.add_argument("--inline-rays", action="store_true", help="trace shadow rays inside the shading kernel instead of the wavefront path")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-modes", dest="no_other_modes", action="store_true", help="do not also measure the workload in the exact (polynomial arctangent) and fast arithmetic modes (reported as \"other_modes\" next to the headline)")
    ap.add_argument("--no-fast-mode", action="store_true", help="other_modes without the fast arithmetic mode (approximate reciprocals and roots, contraction, polynomial transcendentals)")
    ap.add_argument("--no-large-scene", action="store_true", help="do not also run config 3 on the large scene (2.6 M triangles) as an extra workload")
    ap.add_argument("--no-extra", action="store_true", help="do not also run the north_star target shape (1920x1080, 4 spp, 1 light) and BASELINE config 2 as short extra workloads")
    ap.add_argument("--traversal-stats", action="store_true", help="attach BVH traversal work counters to the secondary workload too (diagnostics)")
    ap.add_argument("--frames-in-flight", type=int, default=None, choices=(1, 2, 3, 4, 5, 6, 7, 8),
                    help="n >= 2: n consecutive frames overlap on the device's frame streams, like the frames of the reference's frame queue, which is as deep as "
                         "its swapchain (main.c:1498: typically 3).  Default: 3, and 4 from eight ranks on (renderer.frames_in_flight_for)")
    ap.add_argument("--timing-stride", type=int, default=8, help="bracket every n-th frame of the timed region with HIP events (rounded up to a multiple of the frames in flight: protocol_window())")
    ap.add_argument("--prewarm-frames", type=int, default=200, help="untimed frames before --warmup that bring clocks and the frame pipeline to their steady state")
    ap.add_argument("--prewarm-seconds", type=float, default=1.5, help="... but no longer than this (after the first eight)")
    ap.add_argument("--ltc-resolution", type=int, default=64, help="roughness / inclination resolution R of the generated LTC tables (SURVEY.md 8d: 64)")
    ap.add_argument("--no-host-frames", action="store_true", help="do not also measure the rate with every frame read back to the host (with_readback)")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not measure roofline.traffic in this run (two short rocprofv3 --pmc passes around a child run of the headline workload); the committed passes of profiles/pmc_traffic.json are used instead")
    ap.add_argument("--force-distributed", action="store_true", help="run the N > 1 code path (slab layout, exchange) even with one rank")
    ap.add_argument("--details", default=None, help="where the full record of the run goes (default gpurun_out/bench_details.json); the printed line is the short one")
    ap.add_argument("--dry-line", default=None, metavar="RECORD", help="no GPU work: print the short line for the full record of an earlier run (a details file, or a file whose last line is a record)")
    ap.add_argument("--dry-launch", action="store_true", help="no GPU work: the ranks join a gloo process group, exchange a token, print one JSON line and leave (checks the launch path of --gpus N)")
    args = ap.parse_args()

    if args.dry_line:
        text = open(args.dry_line).read()
        try:
            record = json.loads(text)
        except ValueError:
            record = json.loads([line for line in text.splitlines() if line.startswith("{")][-1])
        print(json.dumps(short_line(record, os.path.relpath(os.path.abspath(args.dry_line), ROOT)), separators=(",", ":")), flush=True)
        return
    # `python bench.py --gpus N` by itself: become the launcher of N ranks (one per GPU)
    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and "VKR_BENCH_SELF_LAUNCHED" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.dry_launch:
        dry_launch(args)
        return

    job = Job(args)
    result = run_workload(job, args.config, "primary")
    customised = bool(args.width or args.height or args.spp)
    if not args.no_extra and args.config == 3 and not customised and job.world == 1:
        # the other 1920x1080 shapes the contract names, on the same clock: north_star's target and BASELINE config 2
        keep = ("value", "unit", "steps", "warmup", "ms_per_step", "median_frame_period_ms", "value_from_median", "latency_ms", "value_single_frame", "shaded_fraction",
                "config", "shadow_rays_per_frame", "Mrays_per_s", "light_shafts", "roofline", "parity", "traversal", "setup", "cpu_only")
        result["extra_workloads"] = {}
        for extra, scene in (("target", args.scene), (2, args.scene), (1, args.scene)) + ((("3", "large"),) if args.scene == "bench" and not args.no_large_scene else ()):
            line = run_workload(job, int(extra) if extra != "target" else extra, "extra", scene)
            result["extra_workloads"]["config_%s%s" % (extra, "" if scene == args.scene else "_large_scene")] = {k: line[k] for k in keep if k in line}
        target_value = result["extra_workloads"]["config_target"]["value"]
        result["north_star_target"] = {"shape": "1920x1080, 4 spp, 1 polygonal light", "target_Msamples_per_s": 1000.0, "value": target_value, "met": bool(target_value >= 1000.0),
                                       "parity": result["extra_workloads"]["config_target"].get("parity")}
    if not args.no_extra and args.config == 3 and not customised and job.world == 1 and job.rank == 0:
        try:
            result["timing_matrix_cells"] = timing_matrix_cells(job)
        except Exception as error:  # (diagnostics for the details file: never a reason to lose the line)
            result["timing_matrix_cells"] = {"error": repr(error)}
    if (job.world > 1 or args.force_distributed) and args.exchange == "rgba32f" and not args.no_extra:
        # The same tiled frame with the ENCODED output as the exchanged format (packed RGB8 - the format the reference's pass
        # writes, shading_pass.frag.glsl:871-892 - 3 instead of 16 bytes per pixel over the links), on the same clock: if the
        # float exchange is bound by the links at this N, this line shows what the shading side allows.
        encoded = run_workload(job, args.config, "extra", exchange_format="rgb8")
        result["exchange_rgb8"] = {k: encoded[k] for k in ("value", "unit", "steps", "ms_per_step", "median_frame_period_ms", "scaling_parity", "stages") if k in encoded}
    if not args.no_secondary and args.config != 4 and not customised:
        second = run_workload(job, 4, "secondary")
        keep = ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "config", "shadow_rays_per_frame", "Mrays_per_s", "light_shafts", "stages", "scaling_parity", "setup", "roofline", "traversal")
        result["secondary"] = {k: second[k] for k in keep if k in second}
    # the library reports like the reference does (printf): every rank flushes C stdio before rank 0
    # prints, so that the JSON line is the last line of the job's output
    ctypes.CDLL(None).fflush(None)
    job.barrier()
    if job.rank == 0:
        details_path = write_details(result, args.details)
        print(json.dumps(short_line(result, details_path), separators=(",", ":")), flush=True)
    job.close()


if __name__ == "__main__":
    main()
