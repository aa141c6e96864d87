#!/usr/bin/env python3
"""Benchmark of the shading pass (BASELINE.json: Msamples/s = pixels x spp / s).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one pass of the shading kernels over one frame: write_constants ->
upload -> shade / trace / resolve over the rank's tiles.  Inputs (scene, LBVH,
LTC and noise tables, visibility buffer) are resident in HBM before the timed
region; data is synthetic (seeded generators, vulkan_renderer_amd/synthetic.py).

N = 1 runs BASELINE config 2 (1920x1080, 1 spp, one pentagon light, GGX MIS with
projected-solid-angle sampling, LBVH shadow rays) unless --config says otherwise.
For N > 1 the scaling is weak: the frame grows to 1920 x (1080 N) pixels, tiles
are dealt round-robin to the ranks, so every GPU shades one 1080p frame worth of
pixels per step into its slab of the frame, which stays resident in its HBM like the
frame does at N = 1: pixels are independent, the pass has no exchange step, hence no
data-path collective (--exchange none, the default).  --exchange rgb8 | rgba8 | rgba32f
adds what a consumer of whole frames on every GPU would need: an RCCL all-gather of the
(encoded) slabs per frame, overlapped with the next frame, and the scatter into a frame.

PyTorch is plumbing here: device selection, the stream, torch.distributed.
"""
import argparse
import ctypes
import json
import math
import os

# The two frame streams of the shading pass need hardware queues of their own next to torch's and
# RCCL's streams; the HIP runtime multiplexes all streams onto 4 queues unless told otherwise.
# Must be set before the runtime initialises (i.e. before torch is imported).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def algorithmic_bytes_per_pixel(light_count, sample_count, techniques):
    """visibility id + 3 vertices (positions, normals/uv) + material id
    + 4 LTC texels + noise texels + RGBA32F out."""
    noise_fetches = math.ceil(light_count * sample_count * techniques / 2)
    return 4 + 49 + 48 + 8 * noise_fetches + 16


def available_cpus():
    """Host threads this process may really use: affinity mask and the cgroup's CPU quota (a
    container sees all cores of the machine in os.cpu_count() but is throttled to its quota:
    256 threads on a quota of a few cores ran in bursts of 100 ms periods)."""
    count = os.cpu_count() or 1
    try:
        count = min(count, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        fields = open("/sys/fs/cgroup/cpu.max").read().split()  # cgroup v2: "<quota|max> <period>"
        if fields and fields[0] != "max":
            quota = float(fields[0]) / float(fields[1])
    except (OSError, ValueError, IndexError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota:
        count = max(1, min(count, int(math.ceil(quota))))
    return count


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed frames (default 2000 / 2000 / 500 / 100 for configs 1-4); a few hundred are needed before the clocks and the frame pipeline are in steady state (config 2: 100 frames are 14 ms)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed frames first (default: a tenth of the steps)")
    ap.add_argument("--config", type=int, default=2, choices=[1, 2, 3, 4])
    ap.add_argument("--mode", default="exact", choices=["fast", "exact"],
                    help="exact: IEEE arithmetic, bit-identical to the CPU oracle (default; it is as fast); fast: approximate reciprocals + contraction")
    ap.add_argument("--tile-size", type=int, default=32)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--no-rays", action="store_true", help="disable shadow rays (TRACE_SHADOW_RAYS=0) for experiments")
    ap.add_argument("--inline-rays", action="store_true", help="trace shadow rays inside the shading kernel instead of the wavefront path")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--traversal-stats", action="store_true", help="attach BVH traversal work counters (diagnostics)")
    ap.add_argument("--frames-in-flight", type=int, default=2, choices=(1, 2, 3, 4), help="n >= 2: n consecutive frames overlap on the device's frame streams (like the reference's frame queue)")
    ap.add_argument("--timing-stride", type=int, default=8, help="bracket every n-th frame with HIP events for the kernel time (roofline)")
    ap.add_argument("--exchange", choices=("none", "rgb8", "rgba8", "rgba32f"), default="none", help="N > 1: none (default) leaves every rank's slab of the frame in its HBM; the others all-gather the slabs per frame and reassemble the frame on every rank: the encoded frame as packed RGB8 (3 bytes per pixel), as RGBA8, or float radiance")
    ap.add_argument("--force-distributed", action="store_true", help="run the N > 1 code path even with one rank")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = {1: 2000, 2: 2000, 3: 500, 4: 100}[args.config]
    if args.warmup is None:
        args.warmup = max(args.steps // 10, 1)
    if args.steps < 4 * args.timing_stride:
        args.timing_stride = 1  # short runs: time every frame

    import torch
    import torch.distributed as dist

    from vulkan_renderer_amd import renderer, synthetic

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1 or args.force_distributed
    if world != args.gpus and not (world == 1 and args.gpus == 1):
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback for the shading pass")
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    config = args.config
    settings = dict(synthetic.CONFIG_SETTINGS[config])
    width = args.width or settings["width"]
    height_per_gpu = args.height or settings["height"]
    height = height_per_gpu * world
    if args.spp:
        settings["sample_count"] = args.spp
    sample_count = settings["sample_count"]
    if args.no_rays:
        settings["trace_shadow_rays"] = False

    tmp = tempfile.TemporaryDirectory(prefix="vkr_bench_%d_" % rank)
    dataset = synthetic.write_dataset(tmp.name, grid=256, box_count=64, seed=1234, ltc_resolution=32, fresnel_count=51)
    stream = torch.cuda.current_stream()
    r = renderer.Renderer(hip_device=local_rank, stream=stream.cuda_stream, fast_math=(args.mode == "fast"), inline_rays=args.inline_rays,
                          timing_stride=args.timing_stride, frames_in_flight=args.frames_in_flight)
    renderer.setup_config(r, config, dataset, width=width, height=height, sample_count=sample_count, acceleration_structure=True,
                          trace_shadow_rays=settings["trace_shadow_rays"])
    r.set_tiles(args.tile_size if distributed else 16, rank, world if distributed else 1)
    r.create_targets()
    r.create_pass()
    r.render_visibility()
    r.sync()
    light_count = r.app.scene_specification.polygonal_light_count
    techniques = 1 if settings["sampling_strategies"] == "diffuse_only" else 2

    if distributed:
        # Every rank shades its tiles into a dense slab.  What is exchanged is the pass's
        # real output, the encoded RGBA8 frame (reference: the swapchain image), a quarter of
        # the bytes of the float radiance.  The all-gather of frame k runs on RCCL's stream
        # while frame k + 1 is shaded (two sets of buffers); a frame counts as done when it
        # has been reassembled, and the timed region ends only after the last one has.
        slab_pixels = r.slab_pixel_count(0)
        # one set of buffers per frame in flight (at least two): frame k + 1 is shaded while frame k
        # is encoded and exchanged
        sets = max(2, args.frames_in_flight)
        slab = [torch.zeros((slab_pixels, 4), dtype=torch.float32, device="cuda") for _ in range(sets)]
        if args.exchange == "none":
            send = gathered = frame = None
        elif args.exchange == "rgb8":
            send = [torch.zeros(3 * slab_pixels, dtype=torch.uint8, device="cuda") for _ in range(sets)]
            gathered = [torch.zeros(world * 3 * slab_pixels, dtype=torch.uint8, device="cuda") for _ in range(sets)]
            frame = torch.zeros((height, width), dtype=torch.int32, device="cuda")
        elif args.exchange == "rgba8":
            send = [torch.zeros(slab_pixels, dtype=torch.int32, device="cuda") for _ in range(sets)]
            gathered = [torch.zeros(world * slab_pixels, dtype=torch.int32, device="cuda") for _ in range(sets)]
            frame = torch.zeros((height, width), dtype=torch.int32, device="cuda")
        else:
            send = [torch.zeros((slab_pixels, 4), dtype=torch.float32, device="cuda") for _ in range(sets)]
            gathered = [torch.zeros((world * slab_pixels, 4), dtype=torch.float32, device="cuda") for _ in range(sets)]
            frame = torch.zeros((height, width, 4), dtype=torch.float32, device="cuda")
        pending = [None] * sets
        frame_counter = [0]
        # The frames run on the device's frame streams, not on torch's stream.  Before a frame
        # overwrites buffer set b, those streams wait for the last reader of that set (the encode
        # kernel resp. the collective), via an event recorded on torch's stream.
        frame_streams = [torch.cuda.ExternalStream(int(r.app.device.frame_streams[i])) for i in range(args.frames_in_flight)] if args.frames_in_flight >= 2 else []
        readers_done = [None] * sets

        def finish(b):
            if pending[b] is not None:
                pending[b].wait()  # the compute stream waits for the collective, not the host
                if args.exchange == "rgb8":
                    r.assemble_rgb8(gathered[b].data_ptr(), frame.data_ptr())
                elif args.exchange == "rgba8":
                    r.assemble_encoded(gathered[b].data_ptr(), frame.data_ptr())
                else:
                    r.assemble(gathered[b].data_ptr(), frame.data_ptr())
                    readers_done[b] = torch.cuda.Event()
                    readers_done[b].record()
                pending[b] = None

        def step():
            if args.exchange == "none":
                return r.render(slab[0].data_ptr())
            b = frame_counter[0] % sets
            frame_counter[0] += 1
            finish(b)  # frame k - sets is complete, its buffers are free again
            if readers_done[b] is not None:
                for s in frame_streams:
                    s.wait_event(readers_done[b])
            if args.exchange in ("rgb8", "rgba8"):
                r.render(slab[b].data_ptr())
                if args.exchange == "rgb8":
                    r.encode_slab_rgb8(slab[b].data_ptr(), send[b].data_ptr(), slab_pixels)
                else:
                    r.encode_slab(slab[b].data_ptr(), send[b].data_ptr(), slab_pixels)
                readers_done[b] = torch.cuda.Event()
                readers_done[b].record()
            else:
                r.render(send[b].data_ptr())
                r.finish_frames()  # torch's stream waits for the frame before RCCL reads it
            pending[b] = dist.all_gather_into_tensor(gathered[b], send[b], async_op=True)

        def drain():
            for i in range(sets):
                finish((frame_counter[0] + i) % sets)
    else:
        def step():
            r.render()

        def drain():
            pass

    def fence():
        drain()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    issue_seconds = time.perf_counter() - t0  # host time to queue the steps (a bound if the host cannot keep up)
    fence()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    total_pixels = width * height
    value = total_pixels * sample_count / (elapsed / args.steps) / 1e6

    # ---- roofline of the shading kernel, from HIP events recorded inside the timed region --------
    timed_frames = max(1, min(args.steps // max(args.timing_stride, 1), 256))
    launch_ms = r.dispatch_ms(timed_frames)
    period_ms = r.frame_period_ms(max(1, timed_frames - 1))
    # The dominant kernel is shade_pixels; its launches are bracketed by HIP events on the stream
    # they run on (every timing_stride-th frame).  One launch of the whole pass is shade + trace +
    # resolve; with frames in flight two passes share the GPU, so the pass duration that counts
    # is the period between completions.
    pipelined = bool(r.app.shading_pass.last_frame_in_flight)
    kernel_ms = r.shading_kernel_ms(timed_frames)
    pass_ms = float(np.mean(period_ms if (pipelined and period_ms) else launch_ms)) if launch_ms else float("nan")
    kernel_avg_ms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
    visibility = r.read_visibility()
    own_pixels = r.slab_pixel_count(rank) if distributed else total_pixels
    if distributed:
        # shaded fraction of the pixels this rank owns
        import ctypes as C
        xy = np.zeros((own_pixels, 2), np.uint32)
        slots = r.lib.get_slab_pixel_coordinates(C.byref(r.app), rank, xy.ctypes.data, own_pixels)
        valid = xy[:slots, 0] != 0xFFFFFFFF
        own_visibility = visibility[xy[:slots][valid, 1], xy[:slots][valid, 0]]
    else:
        own_visibility = visibility.ravel()
    shaded = int((own_visibility != 0xFFFFFFFF).sum())
    background = int(own_visibility.size - shaded)
    bytes_per_launch = shaded * algorithmic_bytes_per_pixel(light_count, sample_count, techniques) + background * 20
    achieved = bytes_per_launch / (kernel_avg_ms * 1e-3) / 1e9
    traffic = None
    valu_floor_us = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path):
        try:
            table = json.load(open(pmc_path))
            entry = table.get("config%d_%s" % (config, args.mode))
            if entry and world == 1 and width == entry.get("width") and height == entry.get("height"):
                traffic = entry["hbm_bytes_per_launch"]
                valu_floor_us = table.get("config%d_%s_valu_floor_us" % (config, args.mode))
        except Exception:
            traffic = None
    rays = r.last_ray_count()
    traversal = None
    if args.traversal_stats and rays and not args.inline_rays:
        traversal = r.traversal_statistics()
        traversal["visits_per_ray"] = round(traversal["node_visits"] / max(traversal["rays"], 1), 2)
        traversal["tests_per_ray"] = round(traversal["triangle_tests"] / max(traversal["rays"], 1), 2)
        traversal["lane_use"] = round(traversal["node_visits"] / max(64 * traversal["wave_steps"], 1), 3)
    roofline = {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": traffic,
                "kernel_ms": round(kernel_avg_ms, 4), "pass_ms": round(pass_ms, 4),
                "achieved_over_pass": round(bytes_per_launch / (pass_ms * 1e-3) / 1e9, 3),
                "pass_latency_ms": round(float(np.mean(launch_ms)), 4) if launch_ms else None,
                "frames_in_flight": int(r.app.shading_pass.last_frame_in_flight) if pipelined else 1, "algorithmic_bytes_per_launch": bytes_per_launch,
                "kernel": "shade_pixels<%s, V=%d, rays=%d, %s>" % (settings["sampling_strategies"], r.app.shading_pass.max_polygon_vertex_count,
                                                                   int(r.app.shading_pass.use_ray_tracing), args.mode),
                "note": "kernel_ms = shade_pixels alone (dominant kernel), pass_ms = shade + trace + resolve per frame; "
                        "compute-bound pass: FP32 VALU issue and BVH latency limit it, not HBM (SURVEY.md 8d)"}

    if valu_floor_us:
        # the bound that actually holds: wave64 VALU instructions counted by the PMC
        # pass in profiles/ x 4 clocks / 1024 SIMDs / 2.4 GHz, per kernel of the pass, against the
        # live frame period
        floor_ms = sum(valu_floor_us.values()) * 1e-3
        roofline["valu_issue"] = {"floor_ms_per_pass": round(floor_ms, 4), "frac": round(floor_ms / pass_ms, 4),
                                  "shade_pixels_floor_ms": round(valu_floor_us.get("shade_pixels", 0.0) * 1e-3, 4),
                                  "source": "profiles/pmc_traffic.json (instruction counts from rocprofv3 --pmc), time live"}

    # ---- CPU baseline and parity on a bounded sample (rank 0, N = 1 only) ------------------------
    cpu_baseline = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle
        gpu_image = r.read_radiance()
        inputs = r.host_inputs(visibility)
        bvh = oracle.Bvh(inputs["quantized_positions"], inputs["dequantization_factor"], inputs["dequantization_summand"])
        frame_o = oracle.make_frame(inputs, r.oracle_settings(), bvh)
        cores = available_cpus()
        # The oracle deals 64-pixel chunks to all host threads; bands of at least `cores` rows keep
        # the per-call overhead (256 threads waking up) small next to the work.  Calibrate on one
        # band in the middle (after a call that starts the thread pool), then spread bands over
        # the frame for ~12 s of CPU time.
        band = int(min(height, max(24, cores)))
        mid = max(0, height // 2 - band // 2)
        # exact mode is compared with the oracle's matching polynomial math (bit-comparable),
        # fast mode with the libm oracle
        oracle.set_math_mode(1 if args.mode == "exact" else 0)
        oracle.shade(frame_o, mid, mid + band, cores)
        t = time.perf_counter()
        oracle.shade(frame_o, mid, mid + band, cores)
        per_row = max((time.perf_counter() - t) / band, 1e-7)
        rows_budget = int(min(height, max(band, 12.0 / per_row)))
        bands = max(1, rows_budget // band)
        starts = [int(i * (height - band) / max(bands - 1, 1)) for i in range(bands)]
        starts = sorted(set(starts))
        cpu_time = 0.0
        sq, cnt, worst, nan = 0.0, 0, 0.0, int(np.isnan(gpu_image).sum())
        flipped, sq_without_flips, mismatched = 0, 0.0, 0
        for y0 in starts:
            t = time.perf_counter()
            cpu = oracle.shade(frame_o, y0, y0 + band, cores)
            cpu_time += time.perf_counter() - t
            d = gpu_image[y0:y0 + band, :, :3].astype(np.float64) - cpu[y0:y0 + band, :, :3]
            sq += float((d ** 2).sum())
            cnt += d.size
            worst = max(worst, float(np.abs(d).max()))
            per_pixel = np.abs(d).max(axis=-1)
            flipped += int((per_pixel > 1e-2).sum())
            mismatched += int((per_pixel > 0).sum())
            sq_without_flips += float((d[per_pixel <= 1e-2] ** 2).sum())
        sample_pixels = len(starts) * band * width
        # cheap configurations: repeat the sample until about ten seconds of CPU work are timed
        passes = 1
        while cpu_time < 10.0 and passes < 4096:
            t = time.perf_counter()
            for y0 in starts:
                oracle.shade(frame_o, y0, y0 + band, cores)
            cpu_time += time.perf_counter() - t
            passes += 1
        cpu_baseline = {"value": round(passes * sample_pixels * sample_count / cpu_time / 1e6, 4), "unit": "Msamples/s", "cores": cores,
                        "kind": "port", "seconds": round(cpu_time, 2), "sample": "%d pass(es) over %d band(s) of %d rows (%d of %d pixels), CPU oracle (C99 restatement of the reference GLSL, OpenMP over 64-pixel chunks, %s)" % (passes, len(starts), band, sample_pixels, total_pixels, "deterministic polynomial math" if args.mode == "exact" else "libm")}
        oracle.set_math_mode(0)
        parity = {"rmse_vs_oracle": math.sqrt(sq / max(cnt, 1)), "max_abs": worst, "nan": nan, "tolerance_rmse": 1e-4,
                  "sample_pixels": sample_pixels, "pixels_differing": mismatched, "pixels_over_1e-2": flipped,
                  "rmse_without_those": math.sqrt(sq_without_flips / max(cnt, 1)),
                  "oracle_math": "polynomial (bit-comparable)" if args.mode == "exact" else "libm"}

    # the library reports like the reference does (printf): every rank flushes C stdio before rank 0
    # prints, so that the JSON line is the last line of the job's output
    ctypes.CDLL(None).fflush(None)
    if distributed:
        dist.barrier()
    if rank == 0:
        result = {
            "metric": "Msamples/s (pixels x spp / s), shading pass", "value": round(value, 3), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE config %d: %dx%d, %d spp per technique, %d polygonal light(s), %s + %s, %s, %s arithmetic"
                                   % (config, width, height, sample_count, light_count, settings["sampling_strategies"],
                                      settings["polygon_technique"], "LBVH shadow rays" if settings["trace_shadow_rays"] else "no shadow rays", args.mode),
                       "width": width, "height": height, "spp": sample_count, "lights": light_count, "techniques": techniques,
                       "parallelism": "tiles %dx%d round-robin over %d rank(s)%s" % (args.tile_size, args.tile_size, world, (", every rank keeps its slab of the frame (no data-path collective)" if args.exchange == "none" else " + RCCL all-gather of %s slabs overlapped with the next frame" % args.exchange) if distributed else ""),
                       "scene_triangles": int(r.app.scene.mesh.triangle_count)},
            "host_issue_ms_per_step": round(issue_seconds / args.steps * 1e3, 4),
            "shadow_rays_per_frame": rays, "Mrays_per_s": round(rays / (pass_ms * 1e-3) / 1e6, 2) if rays else 0.0,
            "roofline": roofline,
        }
        if traversal:
            result["traversal"] = traversal
        if cpu_baseline:
            result["cpu_baseline"] = cpu_baseline
            result["speedup_vs_cpu"] = round(value / cpu_baseline["value"], 1)
        if parity:
            result["parity"] = parity
        print(json.dumps(result), flush=True)
    r.close()
    if distributed:
        dist.destroy_process_group()
    tmp.cleanup()


if __name__ == "__main__":
    main()
