"""One BASELINE configuration set up, timed and described (run_workload); the same workload in the other arithmetic modes
(mode_companion); two cells of the reference's timing matrix."""
import ctypes
import json
import os
import sys
import time

import numpy as np

from .common import ROOT, algorithmic_bytes_per_pixel
from .parity import band_parity, cpu_baseline_and_parity
from .roofline import build_roofline


def protocol_window(frames_in_flight, least=8):
    """Frames between two timing brackets of a pipelined run: the smallest multiple of the frames in flight that is at
    least `least`.  Frames in flight finish in BURSTS - n shading kernels share the GPU and end together, then nothing ends
    for n frame times (profiles/r10a/frame_periods.jsonl: periods of 0.2, 0.2, 3.0 ms with three in flight) - so the time
    between the ends of two frames that are k frames apart is a whole number of bursts, and its median over windows of k
    frames is biased unless n divides k: with k = 8 and n = 3 two windows in three span three bursts, one spans two, and
    the median sits 10 % above the mean (rounds 4 and 5 reported exactly that gap between `value` and `value_from_median`)."""
    n = max(1, int(frames_in_flight))
    return ((max(1, int(least)) + n - 1) // n) * n


def run_workload(job, config, role, scene=None, exchange_format=None):
    """Sets one BASELINE configuration up, times it and returns the dict that describes the run.
    role: "primary" (the headline: CPU baseline, parity, other arithmetic modes), "extra" (a short run of another
    1920x1080 configuration with parity bits and roofline, attached to the headline line) or "secondary" (config 4)."""
    primary = role == "primary"
    from vulkan_renderer_amd import renderer, synthetic
    args, torch = job.args, job.torch
    scene = scene or args.scene
    dataset = job.dataset_of(scene)
    rank, world = job.rank, job.world
    settings = dict(synthetic.CONFIG_SETTINGS[config])
    strong = args.scaling == "strong"
    # (exchange_format: another format than --exchange, for the companion run of main())
    exchange = (exchange_format or args.exchange) if (world > 1 or args.force_distributed) else "none"
    distributed = world > 1 or args.force_distributed
    width = args.width or settings["width"]
    height = (args.height or settings["height"]) * (1 if strong else world)
    if args.spp:
        settings["sample_count"] = args.spp
    sample_count = settings["sample_count"]
    if args.no_rays:
        settings["trace_shadow_rays"] = False
    steps = args.steps if args.steps is not None else {1: 2000, 2: 2000, 3: 500, 4: 100, "target": 1000}[config]
    warmup = args.warmup if args.warmup is not None else max(steps // 10, 1)
    if role == "secondary":
        steps, warmup = max(4, min(steps, 25)), max(1, min(warmup, 5))
    elif role == "extra":
        # at least 100 timed frames, so that the reference's protocol (median of >= 100 frame times) applies
        steps, warmup = 200, 20
    frames_in_flight_requested = args.frames_in_flight or renderer.frames_in_flight_for(world if (world > 1 or args.force_distributed) else 1)
    # frames between two timing brackets: a multiple of the frames in flight (protocol_window())
    window = protocol_window(frames_in_flight_requested, args.timing_stride)
    timing_stride = 1 if steps < 4 * window else window

    # ---- set-up (untimed, reported separately: BASELINE.md section 3) -------------------------
    r = renderer.Renderer(hip_device=job.local_rank, stream=job.stream.cuda_stream, arithmetic=args.mode, inline_rays=args.inline_rays,
                          timing_stride=timing_stride, frames_in_flight=frames_in_flight_requested, binary_traversal=args.binary_traversal)
    t = time.perf_counter()
    renderer.setup_config(r, config, dataset, width=width, height=height, sample_count=sample_count, acceleration_structure=args.bvh,
                          trace_shadow_rays=settings["trace_shadow_rays"])
    r.sync()
    load_ms = (time.perf_counter() - t) * 1e3
    structure = r.app.scene.acceleration_structure
    r.set_tiles(args.tile_size if distributed else int(os.environ.get("VKR_BENCH_TILE", "0")), rank, world if distributed else 1, slab_layout=distributed)
    r.create_targets()
    r.create_pass()
    t = time.perf_counter()
    r.render_visibility()
    r.sync()
    first_visibility_ms = (time.perf_counter() - t) * 1e3
    # the first launch pays for code-object loading and buffer creation: the cost per frame is that of the later ones
    t = time.perf_counter()
    for _ in range(4):
        r.render_visibility()
    r.sync()
    visibility_ms = (time.perf_counter() - t) * 1e3 / 4
    light_count = r.app.scene_specification.polygonal_light_count
    techniques = 1 if settings["sampling_strategies"] == "diffuse_only" else 2
    total_pixels = width * height

    slab = None
    if exchange != "none":
        # the rendezvous token comes from rank 0 (ncclGetUniqueId behind the C-ABI) over the process group
        if job.backend == "nccl":
            token = job.broadcast_bytes(r.exchange_id() if rank == 0 else b"", 128)
            r.create_exchange(token, exchange)
        else:
            # CPU process group (VKR_BENCH_BACKEND=gloo: several ranks on ONE GPU, where RCCL refuses to form a
            # communicator): the same schedule with the collective staged through the host
            r.create_exchange_with_gather(job.host_staged_gather(), exchange)

        # the gathered slabs, tile-major, are the frame every rank holds; rows are made when somebody reads (the fences
        # of this run, once each) unless --assemble every-frame asks for the scatter kernel behind every all-gather
        r.assemble_on_demand(args.assemble == "on-demand")

        def step():
            r.render_and_exchange(None)

        def drain():
            r.finish_exchange()
    elif distributed:
        slab = torch.zeros((r.slab_pixel_count(0), 4), dtype=torch.float32, device="cuda")

        def step():
            r.render(slab.data_ptr())

        def drain():
            r.finish_frames()
    else:
        def step():
            r.render()

        def drain():
            r.finish_frames()

    def fence():
        drain()
        job.barrier()

    # clocks and the frame pipeline reach their steady state only after a few hundred frames (config 2:
    # 100 frames are 14 ms); the driver's --warmup 5 alone would time a cold GPU
    # Every rank must submit the SAME number of frames (each frame is one collective): the ranks agree on when the
    # prewarm ends - after a chunk of frames the slowest rank's clock decides for all.  (Until round 3 every rank
    # looked at its own clock, and a rank that fitted one frame more into the time than its peers left the job hanging
    # in its last all-gather: found by the first self-launched two-rank run, profiles/r05a/.)
    prewarm = 0
    t0 = time.perf_counter()
    prewarm_seconds = args.prewarm_seconds if role != "extra" else min(args.prewarm_seconds, 0.5)
    while prewarm < args.prewarm_frames:
        for _ in range(min(8 if prewarm == 0 else 16, args.prewarm_frames - prewarm)):
            step()
            prewarm += 1
        drain()
        torch.cuda.synchronize()
        if job.max_over_ranks(time.perf_counter() - t0) >= prewarm_seconds:
            break
    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    issue_seconds = time.perf_counter() - t0  # host time to queue the steps (a bound if the host cannot keep up)
    fence()
    elapsed = job.max_over_ranks(time.perf_counter() - t0)
    ms_per_step = elapsed / steps * 1e3
    value = total_pixels * sample_count / (elapsed / steps) / 1e6

    # ---- what the timed region looked like from the inside ---------------------------------------
    timed_frames = max(1, min(steps // max(timing_stride, 1), 256))
    overlapped_kernel_ms = r.shading_kernel_ms(timed_frames)
    period_ms = r.frame_period_ms(max(1, timed_frames - 1))
    launch_ms = r.dispatch_ms(timed_frames)
    pipelined = bool(r.app.shading_pass.last_frame_in_flight)
    bands_per_frame = int(r.app.shading_pass.last_band_count)
    frames_in_flight = int(r.app.shading_pass.last_frame_in_flight) if pipelined else 1
    rays = r.last_ray_count()
    shafts = r.light_shaft_statistics()
    # The reference's protocol is the median of at least 100 frame times (src/frame_timer.c:24,47-72, main.c:1958-1959).  A
    # run with fewer timed steps (the driver's --steps 20) renders 128 more frames behind the timed region for it; `value`
    # and `ms_per_step` stay those of the K timed steps.
    protocol_periods = None
    if primary and steps < 100:
        # sixteen more windows of `window` frames each (>= 128 frames), bracketed like the windows of a long run
        r.app.shading_pass.timing_stride = window
        for _ in range(17 * window):
            step()
        fence()
        protocol_periods = r.frame_period_ms(15)
        r.app.shading_pass.timing_stride = timing_stride
    # ---- every frame to the host (PCIe-inclusive; never `value`): a ring of targets, each read back through pinned staging
    # on the pass's copy stream while the next frames render (begin_read_back / end_read_back, include/vkr_shading_pass.h)
    with_readback = None
    if primary and not distributed and not args.no_host_frames:
        ring = [torch.empty((height, width, 4), dtype=torch.float32, device="cuda") for _ in range(frames_in_flight_requested + 1)]
        frame_bytes = width * height * 16

        def host_frames(count):
            for i in range(count):
                slot = i % len(ring)
                if i >= len(ring):
                    r.end_read_back(slot)  # the consumer takes frame i - len(ring) before its target and staging are reused
                r.render(ring[slot].data_ptr())
                r.begin_read_back(slot, ring[slot].data_ptr(), frame_bytes)
            for slot in range(min(count, len(ring))):
                r.end_read_back(slot)
        host_frames(2 * len(ring))  # (the first use of a slot allocates its pinned memory)
        frames_to_host = max(120, min(steps, 200))
        fence()
        t = time.perf_counter()
        host_frames(frames_to_host)
        fence()
        host_ms = (time.perf_counter() - t) / frames_to_host * 1e3
        with_readback = {"ms_per_frame": round(host_ms, 4), "value": round(total_pixels * sample_count / (host_ms * 1e-3) / 1e6, 3), "unit": "Msamples/s", "frames": frames_to_host,
                         "bytes_per_frame": frame_bytes, "GB_per_s": round(frame_bytes / (host_ms * 1e-3) / 1e9, 2), "targets": len(ring),
                         "note": "every frame lands in pinned host memory (RGBA32F, whole frame): render into a ring of device targets, begin_read_back() behind each frame on the pass's copy stream, end_read_back() when the ring comes round; PCIe-inclusive, never `value`"}
        del ring
    stages = None
    if exchange != "none":
        mine = r.exchange_ms() or [float("nan")] * 3
        per_rank = job.gather_floats(mine)
        stages = {"shade_ms": [round(v[0], 4) for v in per_rank], "all_gather_ms": [round(v[1], 4) for v in per_rank],
                  "scatter_ms": [round(v[2], 4) for v in per_rank],
                  "note": "per rank, HIP events of the most recent timed frame: shading (+ encoding) of the rank's slab on its frame stream, ncclAllGather and scatter on the exchange stream; they overlap the next frame, so they do not add up to ms_per_step"}
    # the assembled frame against a single-GPU render of the whole frame (rank 0)
    scaling_parity = None
    assembled = None
    if exchange != "none" and rank == 0:
        if exchange == "rgba32f":
            assembled = r.read_radiance()
        else:
            assembled = np.zeros((height, width, 4), np.uint8)
            r.lib.read_back_encoded(ctypes.byref(r.app), assembled.ctypes.data)
    if exchange != "none":
        r.destroy_exchange()
    visibility = r.read_visibility()
    own_pixels = r.slab_pixel_count(rank) if distributed else total_pixels
    if distributed:
        # shaded fraction of the pixels this rank owns
        xy = np.zeros((own_pixels, 2), np.uint32)
        slots = r.lib.get_slab_pixel_coordinates(ctypes.byref(r.app), rank, xy.ctypes.data, own_pixels)
        valid = xy[:slots, 0] != 0xFFFFFFFF
        own_visibility = visibility[xy[:slots][valid, 1], xy[:slots][valid, 0]]
    else:
        own_visibility = visibility.ravel()
    shaded = int((own_visibility != 0xFFFFFFFF).sum())
    background = int(own_visibility.size - shaded)
    bytes_per_launch = shaded * algorithmic_bytes_per_pixel(light_count, sample_count, techniques) + background * 20

    # ---- the dominant kernel alone: a short pass with one frame at a time, every frame timed -------
    # (one launch per frame, so that the events around the shading kernel bracket that kernel and nothing else)
    r.frames_in_flight, r.timing_stride, r.band_count = 1, 1, 1
    r.create_pass()
    target = slab.data_ptr() if slab is not None else None
    alone_frames = max(4, min(steps, 16))
    for _ in range(alone_frames + 3):
        r.render(target)
    r.sync()
    kernel_alone_ms = r.shading_kernel_ms(alone_frames)
    shaft_alone_ms = r.light_shaft_ms(alone_frames)
    pass_alone_ms = r.dispatch_ms(alone_frames)
    kernel_ms = float(np.mean(kernel_alone_ms)) if kernel_alone_ms else float("nan")
    traversal = None
    if (args.traversal_stats or primary or role == "extra") and rays and not args.inline_rays and world == 1:
        traversal = {}
        for wide in ([True, False] if structure.wide_nodes else [False]):
            s = r.traversal_statistics(wide)
            traversal[s["tree"]] = {"fetches_per_ray": round(s["node_visits"] / max(s["rays"], 1), 2), "boxes_tested_per_ray": round(s["boxes_tested"] / max(s["rays"], 1), 2),
                                    "triangle_tests_per_ray": round(s["triangle_tests"] / max(s["rays"], 1), 2), "lane_use": round(s["node_visits"] / max(64 * s["wave_steps"], 1), 3),
                                    "longest_ray_fetches": s["longest_ray_visits"], "blocked_fraction": round(s["blocked_rays"] / max(s["rays"], 1), 4)}
            if wide:
                visible_rays = max(s["rays"] - s["blocked_rays"], 1)
                traversal[s["tree"]].update({"fetches_per_blocked_ray": round(s["node_visits_of_blocked_rays"] / max(s["blocked_rays"], 1), 2),
                                             "fetches_per_visible_ray": round((s["node_visits"] - s["node_visits_of_blocked_rays"]) / visible_rays, 2),
                                             "deepest_stack": s["deepest_stack"], "rays_beyond_lds_stack": s["rays_beyond_lds_stack"]})
        traversal["walked"] = "wide" if (structure.wide_nodes and not args.binary_traversal) else "binary"
        traversal["note"] = ("replayed by a statistics kernel that walks the queued rays in batches of 64, each to its end: lane_use is what share of the lanes of such a batch "
                             "is busy per step - the figure by which a tracing wave decides to hand rays to idle lanes instead (below 0.65, csrc/wavefront_kernels.h)")
    if assembled is not None:
        # single-GPU render of the whole frame with the same pass settings
        r.set_tiles(16, 0, 1, slab_layout=False)
        r.render()
        single = r.read_radiance() if exchange == "rgba32f" else r.read_encoded(False, 0)
        differing = int((assembled.view(np.uint32) != single.view(np.uint32)).any(axis=-1).sum()) if exchange == "rgba32f" else int((assembled != single).any(axis=-1).sum())
        scaling_parity = {"pixels_differing_from_single_gpu_frame": differing, "pixels": total_pixels, "format": exchange}
        r.set_tiles(args.tile_size, rank, world, slab_layout=True)
    # PCIe-inclusive figures (never part of `value`): the frame to the host, a visibility buffer from the host
    t = time.perf_counter()
    gpu_image = r.read_radiance()
    pageable_readback_ms = (time.perf_counter() - t) * 1e3
    # ... and through the pinned staging of begin_read_back() / end_read_back() (second use of the slot: the first one allocates)
    r.begin_read_back(0)
    r.end_read_back(0)
    t = time.perf_counter()
    r.begin_read_back(0)
    r.end_read_back(0)
    readback_ms = (time.perf_counter() - t) * 1e3
    t = time.perf_counter()
    r.upload_visibility(visibility)
    r.sync()
    upload_ms = (time.perf_counter() - t) * 1e3

    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
    pass_ms = float(np.mean(period_ms if (pipelined and period_ms) else launch_ms)) if launch_ms else float("nan")
    # the reference's protocol (src/frame_timer.c:24,47-72, main.c:1958-1959): the median of at least 100 frame
    # times; here of the periods between the ends of consecutive timed frames inside the timed region
    median_ms = None
    if steps >= 100:
        slowest = job.max_over_ranks(float(np.median(period_ms)) if (period_ms and len(period_ms) >= 8) else -1.0)
        median_ms = slowest if slowest > 0.0 else None
    median_frames = (len(period_ms) * window) if median_ms else None
    if primary and steps < 100:
        # (every rank takes part in the reduction whatever it measured: a collective behind a local condition would hang)
        local = float(np.median(protocol_periods)) if (protocol_periods and len(protocol_periods) >= 8) else -1.0
        slowest = job.max_over_ranks(local)
        if slowest > 0.0:
            median_ms, median_frames = slowest, len(protocol_periods) * window
    roofline = build_roofline(job, r, primary, config, scene, settings, width, height, achieved=achieved, kernel_ms=kernel_ms, bytes_per_launch=bytes_per_launch, alone_frames=alone_frames,
                              pass_alone_ms=pass_alone_ms, shaft_alone_ms=shaft_alone_ms, shafts=shafts, frames_in_flight=frames_in_flight, pass_ms=pass_ms, overlapped_kernel_ms=overlapped_kernel_ms,
                              ms_per_step=ms_per_step, distributed=distributed)
    shaded_fraction = float(job.max_over_ranks(shaded / max(own_visibility.size, 1))) if distributed else shaded / max(own_visibility.size, 1)
    result = {
        "metric": "Msamples/s (pixels x spp / s), shading pass", "value": round(value, 3), "unit": "Msamples/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(ms_per_step, 4),
        "median_frame_period_ms": round(median_ms, 4) if median_ms else None, "median_over_frames": median_frames, "median_window_frames": window if median_ms else None,
        "value_from_median": round(total_pixels * sample_count / (median_ms * 1e-3) / 1e6, 3) if median_ms else None,
        "latency_ms": round(float(np.mean(pass_alone_ms)), 4) if pass_alone_ms else None,
        "value_single_frame": round(total_pixels * sample_count / (float(np.mean(pass_alone_ms)) * 1e-3) / 1e6, 3) if pass_alone_ms else None,
        "shaded_fraction": round(shaded_fraction, 4), "value_shaded_only": round(value * shaded_fraction, 3),
        "value_note": "value = W x H x spp / time over ALL pixels of the frame (SURVEY.md 8d), background included, with config.frames_in_flight frames queued like the reference's frame queue (main.h:374-390); latency_ms / value_single_frame = one frame at a time (roofline.pass_alone_ms); value_shaded_only counts the pixels that see geometry"
                      + ("; median_frame_period_ms = median over windows of median_window_frames frames - a multiple of the frames in flight, which finish in bursts - of the time between the ends of the window's first and last frame, per frame (the reference's protocol: median of >= 100 frame times, src/frame_timer.c:47-72)" if median_ms else "; the median of frame periods is reported from 100 steps on"),
        "higher_is_better": True, "scaling": args.scaling if world > 1 else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE config %s: %dx%d, %d spp per technique, %d polygonal light(s), %s + %s, %s, %s arithmetic"
                               % ("%s%s" % (config, "" if scene == "bench" else " on the large scene"), width, height, sample_count, light_count, settings["sampling_strategies"], settings["polygon_technique"],
                                  ("shadow rays through the %s BVH (%s)" % ("four-wide" if (structure.wide_nodes and not args.binary_traversal and not args.inline_rays) else "binary",
                                                                            renderer.BVH_BUILDER_NAME[int(structure.builder)])) if settings["trace_shadow_rays"] else "no shadow rays", args.mode),
                   "width": width, "height": height, "spp": sample_count, "lights": light_count, "techniques": techniques,
                   "parallelism": ("tiles %dx%d round-robin over %d rank(s), %s" % (args.tile_size, args.tile_size, world,
                                   ("RCCL all-gather of %s slabs (ncclAllGather from C, in place) %s inside the timed region, overlapped with the next frame" % (exchange, "+ scatter per frame" if args.assemble == "every-frame" else "per frame, un-tiled when read")) if exchange != "none" else "every rank keeps its slab of the frame (no data-path collective)")) if distributed else "one GPU, whole frame",
                   "scene": scene, "scene_triangles": int(r.app.scene.mesh.triangle_count), "scene_materials": int(r.app.scene.materials.material_count), "ltc_resolution": int(r.app.ltc_table.roughness_count),
                   "arithmetic": args.mode, "bands_per_frame": bands_per_frame, "frames_in_flight": frames_in_flight},
        "prewarm_frames": prewarm, "host_issue_ms_per_step": round(issue_seconds / steps * 1e3, 4),
        "shadow_rays_per_frame": rays, "Mrays_per_s": round(rays / (ms_per_step * 1e-3) / 1e6, 2) if rays else 0.0,
        "light_shafts": {"patch_light_pairs": shafts["pairs"], "clear_pairs": shafts["clear_pairs"], "clear_fraction": round(shafts["clear_pairs"] / max(shafts["pairs"], 1), 4), "occluder_list_pairs": shafts["list_pairs"], "triangles_per_occluder_list": round(shafts["listed_triangles"] / max(shafts["list_pairs"], 1), 2),
                         "not_clear": shafts["not_clear"], "work": shafts["work"],
                         "note": "csrc/light_shafts.h: one conservative BVH walk per (8x8 pixel patch, light) pair of the last launch; clear_pairs: no shadow ray of the pair can be blocked, none is queued; occluder_list_pairs: its rays can only meet the (at most 12) triangles of a list, and the shading kernel decides them against that list with the tracing kernel's triangle test; not_clear: the rays are queued and traced - shadow_rays_per_frame counts those; VKR_LIGHT_SHAFTS=0 traces all rays, VKR_SHAFT_LISTS=0 all but the clear pairs'; frames are bit-identical either way (tests/test_gpu_light_shafts.py)"},
        "setup": {"load_and_upload_ms": round(load_ms - structure.build_milliseconds, 2), "bvh_build_ms": round(float(structure.build_milliseconds), 3),
                  "bvh_builder": renderer.BVH_BUILDER_NAME[int(structure.builder)], "bvh_node_bytes": 16 * int(structure.node_count) + 64 * int(structure.wide_node_count),
                  "bvh_wide_nodes": int(structure.wide_node_count), "bvh_stack_need": int(structure.wide_stack_need), "visibility_pass_ms": round(visibility_ms, 3), "first_visibility_pass_ms": round(first_visibility_ms, 3),
                  "readback_ms": round(readback_ms, 3), "pageable_readback_ms": round(pageable_readback_ms, 3), "upload_ms": round(upload_ms, 3),
                  "note": "untimed set-up, once per scene (load = parse .vks / LTC fits / noise + copies to the device); visibility_pass_ms = primary visibility per frame (mean of 4 launches after the first, host clock around a synchronised device), first_visibility_pass_ms includes one-time costs; readback = RGBA32F frame into pinned staging (begin_read_back + end_read_back, nothing else running), pageable_readback = read_back_radiance() into pageable memory, upload = a visibility buffer from the host; with_readback (top level) = frames per second when EVERY frame goes to the host while the next ones render; never part of value"},
        "roofline": roofline,
    }
    if with_readback:
        result["with_readback"] = with_readback
    if stages:
        result["stages"] = stages
    if scaling_parity:
        result["scaling_parity"] = scaling_parity
    if traversal:
        result["traversal"] = traversal

    # ---- CPU baseline and parity (rank 0, N = 1) ----------------------------------------------------------
    if rank == 0 and world == 1 and not distributed and not args.no_cpu_baseline:
        if primary:
            result.update(cpu_baseline_and_parity(args, r, visibility, gpu_image, width, height, sample_count, value))
        elif role == "extra":
            result.update(band_parity(args, r, visibility, gpu_image, config, width, height, sample_count))
    if primary and rank == 0 and world == 1 and not distributed and args.mode != "fast" and not args.no_other_modes and not args.inline_rays and not args.no_rays:
        r.close()
        result["other_modes"] = {}
        for other in ("exact",) if args.no_fast_mode else ("exact", "fast"):
            if other != args.mode:
                result["other_modes"][other] = mode_companion(job, config, other, gpu_image, width, height, sample_count, max(20, min(steps, 200)), frames_in_flight_requested)
        return result
    r.close()
    return result


def mode_companion(job, config, mode, headline_image, width, height, sample_count, steps, frames_in_flight):
    """The same workload in one of the cheaper arithmetic modes - exact: polynomial transcendentals, IEEE
    otherwise; fast: v_rcp / v_rsq / v_sqrt and contraction -, timed the same way and compared with the
    headline frame of this run (libm mode: the oracle's, bit for bit) under the same outlier rule.
    Reported next to the headline, never as the headline."""
    from vulkan_renderer_amd import renderer
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import classify_outliers
    args, torch = job.args, job.torch
    r = renderer.Renderer(hip_device=job.local_rank, stream=job.stream.cuda_stream, arithmetic=mode, timing_stride=protocol_window(frames_in_flight, args.timing_stride), frames_in_flight=frames_in_flight)
    renderer.setup_config(r, config, job.dataset, width=width, height=height, sample_count=sample_count, acceleration_structure=args.bvh)
    r.set_tiles(0, 0, 1, slab_layout=False)
    r.create_targets()
    r.create_pass()
    r.render_visibility()
    for _ in range(max(8, steps // 10)):
        r.render()
    r.finish_frames()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r.render()
    r.finish_frames()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    image = r.read_radiance()
    r.close()
    stats = classify_outliers(image, headline_image)
    if stats["pixels_over_threshold"] != stats["guard_pixels"]:
        # some outlier is not a NaN-guard pixel: the two modes' frames WITHOUT shadow rays tell a silhouette (a ray that passes
        # a triangle edge on the other side: the frames agree at that pixel once no ray is traced) from anything else
        without = {}
        for m in (mode, args.mode):
            q = renderer.Renderer(hip_device=job.local_rank, stream=job.stream.cuda_stream, arithmetic=m, frames_in_flight=1)
            renderer.setup_config(q, config, job.dataset, width=width, height=height, sample_count=sample_count, acceleration_structure=args.bvh, trace_shadow_rays=False)
            q.create_targets()
            q.create_pass()
            q.render_visibility()
            q.render()
            without[m] = q.read_radiance()
            q.close()
        stats = classify_outliers(image, headline_image, without[mode], without[args.mode])
    stats.pop("other_coordinates", None)
    # the rule of DESIGN.md section 2: every pixel that differs by more than 1e-2 is a NaN-guard pixel or a shadow-ray
    # silhouette, the rest is within 1e-4 RMSE
    within = bool(stats["rmse_without_outliers"] <= 1e-4 and stats["other_pixels"] == 0 and not np.isnan(image).any())
    return {"mode": mode, "within_tolerance": within, "value": round(width * height * sample_count / (ms * 1e-3) / 1e6, 3), "unit": "Msamples/s", "ms_per_step": round(ms, 4), "steps": steps,
            "rmse": stats["rmse"], "rmse_without_discontinuity_pixels": stats["rmse_without_outliers"], "guard_pixels": stats["guard_pixels"], "silhouette_pixels": stats["silhouette_pixels"], "other_pixels": stats["other_pixels"],
            "vs_headline_frame": stats, "nan": int(np.isnan(image).sum()), "tolerance_rmse": 1e-4,
            "note": "against the headline frame of this run (libm: the oracle's, bit for bit); pixels over 1e-2 are classified as NaN-guard pixels, shadow-ray silhouettes (the two modes agree there without shadow rays) or `other`, which fails the tolerance"}


def timing_matrix_cells(job):
    """Two cells of the reference's own timing matrix (src/experiment_list.c:366-409; all 260: profiles/tools/timing_matrix.py
    and profiles/r07h_timing_matrix.md), measured in this run by the reference's protocol - median frame time of 110 frames -
    with the matrix's settings: 1920x1080, diffuse only, projected solid angle sampling, no shadow rays, a decentral quad;
    128 lights x 1 sample and 1 light x 128 samples.  For the details file."""
    import importlib.util
    from vulkan_renderer_amd import renderer, synthetic
    spec = importlib.util.spec_from_file_location("timing_matrix", os.path.join(ROOT, "profiles", "tools", "timing_matrix.py"))
    matrix = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(matrix)
    dataset = job.dataset_of("bench")
    cells = {}
    r = renderer.Renderer(hip_device=job.local_rank, stream=job.stream.cuda_stream, frames_in_flight=1, timing_stride=1, arithmetic=job.args.mode)
    try:
        r.load_scene(dataset["scene"], dataset["textures"], acceleration_structure=True)
        r.load_ltc_table(dataset["ltc"], dataset["fresnel_count"])
        r.load_noise_table("white")
        cam = synthetic.DEFAULT_CAMERA
        r.set_camera(cam["position"], cam["rotation_x"], cam["rotation_z"], cam["vertical_fov"], cam["near"], cam["far"])
        r.set_settings(width=1920, height=1080, trace_shadow_rays=False, sampling_strategies="diffuse_only", polygon_technique="projected_solid_angle")
        r.set_lights(matrix.timing_lights(4, False, 1))
        r.create_targets()
        r.create_pass()
        r.render_visibility()
        for light_count, sample_count in ((128, 1), (1, 128)):
            r.set_settings(sample_count=sample_count)
            r.set_lights(matrix.timing_lights(4, False, light_count))
            r.create_pass()
            for _ in range(8):
                r.render()
            r.sync()
            for _ in range(110):
                r.render()
            r.sync()
            times = sorted(r.dispatch_ms(110))
            median = times[len(times) // 2]
            cells["%d_lights_x_%d_samples" % (light_count, sample_count)] = {
                "frame_ms": round(median, 4), "light_samples_per_s": round(1920 * 1080 * light_count * sample_count / (median * 1e-3), 0),
                "experiment": "timings_decentral_4%s_projected_solid_angle_ours" % ("_128" if light_count == 128 else "")}
    finally:
        # (also when a launch fails: the config-4 secondary that follows needs the memory)
        r.close()
    cells["protocol"] = "median of 110 frame times, one frame at a time (no shadow rays: a frame is one kernel), %s arithmetic; scene, noise and lights are stand-ins for the reference's downloaded assets (profiles/tools/timing_matrix.py)" % job.args.mode
    return cells
