"""The `roofline` object of a workload's record (bench.py contract: bound / achieved / peak / unit / frac / traffic)."""
import json
import os
import sys
import tempfile

import numpy as np

from .common import BENCH_PY, FP32_VECTOR_PEAK_TFLOPS, HBM_PEAK_GBPS, ROOT, kernel_source_hash, pmc_entry_for


def live_traffic(job, config, scene, timeout=150):
    """HBM bytes per launch of the dominant kernel, MEASURED IN THIS RUN: two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do
    not fit one pass: MI355X_MICROARCH.md, PMC slots) around a short child run of this file on the same workload - one frame at a
    time, one launch per frame -, counters averaged over the dispatches of shade_pixels.  FETCH_SIZE counts wide coalesced reads
    at half their bytes on gfx950 (the guide's HBM section): doubled.  None if rocprofv3 is not usable here (the caller then falls
    back to the committed passes of profiles/pmc_traffic.json and says so)."""
    import csv
    import glob
    import shutil
    import subprocess
    args = job.args
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof) or any(k.startswith(("ROCPROFILER_", "ROCP_TOOL")) for k in os.environ):
        return None
    out = tempfile.mkdtemp(prefix="vkr_bench_pmc_")
    child = [sys.executable, BENCH_PY, "--config", str(config), "--scene", scene, "--mode", args.mode, "--bvh", args.bvh, "--ltc-resolution", str(args.ltc_resolution),
             "--no-cpu-baseline", "--no-secondary", "--no-other-modes", "--no-extra", "--no-live-pmc", "--no-host-frames", "--frames-in-flight", "1", "--steps", "6", "--warmup", "2", "--prewarm-frames", "8",
             "--details", os.path.join(out, "child_details.json")]
    env = dict(os.environ, TMPDIR="/tmp", VKR_BENCH_DATASET_CACHE=job.dataset_cache or job.tmp.name, VKR_BAND_COUNT="1")
    for key in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)
    sums = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            directory = os.path.join(out, counter)
            command = [rocprof, "--kernel-trace", "--kernel-include-regex", "shade_pixels", "--output-format", "csv", "--pmc", counter, "-d", directory, "-o", "pmc", "--"] + child
            done = subprocess.run(command, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout)
            values = []
            for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(path)):
                    if "shade_pixels" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                        values.append(float(row["Counter_Value"]))
            if done.returncode != 0 or not values:
                return None
            sums[counter] = (sum(values) / len(values), len(values))
    except Exception:
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)
    fetch_kib, write_kib = sums["FETCH_SIZE"][0], sums["WRITE_SIZE"][0]
    return {"hbm_bytes_per_launch": int((2.0 * fetch_kib + write_kib) * 1024), "fetch_size_kib": round(fetch_kib, 1), "write_size_kib": round(write_kib, 1),
            "dispatches": [sums["FETCH_SIZE"][1], sums["WRITE_SIZE"][1]],
            "source": "measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over %d / %d dispatches of the kernel (one frame at a time), 2 x FETCH_SIZE + WRITE_SIZE (the guide's correction for wide reads on gfx950)" % (sums["FETCH_SIZE"][1], sums["WRITE_SIZE"][1])}


def build_roofline(job, r, primary, config, scene, settings, width, height, *, achieved, kernel_ms, bytes_per_launch, alone_frames, pass_alone_ms, shaft_alone_ms, shafts,
                   frames_in_flight, pass_ms, overlapped_kernel_ms, ms_per_step, distributed):
    """The `roofline` object of a workload's record: the nominal HBM figures SURVEY.md 8(d) prescribes (algorithmic bytes over the
    dominant kernel's duration alone, against 8 TB/s), the HBM traffic of that kernel - measured in this run for the headline
    workload (live_traffic), else from the committed passes of the same kernel sources - and what actually binds the kernel:
    FP32 throughput and VALU issue from the per-class instruction counts of profiles/pmc_traffic.json."""
    args, rank, world = job.args, job.rank, job.world
    pmc = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path):
        try:
            pmc = pmc_entry_for(json.load(open(pmc_path)), config, args.mode, scene, width, height, world, kernel_source_hash())
        except Exception:
            pmc = None
    # bound: what binds the dominant kernel - the issue of its VALU instructions (valu_issue below).  achieved /
    # peak / frac are the NOMINAL HBM figures SURVEY.md 8(d) prescribes for the metric (algorithmic bytes over the kernel's
    # duration against 8 TB/s): nominal_bound says so.
    live = None
    if primary and world == 1 and not distributed and not args.no_live_pmc and rank == 0:
        r.sync()
        live = live_traffic(job, config, scene)
    roofline = {"bound": "valu_issue", "nominal_bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 6),
                "traffic": live["hbm_bytes_per_launch"] if live else (pmc["hbm_bytes_per_launch"] if (pmc and not pmc["stale"]) else None),
                "traffic_live": live,
                "traffic_source": live["source"] if live else (("%s: rocprofv3 --pmc passes of this configuration and arithmetic mode (profiles/collect.sh), kernel sources %s" % (pmc.get("source", "profiles/pmc_traffic.json"), pmc.get("csrc_hash")))
                                   if not pmc["stale"] else "profiles/pmc_traffic.json has an entry, but for other kernel sources (%s, now %s): not attached" % (pmc.get("csrc_hash"), kernel_source_hash())) if pmc else None,
                "kernel": "shade_pixels<%s, V=%d, rays=%d, %s>" % (settings["sampling_strategies"], r.app.shading_pass.max_polygon_vertex_count, int(r.app.shading_pass.use_ray_tracing), args.mode),
                "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes_per_launch": bytes_per_launch,
                "kernel_ms_source": "HIP events around the kernel on its stream, %d frames with one frame at a time (nothing else on the GPU), run right after the timed region" % alone_frames,
                "pass_alone_ms": round(float(np.mean(pass_alone_ms)), 4) if pass_alone_ms else None,
                "light_shaft_kernel_ms": round(float(np.mean(shaft_alone_ms)), 4) if (shaft_alone_ms and shafts["pairs"]) else None,
                "overlapped": {"frames_in_flight": frames_in_flight, "frame_period_ms": round(pass_ms, 4),
                               "kernel_bracket_ms": round(float(np.mean(overlapped_kernel_ms)), 4) if overlapped_kernel_ms else None,
                               "note": "inside the timed region %d frames share the GPU: the bracket of one frame's shade_pixels then spans time in which the other frames' trace and resolve kernels run too, so it can exceed ms_per_step; it is not used for `achieved`" % frames_in_flight},
                "note": "nominal roofline (SURVEY.md 8d): the pass is bound by FP32 VALU issue and BVH latency, not by HBM"}
    if pmc and not pmc["stale"] and pmc.get("fp32_flop_per_launch"):
        # FP32 arithmetic of the dominant kernel: (ADD + MUL + 2 FMA) wave instructions x 64 lanes from the PMC passes over its
        # duration alone, against the FP32 vector peak
        tflops = pmc["fp32_flop_per_launch"] / (kernel_ms * 1e-3) / 1e12
        roofline["flops"] = {"achieved": round(tflops, 2), "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tflops / FP32_VECTOR_PEAK_TFLOPS, 4),
                             "fp32_flop_per_launch": pmc["fp32_flop_per_launch"], "source": "SQ_INSTS_VALU_{ADD,MUL,FMA}_F32 of %s (all 64 lanes counted), kernel sources %s; time live" % (pmc.get("source"), pmc.get("csrc_hash"))}
    if pmc and not pmc["stale"] and pmc.get("valu_floor_us"):
        # the bound that actually holds: wave64 VALU instructions per class, counted by the PMC passes in
        # profiles/, priced with the issue cost measured per class on this GPU (profiles/tools/valu_rate.hip:
        # 2.5 clocks add / mul / fma, 8.2 transcendental, 2.5 - 4.3 the rest; the mid-point is used), on 1024 SIMDs at 2.4 GHz
        floors = {k: v for k, v in pmc["valu_floor_us"].items() if isinstance(v, (int, float))}
        roofline["valu_issue"] = {"shade_pixels_floor_ms": round(floors.get("shade_pixels", 0.0) * 1e-3, 4),
                                  "shade_pixels_frac": round(floors.get("shade_pixels", 0.0) * 1e-3 / kernel_ms, 4),
                                  "floor_ms_per_pass": round(sum(floors.values()) * 1e-3, 4), "frac_of_ms_per_step": round(sum(floors.values()) * 1e-3 / ms_per_step, 4),
                                  "source": "per-class instruction counts from %s (rocprofv3 --pmc, not measured in this run) x issue clocks per class from profiles/r02d_valu_rate.txt; times live" % pmc.get("source", "profiles/pmc_traffic.json")}

    return roofline
