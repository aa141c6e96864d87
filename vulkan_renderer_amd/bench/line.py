"""The ONE line bench.py prints - the contract's keys, numbers and short identifiers, at most LINE_LIMIT characters - and the
details file that carries everything else."""
import json
import os
import sys

from .common import ROOT

LINE_LIMIT = 4096  # the driver keeps 8 KB of stdout; round 4's 25 KB line could not be parsed from that


def _pick(source, keys):
    return {k: source[k] for k in keys if source and k in source and source[k] is not None}


def _short_roofline(roofline):
    """bound / achieved / peak / unit / frac / traffic (the contract) + which kernel, its duration alone and the two
    fractions that say what really bounds it - numbers only, the sources are in the details file"""
    if not roofline:
        return None
    out = {k: roofline.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
    if roofline.get("nominal_bound"):
        out["nominal_bound"] = roofline["nominal_bound"]
    # (kernel_ms brackets shade_pixels alone; the shaft kernel that runs in front of it since round 4 is named next to it)
    out.update(_pick(roofline, ("kernel", "kernel_ms", "light_shaft_kernel_ms", "algorithmic_bytes_per_launch", "pass_alone_ms")))
    if roofline.get("flops"):
        out["flops"] = _pick(roofline["flops"], ("achieved", "peak", "unit", "frac"))
    if roofline.get("valu_issue"):
        out["valu_issue"] = _pick(roofline["valu_issue"], ("shade_pixels_frac", "frac_of_ms_per_step"))
    return out


def _short_parity(parity):
    if not parity:
        return None
    out = {"pixels_differing": parity.get("pixels_differing", parity.get("pixels_differing_in_bits")),
           "rmse": parity.get("rmse_vs_libm_oracle", parity.get("rmse_vs_oracle")), "tolerance_rmse": 1e-4}
    out.update(_pick(parity, ("sample_pixels", "nan", "within_tolerance")))
    if out["rmse"] is None:
        del out["rmse"]
    return out


def _short_workload(w):
    """a workload other than the headline: what it is, its value, its time, its roofline fraction, its parity"""
    out = {"workload": "%dx%d, %d spp, %d light(s)%s" % (w["config"]["width"], w["config"]["height"], w["config"]["spp"], w["config"]["lights"],
                                                        "" if w["config"].get("scene", "bench") == "bench" else ", %s scene" % w["config"]["scene"])}
    out.update(_pick(w, ("value", "steps", "ms_per_step", "median_frame_period_ms")))
    if w.get("roofline"):
        out["roofline"] = _pick(w["roofline"], ("frac", "traffic", "kernel_ms"))
    if w.get("parity"):
        out["parity"] = _pick(_short_parity(w["parity"]), ("pixels_differing", "sample_pixels", "within_tolerance"))
    if w.get("scaling_parity"):
        out["scaling_parity"] = _pick(w["scaling_parity"], ("pixels_differing_from_single_gpu_frame", "pixels"))
    if w.get("cpu_only"):
        out["cpu_only"] = _pick(w["cpu_only"], ("value", "cores"))
    return out


def short_line(result, details_path=None):
    """The ONE line bench.py prints: the contract's keys, numbers and short identifiers only (no prose), at most
    LINE_LIMIT characters.  Everything else the run measured - extra workloads in full, traversal and light-shaft
    statistics, set-up times, the other arithmetic mode, where each number comes from - goes to the details file."""
    line = _pick(result, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "median_frame_period_ms", "median_over_frames", "median_window_frames", "value_from_median",
                          "higher_is_better", "scaling", "dtype", "data"))
    line["vs_baseline"] = result.get("vs_baseline")
    line.update(_pick(result, ("value_shaded_only", "shaded_fraction", "latency_ms")))
    if result.get("with_readback"):
        line["value_with_readback"] = result["with_readback"].get("value")
    cfg = result.get("config", {})
    line["config"] = _pick(cfg, ("workload", "width", "height", "spp", "lights", "techniques", "scene", "scene_triangles", "arithmetic", "frames_in_flight", "parallelism"))
    for key in ("workload", "parallelism"):
        if len(str(line["config"].get(key, ""))) > 240:
            line["config"][key] = line["config"][key][:240]
    line["roofline"] = _short_roofline(result.get("roofline"))
    if result.get("cpu_baseline"):
        line["cpu_baseline"] = _pick(result["cpu_baseline"], ("value", "unit", "cores", "kind", "cpu", "seconds"))
        line["cpu_baseline"]["sample"] = str(result["cpu_baseline"].get("sample", ""))[:120]
    if result.get("parity"):
        line["parity"] = _short_parity(result["parity"])
    if result.get("scaling_parity"):
        line["scaling_parity"] = _pick(result["scaling_parity"], ("pixels_differing_from_single_gpu_frame", "pixels", "format"))
    if result.get("stages"):
        line["stages"] = {k: [round(v, 3) for v in result["stages"][k]] for k in ("shade_ms", "all_gather_ms", "scatter_ms") if k in result["stages"]}
    if result.get("north_star_target"):
        target = result["north_star_target"]
        line["north_star_target"] = _pick(target, ("shape", "target_Msamples_per_s", "value", "met"))
        if target.get("parity"):
            line["north_star_target"]["pixels_differing"] = _short_parity(target["parity"])["pixels_differing"]
    if result.get("exchange_rgb8"):
        line["exchange_rgb8"] = _pick(result["exchange_rgb8"], ("value", "ms_per_step"))
        if result["exchange_rgb8"].get("scaling_parity"):
            line["exchange_rgb8"]["pixels_differing"] = result["exchange_rgb8"]["scaling_parity"].get("pixels_differing_from_single_gpu_frame")
    if result.get("secondary"):
        line["secondary"] = _short_workload(result["secondary"])
    extras = result.get("extra_workloads") or {}
    if extras:
        line["extra_workloads"] = {name: _pick(_short_workload(w), ("value", "ms_per_step", "parity", "cpu_only")) for name, w in extras.items()}
    if result.get("other_modes"):
        line["other_modes"] = {m: _pick(v, ("value", "ms_per_step", "within_tolerance", "rmse", "rmse_without_discontinuity_pixels", "guard_pixels", "silhouette_pixels", "other_pixels")) for m, v in result["other_modes"].items()}
    line["details"] = details_path
    # a long workload string is the first thing to go if the line ever outgrows the driver's buffer
    for drop in ("other_modes", "extra_workloads", "stages"):
        if len(json.dumps(line, separators=(",", ":"))) <= LINE_LIMIT:
            break
        line.pop(drop, None)
    if len(json.dumps(line, separators=(",", ":"))) > LINE_LIMIT:
        line["config"]["workload"] = str(line["config"].get("workload", ""))[:160]
    return line


def write_details(result, path=None):
    """Everything the run measured, with the prose: gpurun_out/bench_details.json (scratch that gpurun brings back;
    copies that are meant to be judged are committed under profiles/).  Returns the path relative to the repository."""
    path = path or os.path.join(ROOT, "gpurun_out", "bench_details.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(result, f, indent=1)
            f.write("\n")
    except OSError as e:
        print("bench.py: could not write %s: %s" % (path, e), file=sys.stderr)
        return None
    return os.path.relpath(path, ROOT)
