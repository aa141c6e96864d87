"""The GPU frames against the CPU oracle, and the oracle timed as the CPU baseline.  oracle/ is test infrastructure: only
tests/, __graft_entry__.smoke() and this leg of the benchmark call it, as the checker - never as the thing measured."""
import os
import sys
import time

import numpy as np

from .common import ROOT, available_cpus, cpu_model, libm_identity


def cpu_baseline_and_parity(args, r, visibility, gpu_image, width, height, sample_count, value):
    """The headline workload against the CPU oracle (test infrastructure: oracle/, the checker - never the thing measured):
    `cpu_baseline` (the oracle timed on the host cores over a bounded sample of the frame), `speedup_vs_cpu` and `parity` (the
    GPU frame against the oracle's frame over the same rows).  -> the keys to merge into the record."""
    from vulkan_renderer_amd import renderer
    result = {}
    total_pixels = width * height
    import oracle
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import classify_outliers
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
    # The oracle in its libm mode is the arithmetic that is pinned bit for bit against the reference's
    # shader source (tests/test_reference_live.py) - what the default mode of the kernels reproduces
    # and what every mode is measured against; the polynomial "exact" mode is also compared with the
    # oracle's matching polynomial mode (bit-comparable).
    matching_mode = renderer.ORACLE_MATH_MODE[args.mode]
    oracle.set_math_mode(matching_mode)
    oracle.shade(frame_o, mid, mid + band, cores)
    t = time.perf_counter()
    oracle.shade(frame_o, mid, mid + band, cores)
    per_row = max((time.perf_counter() - t) / band, 1e-7)
    rows_budget = int(min(height, max(band, 12.0 / per_row)))
    bands = max(1, rows_budget // band)
    starts = sorted(set(int(i * (height - band) / max(bands - 1, 1)) for i in range(bands)))
    cpu_time = 0.0
    covered = np.zeros(height, bool)
    cpu_frames = {matching_mode: np.zeros((height, width, 4), np.float32)}
    for y0 in starts:
        t = time.perf_counter()
        cpu = oracle.shade(frame_o, y0, y0 + band, cores)
        cpu_time += time.perf_counter() - t
        cpu_frames[matching_mode][y0:y0 + band] = cpu[y0:y0 + band]
        covered[y0:y0 + band] = True
    sample_pixels = int(covered.sum()) * width
    timed_pixels = len(starts) * band * width
    # cheap configurations: repeat the sample until about ten seconds of CPU work are timed
    passes = 1
    while cpu_time < 10.0 and passes < 4096:
        t = time.perf_counter()
        for y0 in starts:
            oracle.shade(frame_o, y0, y0 + band, cores)
        cpu_time += time.perf_counter() - t
        passes += 1
    result["cpu_baseline"] = {"value": round(passes * timed_pixels * sample_count / cpu_time / 1e6, 4), "unit": "Msamples/s", "cores": cores, "cpu": cpu_model(),
                              "kind": "port", "seconds": round(cpu_time, 2),
                              "sample": "%d passes over %d bands of %d rows: %d of %d pixels of the frame" % (passes, len(starts), band, sample_pixels, total_pixels),
                              "implementation": "CPU oracle (C99 restatement of the reference GLSL, OpenMP over 64-pixel chunks, %s)" % ("deterministic polynomial math" if matching_mode == 1 else "libm math: the mode that is bit-identical to the reference's shader source compiled as C++")}
    result["speedup_vs_cpu"] = round(value / result["cpu_baseline"]["value"], 1)
    # the other oracle mode over the same rows (untimed)
    if matching_mode != 0:
        oracle.set_math_mode(0)
        cpu_frames[0] = np.zeros((height, width, 4), np.float32)
        for y0 in starts:
            cpu_frames[0][y0:y0 + band] = oracle.shade(frame_o, y0, y0 + band, cores)[y0:y0 + band]
    oracle.set_math_mode(0)

    def against(cpu_frame):
        g, c = gpu_image[covered], cpu_frame[covered]
        stats = classify_outliers(g, c)
        stats.pop("other_coordinates", None)
        stats["pixels_differing_in_bits"] = int((g[..., :3].view(np.uint32) != c[..., :3].view(np.uint32)).any(axis=-1).sum())
        stats["max_abs"] = float(np.abs(np.nan_to_num(g[..., :3].astype(np.float64) - c[..., :3], nan=1e3)).max())
        return stats

    libm = against(cpu_frames[0])
    result["parity"] = {
        "tolerance_rmse": 1e-4, "sample_pixels": sample_pixels, "nan": int(np.isnan(gpu_image).sum()),
        "vs_libm_oracle": libm,
        "rmse_vs_libm_oracle": libm["rmse"], "pixels_over_1e-2": libm["pixels_over_threshold"], "guard_pixels": libm["guard_pixels"],
        "libm_oracle": "oracle math mode 0: C library transcendentals, IEEE division / sqrt; bit-identical to the reference's GLSL compiled as C++ (tests/test_reference_live.py, tests/test_oracle_golden.py); what --mode libm reproduces bit for bit",
        "libm": oracle.libm_description() + "; this machine: " + libm_identity(),
        "rule": "RMSE <= 1e-4 over all pixels that do not sit on a discontinuity of the shader; every pixel that differs by more than 1e-2 is a NaN-guard pixel (shading_pass.frag.glsl:861-864) "
                "or a shadow-ray silhouette, else it counts as `other_pixels` and the run is out of tolerance (tests/helpers.py classify_outliers; silhouettes need the frames without rays: tests/test_gpu_full_size.py)",
        "within_tolerance": bool(libm["rmse_without_outliers"] <= 1e-4 and (libm["pixels_over_threshold"] == libm["guard_pixels"])),
    }
    if matching_mode != 0:
        result["parity"]["vs_polynomial_oracle"] = against(cpu_frames[matching_mode])
        result["parity"]["polynomial_oracle"] = "oracle math mode 1: the polynomial transcendentals that --mode exact mirrors operation for operation (bit-comparable)"
    # (kept for readers of earlier rounds' lines: the comparison with the oracle mode that matches --mode)
    matched = result["parity"]["vs_polynomial_oracle"] if matching_mode != 0 else libm
    result["parity"]["rmse_vs_oracle"] = matched["rmse"]
    result["parity"]["pixels_differing"] = matched["pixels_differing_in_bits"]
    return result


def band_parity(args, r, visibility, gpu_image, config, width, height, sample_count):
    """A workload other than the headline: three bands of its frame (config 1: the whole frame, and the oracle timed beside it
    as `cpu_only`) against the oracle in the arithmetic that the mode reproduces.  -> the keys to merge into the record."""
    from vulkan_renderer_amd import renderer
    result = {}
    total_pixels = width * height
    # three bands of the frame against the oracle in the arithmetic the mode reproduces (libm: every bit)
    import oracle
    inputs = r.host_inputs(visibility)
    bvh = oracle.Bvh(inputs["quantized_positions"], inputs["dequantization_factor"], inputs["dequantization_summand"])
    frame_o = oracle.make_frame(inputs, r.oracle_settings(), bvh)
    cores = available_cpus()
    band = int(min(height, 24))
    starts = sorted(set(int(f * (height - band)) for f in (1 / 6, 1 / 2, 5 / 6)))
    if config == 1:
        # BASELINE configs[0] is the CPU-runnable case: the WHOLE frame against the oracle, and the oracle timed beside it below
        band, starts = height, [0]
    oracle.set_math_mode(renderer.ORACLE_MATH_MODE[args.mode])
    differing, compared, worst = 0, 0, 0.0
    for y0 in starts:
        cpu = oracle.shade(frame_o, y0, y0 + band, cores)[y0:y0 + band]
        g = gpu_image[y0:y0 + band]
        differing += int((g[..., :3].view(np.uint32) != cpu[..., :3].view(np.uint32)).any(axis=-1).sum())
        compared += band * width
        worst = max(worst, float(np.abs(np.nan_to_num(g[..., :3].astype(np.float64) - cpu[..., :3], nan=1e3)).max()))
    if config == 1:
        # "CPU C reference of polygon_sampling math (plumbing, no GPU)": the same frame on the host cores, about two seconds of it
        passes, cpu_time = 0, 0.0
        while cpu_time < 2.0 and passes < 4096:
            t = time.perf_counter()
            oracle.shade(frame_o, 0, height, cores)
            cpu_time += time.perf_counter() - t
            passes += 1
        result["cpu_only"] = {"value": round(passes * total_pixels * sample_count / cpu_time / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port", "seconds": round(cpu_time, 2),
                              "ms_per_frame": round(cpu_time / passes * 1e3, 3), "sample": "%d passes over the whole %dx%d frame" % (passes, width, height)}
    oracle.set_math_mode(0)
    result["parity"] = {"sample_pixels": compared, "sample": "%d bands of %d rows" % (len(starts), band), "pixels_differing_in_bits": differing, "max_abs": worst,
                        "nan": int(np.isnan(gpu_image).sum()), "oracle_math_mode": renderer.ORACLE_MATH_MODE[args.mode],
                        "within_tolerance": bool(differing == 0) if args.mode != "fast" else None}
    return result
