#!/usr/bin/env python3
"""The reference's own timing matrix (src/experiment_list.c:366-409) on one MI355X.

    gpurun --timeout 900 -- 'python profiles/tools/timing_matrix.py --out gpurun_out/r02g/timing_matrix'

The experiments come from the experiment table of the library (create_experiment_list, checked
against the reference's compiled table in tests/test_experiments.py): 1920x1080, diffuse only, no
shadow rays, {128 lights x 1 sample, 1 light x 128 samples} x {central, decentral} x 3 ... 7
vertices x the 13 polygon sampling techniques, exposure / light count.  Frame time = median over
110 frames after the experiment was set up (src/main.c:1958-1959 runs an experiment for 110 frames
and one second, src/frame_timer.c:47-72 reports the median), measured with HIP events around the
frame's kernels.

What is NOT the reference's: the downloaded assets.  The scene is the synthetic stand-in for
"roughness planes" (ground plane of 2 x 256^2 triangles + 64 boxes), the noise is white instead of
the Ahmed table, and the quicksaves data/quicksaves/roughness_planes_{central,decentral}_<n>[_128].save
are replaced by generated lights: a regular n-gon of 6 m diameter 3 m above the ground facing down
(central: the zenith of most visible points is inside the polygon) or standing upright beside the
scene (decentral for every point); the 128-light variants repeat that light 128 times with a
millimetre of jitter, so that the geometric case stays what the file name says."""
import argparse
import ctypes as C
import json
import math
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from vulkan_renderer_amd import capi, renderer, synthetic  # noqa: E402


def timing_lights(vertex_count, central, light_count):
    rng = np.random.default_rng(vertex_count * 2 + int(central))
    polygon = synthetic.regular_polygon(vertex_count, 0.5, 0.3)
    lights = []
    for k in range(light_count):
        jitter = rng.uniform(-1.0e-3, 1.0e-3, 3) if k else np.zeros(3)
        if central:
            # plane space (0..1)^2 scaled to 6 m, centred over the middle of the scene, normal pointing down
            translation = np.array([-3.0, 1.0 + 3.0, 3.0]) + jitter
            rotation = (math.pi, 0.0, 0.0)
        else:
            translation = np.array([-6.5, -1.0, 0.2]) + jitter
            rotation = (0.5 * math.pi, 0.0, 0.5 * math.pi)
        lights.append(synthetic.light_spec(polygon, tuple(translation), rotation, (12.0, 11.0, 10.0), (6.0, 6.0)))
    return lights


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/timing_matrix")
    ap.add_argument("--frames", type=int, default=110)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--only", default=None, help="substring of the screenshot path, to run part of the matrix")
    ap.add_argument("--mode", default="libm", choices=["libm", "exact", "fast"], help="arithmetic mode of the kernels (libm: the default of the pass, bit-identical to the pinned oracle)")
    args = ap.parse_args()
    lib = capi.load()
    table = capi.ExperimentList()
    lib.create_experiment_list(C.byref(table))
    names = (C.c_char_p * 13).in_dll(lib, "g_sample_polygon_name") if hasattr(lib, "g_sample_polygon_name") else None
    tmp = tempfile.TemporaryDirectory(prefix="vkr_timings_")
    dataset = synthetic.write_dataset(tmp.name, grid=256, box_count=64, seed=1234, ltc_resolution=64, fresnel_count=51)
    # (the matrix renders without shadow rays: nothing to pipeline, every frame is one kernel on one stream)
    r = renderer.Renderer(frames_in_flight=1, timing_stride=1, arithmetic=args.mode)
    r.load_scene(dataset["scene"], dataset["textures"], acceleration_structure=True)  # primary visibility walks the BVH
    r.load_ltc_table(dataset["ltc"], dataset["fresnel_count"])
    r.load_noise_table("white")
    cam = synthetic.DEFAULT_CAMERA
    r.set_camera(cam["position"], cam["rotation_x"], cam["rotation_z"], cam["vertical_fov"], cam["near"], cam["far"])
    r.set_settings(width=1920, height=1080, trace_shadow_rays=False)
    r.set_lights(timing_lights(3, True, 1))
    r.create_targets()
    r.create_pass()
    r.render_visibility()
    results = []
    started = time.perf_counter()
    for index in range(table.count):
        experiment = table.experiments[index]
        path = experiment.screenshot_path.decode()
        if "/timings_" not in path or (args.only and args.only not in path):
            continue
        # data/experiments/timings_<central|decentral>_<n>[_128]_<technique>_%.3f.png
        stem = os.path.basename(path)[len("timings_"):-len("_%.3f.png")]
        pieces = stem.split("_")
        central = pieces[0] == "central"
        vertex_count = int(pieces[1])
        many_lights = pieces[2] == "128"
        technique = "_".join(pieces[3 if many_lights else 2:])
        light_count = 128 if many_lights else 1
        assert (experiment.width, experiment.height) == (1920, 1080)
        C.memmove(C.byref(r.app.render_settings), C.byref(experiment.render_settings), C.sizeof(r.app.render_settings))
        settings = r.app.render_settings
        settings.noise_type = 0  # white: the Ahmed table is a downloaded asset
        assert not settings.trace_shadow_rays and settings.sample_count == (1 if many_lights else 128)
        r.set_lights(timing_lights(vertex_count, central, light_count))
        try:
            r.create_pass()
            for _ in range(args.warmup):
                r.render()
            r.sync()
            for _ in range(args.frames):
                r.render()
            r.sync()
            times = sorted(r.dispatch_ms(args.frames))
            frame_ms = times[len(times) // 2]
        except RuntimeError as error:
            print("experiment %d (%s): %s" % (index, stem, error), flush=True)
            continue
        samples = 1920 * 1080 * settings.sample_count * light_count
        results.append({"index": index, "name": stem, "technique": technique, "technique_index": int(settings.polygon_sampling_technique), "vertex_count": vertex_count,
                        "central": central, "light_count": light_count, "sample_count": int(settings.sample_count), "frame_ms": round(frame_ms, 4),
                        "min_ms": round(times[0], 4), "max_ms": round(times[-1], 4), "light_samples_per_s": round(samples / (frame_ms * 1e-3), 0)})
        print("%3d %-58s %8.3f ms" % (index, stem, frame_ms), flush=True)
    elapsed = time.perf_counter() - started
    r.close()
    lib.destroy_experiment_list(C.byref(table))
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    json.dump({"frames": args.frames, "seconds": round(elapsed, 1), "arithmetic": args.mode, "results": results}, open(args.out + ".json", "w"), indent=1)
    # two tables like the paper's: rows = technique, columns = geometric case x vertex count
    lines = ["# Timing matrix of the reference (src/experiment_list.c:366-409) on one MI355X", "",
             "`python profiles/tools/timing_matrix.py` - see its docstring for what is the reference's (the experiment table: 1920x1080, diffuse only,",
             "no shadow rays, settings per technique) and what is a stand-in (scene, noise, lights).  Median frame time over %d frames in ms," % args.frames,
             "%s arithmetic, one frame at a time (no shadow rays in these experiments: a frame is one kernel, there is nothing to overlap).  %d experiments in %.0f s." % (args.mode, len(results), elapsed), ""]
    techniques = []
    for entry in results:
        if entry["technique"] not in techniques:
            techniques.append(entry["technique"])
    for light_count, title in ((128, "128 lights x 1 sample per light"), (1, "1 light x 128 samples")):
        lines += ["## %s" % title, "", "| technique | " + " | ".join("%s %d" % (case, n) for case in ("central", "decentral") for n in range(3, 8)) + " |",
                  "|---|" + "---|" * 10]
        for technique in techniques:
            cells = []
            for central in (True, False):
                for n in range(3, 8):
                    match = [e for e in results if e["technique"] == technique and e["central"] == central and e["vertex_count"] == n and e["light_count"] == light_count]
                    cells.append("%.3f" % match[0]["frame_ms"] if match else "-")
            lines.append("| %s | %s |" % (technique, " | ".join(cells)))
        lines.append("")
    open(args.out + ".md", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
