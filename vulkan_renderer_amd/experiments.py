"""Runs entries of the experiment table, like `vulkan_renderer -e<N>` of the reference
(src/main.c:2277-2284 parses the flag, startup_application :1909-1925 applies the
experiment, advance_experiments :1948-2016 renders, times and takes the screenshot).

    python -m vulkan_renderer_amd.experiments -e 25 --data-root /path/with/data
    python -m vulkan_renderer_amd.experiments -e 25 --synthetic /tmp/vkr_data

Scene files, quicksaves, LTC fits and noise tables are looked up below the data root
with the reference's relative paths (data/attic.vks, data/quicksaves/..., data/ggx_ltc_fit,
data/noise/...).  --synthetic writes a generated stand-in data set (one synthetic scene
under every scene name, white noise instead of the Ahmed table, default light) so the
whole procedure can be exercised without the downloaded assets.  The table, the path
handling, the screenshot encoders and the file writers are C code in libvkr_shading.so;
this module only sequences the calls."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

from . import capi, renderer, synthetic


def experiment_table(lib=None):
    lib = lib or capi.load()
    table = capi.ExperimentList()
    lib.create_experiment_list(C.byref(table))
    return table


def write_synthetic_data_root(root, grid=128, box_count=32):
    """A data root with the reference's layout where every scene is the synthetic scene."""
    data = os.path.join(root, "data")
    made = synthetic.write_dataset(os.path.join(root, "synthetic"), grid=grid, box_count=box_count)
    lib = capi.load()
    paths = (C.c_char_p * 4 * 9).in_dll(lib, "g_scene_paths")
    for scene in range(9):
        scene_path = os.path.join(root, paths[scene][1].decode())
        texture_path = os.path.join(root, paths[scene][2].decode())
        os.makedirs(os.path.dirname(scene_path), exist_ok=True)
        for source, target in ((made["scene"], scene_path), (made["textures"], texture_path)):
            if not os.path.lexists(target):
                os.symlink(os.path.abspath(source), target)
    ltc = os.path.join(data, "ggx_ltc_fit")
    if not os.path.lexists(ltc):
        os.symlink(os.path.abspath(made["ltc"]), ltc)
    os.makedirs(os.path.join(data, "quicksaves"), exist_ok=True)
    os.makedirs(os.path.join(data, "experiments"), exist_ok=True)
    return {"fresnel_count": made["fresnel_count"]}


def run_experiment(index, data_root, frames=32, warmup=4, synthetic_inputs=False, fresnel_count=51, hdr=False, hip_device=0, verbose=True):
    """Renders experiment `index` and stores its screenshot.  Returns a dict with the
    frame time and the screenshot path, or raises with the library's message."""
    lib = capi.load()
    table = experiment_table(lib)
    try:
        if not 0 <= index < table.count:
            raise ValueError("experiment index %d is not in [0, %d)" % (index, table.count))
        experiment = table.experiments[index]
        r = renderer.Renderer(hip_device=hip_device)
        try:
            if lib.apply_experiment(C.byref(r.app), C.byref(experiment), data_root.encode()):
                raise RuntimeError("apply_experiment failed")
            spec = r.app.scene_specification
            settings = r.app.render_settings
            if synthetic_inputs:
                # no Ahmed / blue noise tables without the downloaded data
                settings.noise_type = 0
                if spec.polygonal_light_count == 0:
                    r.set_lights(synthetic.config_lights(2))
                    cam = synthetic.DEFAULT_CAMERA
                    r.set_camera(cam["position"], cam["rotation_x"], cam["rotation_z"], cam["vertical_fov"], cam["near"], cam["far"])
            wants_rays = bool(settings.trace_shadow_rays)
            r.load_scene(C.string_at(spec.file_path).decode(), C.string_at(spec.texture_path).decode(), acceleration_structure=True)
            r.load_ltc_table(os.path.join(data_root, "data", "ggx_ltc_fit"), fresnel_count)
            cwd = os.getcwd()
            os.chdir(data_root)  # noise tables are addressed relative to the working directory (noise_table.c)
            try:
                r.load_noise_table(int(settings.noise_type))
                # the quicksave may hold textured lights (reference update_application, main.c:1879);
                # their paths are relative to the working directory too
                r.create_light_textures()
            finally:
                os.chdir(cwd)
            r.create_targets()
            r.create_pass()
            r.render_visibility()
            for _ in range(warmup):
                r.render()
            r.sync()
            for _ in range(frames):
                r.render()
            r.sync()
            times = sorted(r.dispatch_ms(frames))
            frame_ms = times[len(times) // 2]
            screenshot = C.string_at(experiment.screenshot_path).decode()
            if hdr:
                screenshot = screenshot[:-3] + "hdr"
            pointer = lib.format_screenshot_path(os.path.join(data_root, screenshot).encode(), frame_ms)
            path = C.string_at(pointer).decode()
            C.CDLL(None).free(C.c_void_p(pointer))
            os.makedirs(os.path.dirname(path), exist_ok=True)
            if lib.take_screenshot(C.byref(r.app), None if hdr else path.encode(), path.encode() if hdr else None):
                raise RuntimeError("take_screenshot failed for %s" % path)
            result = {"index": index, "frame_ms": frame_ms, "screenshot": path, "width": r.app.swapchain.extent.width,
                      "height": r.app.swapchain.extent.height, "rays": bool(wants_rays and r.app.shading_pass.use_ray_tracing),
                      "Msamples_per_s": r.app.swapchain.extent.width * r.app.swapchain.extent.height * settings.sample_count / (frame_ms * 1e-3) / 1e6}
            if verbose:
                print(result)
            return result
        finally:
            r.close()
    finally:
        lib.destroy_experiment_list(C.byref(table))


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-e", "--experiment", type=int, default=None, help="index into the experiment table; omit to list the table")
    ap.add_argument("--data-root", default=".", help="directory that contains data/ (the reference's working directory)")
    ap.add_argument("--synthetic", metavar="DIR", default=None, help="write a generated stand-in data root to DIR and use it")
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--hdr", action="store_true", help="store *.hdr instead of *.png (take_hdr_screenshots of the reference)")
    args = ap.parse_args(argv)
    lib = capi.load()
    if args.experiment is None:
        table = experiment_table(lib)
        for i in range(table.count):
            print("%03d: %s" % (i, table.experiments[i].screenshot_path.decode()))
        lib.destroy_experiment_list(C.byref(table))
        return 0
    fresnel_count = 51
    root = args.data_root
    if args.synthetic:
        root = args.synthetic
        fresnel_count = write_synthetic_data_root(root)["fresnel_count"]
    run_experiment(args.experiment, root, frames=args.frames, synthetic_inputs=bool(args.synthetic), fresnel_count=fresnel_count, hdr=args.hdr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
