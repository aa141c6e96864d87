"""SURVEY.md 8(f) rank 2 on the GPU: screenshots of a rendered frame and one full
experiment (table entry -> scene, settings, timing, screenshot file)."""
import ctypes as C
import glob
import os
import subprocess

import numpy as np
import pytest

import golden_cases
from test_experiments import decode_hdr, decode_png
from test_gpu_golden import render_case
from vulkan_renderer_amd import experiments, synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def golden_dataset(tmp_path_factory):
    return synthetic.write_dataset(str(tmp_path_factory.mktemp("dataset")), **golden_cases.DATASET)


def test_screenshots_hold_the_encoded_frame(golden_dataset, tmp_path):
    case = golden_cases.FRAME_CASES[3]
    r, radiance = render_case(case, golden_dataset, False, 200, 120)
    png, hdr = str(tmp_path / "shot.png"), str(tmp_path / "shot.hdr")
    assert r.lib.take_screenshot(C.byref(r.app), png.encode(), hdr.encode()) == 0
    # the PNG holds the sRGB-encoded frame without alpha (reference main.c:1664-1674)
    assert np.array_equal(decode_png(open(png, "rb").read()), r.read_encoded(False, 0)[..., :3])
    # the HDR file holds the colour rounded to half precision (two frames with frame_bits 1 and 2
    # in the reference, main.c:1700-1711), stored as RGBE
    half = radiance[..., :3].astype(np.float16).astype(np.float32)
    rgbe = decode_hdr(open(hdr, "rb").read())
    largest = half.max(axis=-1)
    mantissa, exponent = np.frexp(largest)
    scale = np.where(largest >= 1e-32, mantissa * 256.0 / np.where(largest > 0, largest, 1.0), 0.0).astype(np.float32)
    expected = np.zeros_like(rgbe)
    expected[..., :3] = (half * scale[..., None]).astype(np.uint8)
    expected[..., 3] = np.where(largest >= 1e-32, exponent + 128, 0)
    expected[largest < 1e-32] = 0
    assert np.array_equal(rgbe, expected)
    assert r.app.screenshot.frame_bits == 0
    r.close()


def test_experiment_from_the_table_runs_end_to_end(tmp_path):
    """mis_plane with the clamped optimal heuristic (entry 34), on a generated stand-in data root"""
    root = str(tmp_path / "root")
    made = experiments.write_synthetic_data_root(root, grid=64, box_count=16)
    table = experiments.experiment_table()
    index = next(i for i in range(table.count) if table.experiments[i].screenshot_path == b"data/experiments/mis_plane_clamped_optimal_ours_2spp_%.3f.png")
    result = experiments.run_experiment(index, root, frames=6, warmup=2, synthetic_inputs=True, fresnel_count=made["fresnel_count"], verbose=False)
    assert (result["width"], result["height"]) == (1024, 1024) and result["rays"]
    files = glob.glob(os.path.join(root, "data", "experiments", "mis_plane_clamped_optimal_ours_2spp_*.png"))
    assert files == [result["screenshot"]]
    assert os.path.basename(files[0]) == "mis_plane_clamped_optimal_ours_2spp_%.3f.png" % result["frame_ms"]
    image = decode_png(open(files[0], "rb").read())
    assert image.shape == (1024, 1024, 3) and image.max() > 0
    # the sampling-error figure of the paper (entry 5): error display, and the out-of-range
    # sampling strategy that the reference's table carries (experiment_list.c:107)
    result = experiments.run_experiment(5, root, frames=2, warmup=1, synthetic_inputs=True, fresnel_count=made["fresnel_count"], verbose=False)
    assert os.path.basename(result["screenshot"]).startswith("error_attic_backward_") and not result["rays"]
    colors = np.unique(decode_png(open(result["screenshot"], "rb").read()).reshape(-1, 3), axis=0)
    assert 2 <= len(colors) <= 22  # background + a subset of the 20 colour bins
    # the related-work samplers of the comparison figures run too (entry 49: Arvo's projected solid
    # angle sampling in the Cornell box)
    result = experiments.run_experiment(49, root, frames=2, warmup=1, synthetic_inputs=True, fresnel_count=made["fresnel_count"], verbose=False)
    assert os.path.basename(result["screenshot"]).startswith("cornell_box_projected_solid_angle_arvo_1spp_")
    assert decode_png(open(result["screenshot"], "rb").read()).max() > 0


def test_c_program_on_the_c_abi_reproduces_the_python_runner(tmp_path):
    """vkr_experiment (csrc/examples/vkr_experiment.c): the reference's -e<N> run as a plain C
    program on libvkr_shading.so.  Same experiment, same generated data root (with a quicksave
    so that both load the same camera and lights), same number of frames: same screenshot."""
    import subprocess
    from vulkan_renderer_amd import renderer
    binary = os.path.join(os.path.dirname(renderer.__file__), "vkr_experiment")
    assert os.path.exists(binary), "run make -C vulkan_renderer_amd/csrc (build() does)"
    root = str(tmp_path / "root")
    made = experiments.write_synthetic_data_root(root, grid=64, box_count=16)
    table = experiments.experiment_table()
    index = next(i for i in range(table.count) if table.experiments[i].screenshot_path == b"data/experiments/mis_plane_clamped_optimal_ours_2spp_%.3f.png")
    # a quicksave in the place where the experiment looks for it (reference quick_save, main.c:49-80)
    hs = renderer.HostScene()
    hs.set_lights(golden_cases.QUADS[:2])
    cam = synthetic.DEFAULT_CAMERA
    hs.set_camera(cam["position"], cam["rotation_x"], cam["rotation_z"], cam["vertical_fov"], cam["near"], cam["far"])
    save = C.create_string_buffer(os.path.join(root, "data", "quicksaves", "mis_plane.save").encode())
    hs.app.scene_specification.quick_save_path = C.cast(save, C.c_void_p)
    hs.lib.quick_save(C.byref(hs.app.scene_specification))
    hs.app.scene_specification.quick_save_path = None
    hs.close()
    result = experiments.run_experiment(index, root, frames=16, warmup=8, synthetic_inputs=True, fresnel_count=made["fresnel_count"], verbose=False)
    expected = decode_png(open(result["screenshot"], "rb").read())
    os.remove(result["screenshot"])
    done = subprocess.run([binary, "-e%d" % index, "--frames", "16", "--white-noise", "--fresnel", str(made["fresnel_count"]), root], capture_output=True, text=True, timeout=300)
    assert done.returncode == 0, done.stdout + done.stderr
    files = glob.glob(os.path.join(root, "data", "experiments", "mis_plane_clamped_optimal_ours_2spp_*.png"))
    assert len(files) == 1 and "Msamples/s" in done.stdout
    image = decode_png(open(files[0], "rb").read())
    assert image.shape == expected.shape and image.max() > 0
    assert np.array_equal(image, expected)


def test_bench_launches_two_ranks_by_itself_on_one_gpu(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it (the driver's scaling run would have ended
    with "--gpus 8 does not match WORLD_SIZE 1"): two ranks share the one GPU, the process group is gloo, the slabs travel
    through the host (RCCL refuses two ranks on one device); tiles, slabs, exchange, scatter and the bookkeeping of
    bench.py are the code of an N-GPU run.  One JSON line, the assembled frame equal to the single-GPU frame."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "VKR_BENCH_SELF_LAUNCHED")}
    env.update({"VKR_BENCH_DEVICE": "0", "VKR_BENCH_BACKEND": "gloo"})
    done = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "3", "--prewarm-frames", "24", "--prewarm-seconds", "0.2",
                           "--no-secondary", "--no-cpu-baseline", "--no-extra", "--ltc-resolution", "16"], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert done.returncode == 0, done.stdout[-1500:] + done.stderr[-1500:]
    lines = [json.loads(line) for line in done.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1
    line = lines[0]
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    assert line["scaling_parity"]["pixels_differing_from_single_gpu_frame"] == 0
    assert len(line["stages"]["shade_ms"]) == 2
