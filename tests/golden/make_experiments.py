"""TEST INFRASTRUCTURE.  Generates tests/golden/experiments.json and the image-writer
fixtures in tests/golden/writers.npz from the reference itself:
  * the experiment table = create_experiment_list() of the reference's
    src/experiment_list.c, compiled unmodified into oracle/_ref/libref_host.so,
  * a small float image as written by the reference's vendored stb_image_write.h
    (stbi_write_hdr), kept as the raw file bytes, plus half_to_float over all 65536
    bit patterns.
Run from the repository root in the build container (needs /root/reference):
    make -C oracle all && python tests/golden/make_experiments.py"""
import ctypes as C
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import reference  # noqa: E402

FIELDS = ["width", "height", "scene_index", "use_hdr", "quick_save_path", "screenshot_path", "exposure_factor",
          "roughness_factor", "sample_count", "sampling_strategies", "mis_heuristic", "mis_visibility_estimate",
          "polygon_sampling_technique", "error_display", "error_min_exponent", "noise_type", "animate_noise",
          "trace_shadow_rays", "show_polygonal_lights", "show_gui", "v_sync"]


def writer_test_image():
    """64 x 24 float image with smooth parts, constant runs, zeros, tiny and huge values"""
    rng = np.random.default_rng(77)
    image = rng.random((24, 64, 3)).astype(np.float32) * 4.0
    image[4:8] = 0.25
    image[8:10, :, 1] = 0.0
    image[10] = 0.0
    image[11, :32] = 1.0e-33
    image[12, 10:50] = np.linspace(0.0, 3.0e4, 40, dtype=np.float32)[:, None]
    image[13] = image[13, ::-1] * 1.0e-6
    return image


def main():
    experiments = reference.experiments()
    with open(os.path.join(ROOT, "tests", "golden", "experiments.json"), "w") as f:
        json.dump({"fields": FIELDS, "experiments": [[e[k] for k in FIELDS] for e in experiments]}, f, separators=(",", ":"))
    image = writer_test_image()
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "ref.hdr")
        reference.write_hdr(path, image)
        hdr_bytes = np.frombuffer(open(path, "rb").read(), np.uint8)
    halves = np.array([reference.half_to_float_bits(h) for h in range(65536)], np.uint32)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "writers.npz"), image=image, reference_hdr_file=hdr_bytes, half_to_float_bits=halves)
    print("wrote %d experiments, %d bytes of reference .hdr" % (len(experiments), hdr_bytes.size))


if __name__ == "__main__":
    main()
