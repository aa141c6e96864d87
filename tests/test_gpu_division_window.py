"""The two data-dependent assumptions of the kernels' arithmetic, checked on data.

divide() and square_root() of csrc/device_math.h are the correctly rounded IEEE results only inside an exponent
window (|b| in [2^-126, 2^126), a = 0 or |a| >= 2^-103, |a / b| in [2^-126, 2^96); roots of numbers that are 0 or
normal).  Whether an operand of the pass can leave it was argued in a comment.  Here a second build of the library
(libvkr_shading_ieee.so, `make ieee`: the libm-mode kernels with every quotient and root the compiler's full-range
IEEE expansion) renders the golden cases, a slice of the random sweep, both scenes at BASELINE sizes and a set of
extreme operands; every frame must have the digest of the product's frame.  (The extreme frames are also compared
with the oracle, bit for bit, in the product build.)"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import golden_cases
from helpers import compare, oracle_render
from vulkan_renderer_amd import renderer, synthetic

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
IEEE_LIBRARY = os.path.join(ROOT, "vulkan_renderer_amd", "libvkr_shading_ieee.so")


@pytest.fixture(scope="module")
def datasets(tmp_path_factory, big_dataset):
    d = tmp_path_factory.mktemp("window")
    small = synthetic.write_dataset(str(d / "small"), **golden_cases.DATASET)
    large = synthetic.write_dataset(str(d / "large"), seed=4321, ltc_resolution=32, fresnel_count=16, large={})
    paths = {}
    for name, dataset in (("small", small), ("benchmark", big_dataset), ("large", large)):
        paths[name] = str(d / (name + ".json"))
        json.dump(dataset, open(paths[name], "w"))
    return d, paths, small


def frames_of(library, directory, paths, name):
    env = dict(os.environ)
    env.pop("VKR_SHADING_LIBRARY", None)
    if library:
        env["VKR_SHADING_LIBRARY"] = library
    out = str(directory / (name + ".json"))
    done = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "division_window_frames.py"), out, paths["small"], paths["benchmark"], paths["large"]],
                          env=env, capture_output=True, text=True, timeout=1500)
    assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-2000:]
    return json.load(open(out))


def test_full_range_ieee_division_everywhere_changes_no_frame(datasets):
    directory, paths, _ = datasets
    assert os.path.exists(IEEE_LIBRARY), "make -C vulkan_renderer_amd/csrc ieee (build() does)"
    product = frames_of(None, directory, paths, "product")
    check = frames_of(IEEE_LIBRARY, directory, paths, "ieee")
    assert any(path.endswith("libvkr_shading.so") for path in product["library"]) and not any("ieee" in path for path in product["library"]), product["library"]
    assert any(path.endswith("libvkr_shading_ieee.so") for path in check["library"]), check["library"]
    assert product["frames"].keys() == check["frames"].keys() and len(product["frames"]) >= 100
    differing = {key: (product["frames"][key], check["frames"][key]) for key in product["frames"] if product["frames"][key]["sha256"] != check["frames"][key]["sha256"]}
    # fluxes of 1e-20 and 1e+20 are far beyond any scene: reported (where the window ends on real operands), not required
    beyond = {key: value for key, value in differing.items() if "/beyond_" in key}
    print("frames beyond any scene that differ:", sorted(beyond))
    differing = {key: value for key, value in differing.items() if key not in beyond}
    print({group: sum(1 for key in product["frames"] if key.startswith(group)) for group in ("golden", "sweep", "extreme", "benchmark", "large")})
    assert not differing, differing
    # (the frames show something: a list of black frames would prove nothing)
    lit = [key for key, frame in product["frames"].items() if frame["lit"] > 0.02]
    assert len(lit) >= 0.8 * len(product["frames"]), sorted(set(product["frames"]) - set(lit))
    assert all(frame["nan"] == 0 and frame["inf"] == 0 for frame in product["frames"].values())


def test_extreme_operands_equal_the_oracle_in_every_bit(datasets):
    """the same extreme cases against the reference-pinned oracle (libm mode), product build"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import division_window_frames
    _, _, small = datasets
    failures = []
    for case in division_window_frames.extreme_cases():
        if case["key"].startswith("beyond_"):
            continue
        r = renderer.Renderer()
        golden_cases.apply_case(r, case, small, 96, 64)
        if case.get("settings"):
            r.set_settings(**case["settings"])
        r.create_targets()
        r.create_pass()
        r.render_visibility()
        r.render()
        image = r.read_radiance()
        cpu, _, _ = oracle_render(r, visibility=r.read_visibility(), math_mode=0)
        r.close()
        stats = compare(image, cpu)
        if not stats["bit_exact"] or stats["nan"]:
            failures.append((case["key"], stats))
    assert not failures, failures
