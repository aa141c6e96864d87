import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The light shafts of the shading pass (csrc/light_shafts.h) decide by a conservative test which shadow rays need not be
# traced.  The pass switches them on by itself only where they pay (many rays per pixel); the test-suite forces them on
# everywhere, so that every parity test against the oracle also checks that no ray was skipped wrongly.
# (tests/test_gpu_light_shafts.py compares on / off / automatic explicitly.)
os.environ.setdefault("VKR_LIGHT_SHAFTS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


@pytest.fixture(scope="session")
def dataset(tmp_path_factory):
    """Small synthetic scene + LTC fits + material textures in the reference's formats."""
    from vulkan_renderer_amd import synthetic
    d = tmp_path_factory.mktemp("dataset")
    return synthetic.write_dataset(str(d), grid=64, box_count=24, seed=1234, ltc_resolution=16, fresnel_count=8)


@pytest.fixture(scope="session")
def big_dataset(tmp_path_factory):
    """The benchmark scene (2 * 256^2 ground triangles + 64 boxes)."""
    from vulkan_renderer_amd import synthetic
    d = tmp_path_factory.mktemp("big_dataset")
    return synthetic.write_dataset(str(d), grid=256, box_count=64, seed=1234, ltc_resolution=64, fresnel_count=51)
