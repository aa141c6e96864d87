"""When oracle/_ref is present (built from /root/reference in the authoring
container; the prebuilt objects travel to the GPU box) compare the oracle with
the reference's shader on FRESH inputs that are not part of the golden files."""
import numpy as np
import pytest

import golden_cases
import oracle
from oracle import reference

pytestmark = pytest.mark.skipif(not reference.available(), reason="oracle/_ref has not been built")


@pytest.fixture(scope="module")
def other_dataset(tmp_path_factory):
    from vulkan_renderer_amd import synthetic
    return synthetic.write_dataset(str(tmp_path_factory.mktemp("live")), grid=40, box_count=20, seed=99, ltc_resolution=12, fresnel_count=5)


@pytest.mark.parametrize("case", golden_cases.FRAME_CASES[:11], ids=[c["key"] for c in golden_cases.FRAME_CASES[:11]])
def test_oracle_equals_reference_on_fresh_scene(case, other_dataset):
    hs, frame, name = golden_cases.build_frame(case, other_dataset, width=56, height=40)
    a = oracle.shade(frame)
    b = reference.shade(name, frame)
    rays_a = oracle.last_ray_count()
    hs.close()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "max abs diff %g" % np.abs(a - b).max()
    if case.get("rays"):
        assert rays_a == reference.shader(name).ref_last_ray_count() and rays_a > 0


def test_random_polygons_through_both_samplers():
    rng = np.random.default_rng(777)
    checked = 0
    while checked < 300:
        n = int(rng.integers(3, 8))
        name = golden_cases.capacity_variant(n)
        poly = golden_cases.random_polygon(rng, n)
        count, buf = reference.clip_polygon(name, poly, n)
        ocount, obuf = oracle.clip_polygon(poly, max_count=n + 1)
        assert count == ocount
        if count == 0:
            continue
        assert np.array_equal(buf[:count], obuf[:count])
        state = reference.psa_prepare(name, buf, count)
        ostate, cap = oracle.psa_prepare(obuf, count=count, max_count=n + 1)
        assert np.float32(state[48]) == np.float32(ostate[48]) or (np.isnan(state[48]) and np.isnan(ostate[48]))
        if not state[48] > 0:
            continue
        for _ in range(3):
            u0, u1 = map(float, rng.uniform(0, 1, 2).astype(np.float32))
            a = reference.psa_sample(name, state, u0, u1)
            b = oracle.psa_sample(state, cap, u0, u1)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        checked += 1
