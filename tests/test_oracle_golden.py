"""The CPU oracle against golden vectors produced by the reference itself
(tests/golden/make_golden.py: the reference's GLSL compiled as C++).  These pin the
oracle; they need neither a GPU nor the reference checkout."""
import os

import numpy as np
import pytest

import golden_cases
import oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def golden_dataset(tmp_path_factory):
    from vulkan_renderer_amd import synthetic
    return synthetic.write_dataset(str(tmp_path_factory.mktemp("golden_dataset")), **golden_cases.DATASET)


@pytest.fixture(scope="module")
def frames():
    return np.load(os.path.join(GOLDEN, "frames.npz"))


@pytest.fixture(scope="module")
def functions():
    return np.load(os.path.join(GOLDEN, "functions.npz"))


@pytest.mark.parametrize("case", golden_cases.FRAME_CASES, ids=[c["key"] for c in golden_cases.FRAME_CASES])
def test_frame_matches_reference_shader(case, golden_dataset, frames):
    hs, frame, _ = golden_cases.build_frame(case, golden_dataset)
    image = oracle.shade(frame)
    hs.close()
    expected = frames[case["key"]]
    if case.get("output_linear_rgb", True):
        # linear output: the oracle restates the shader with libm, so the bits agree
        assert np.array_equal(image.view(np.uint32), expected.view(np.uint32)), \
            "max abs diff %g" % np.abs(image - expected).max()
    else:
        # sRGB transfer applied in the shader, stored in an UNORM8 target
        stored = np.clip(expected, 0.0, 1.0)
        stored = (stored * 255.0 + 0.5).astype(np.uint8)
        assert np.array_equal(oracle.encode_srgb8(image)[..., :3], stored[..., :3])


def test_clip_polygon_every_sign_mask(functions):
    for v, n, expected, count in zip(functions["clip_in"], functions["clip_n"], functions["clip_out"], functions["clip_count"]):
        got_count, buf = oracle.clip_polygon(v[:n], max_count=n + 1)
        assert got_count == count
        if count:
            assert np.array_equal(buf[:count + (1 if count < n + 1 else 0)], expected[:count + (1 if count < n + 1 else 0)])


def test_projected_solid_angle_prepare_sample_error(functions):
    checked = 0
    for v, (n, count), state, us, dirs, errs in zip(functions["psa_in"], functions["psa_n"], functions["psa_state"],
                                                   functions["psa_u"], functions["psa_dir"], functions["psa_err"]):
        got, cap = oracle.psa_prepare(v[:n + 1], count=int(count), max_count=int(n) + 1)
        used = [0, 37, 38, 48] + [1 + i for i in range(2 * count)] + [19 + i for i in range(2 * count)]
        central = state[37] > 0
        used += [39 + i for i in range(count if central else count - 1)]
        assert np.array_equal(got[used].view(np.uint32), state[used].view(np.uint32))
        for (u0, u1), d, e in zip(us, dirs, errs):
            gd = oracle.psa_sample(state, cap, u0, u1)
            assert np.array_equal(gd.view(np.uint32), d.view(np.uint32))
            ge = oracle.psa_error(state, cap, u0, u1, d)
            assert np.allclose(ge, e, rtol=0, atol=0, equal_nan=True)
            checked += 1
    assert checked >= 2000


def test_solid_angle_sampling(functions):
    for v, n, pos, u, d, sa in zip(functions["sa_in"], functions["sa_n"], functions["sa_pos"], functions["sa_u"],
                                   functions["sa_dir"], functions["sa_value"]):
        got_sa, got_d = oracle.solid_angle_sample(v[:n], pos, u[0], u[1], max_count=4)
        assert np.float32(got_sa) == np.float32(sa)
        assert np.array_equal(got_d.view(np.uint32), d.view(np.uint32))


def test_scalar_helpers(functions):
    L = oracle.lib()
    for x, y in zip(functions["atan_x"], functions["atan_fast"]):
        assert np.float32(L.oracle_fast_positive_atan(float(x))) == y
    for k, y in zip(functions["kahan_in"], functions["kahan_out"]):
        r = np.float32(L.oracle_kahan(*map(float, k)))
        assert r == y or (np.isnan(r) and np.isnan(y))
    import ctypes as C
    fp = C.POINTER(C.c_float)
    fac, summ = functions["position_factor"].copy(), functions["position_summand"].copy()
    out = np.zeros(3, np.float32)
    for q, expected in zip(functions["position_q"], functions["position_out"]):
        L.oracle_decode_position(int(q[0]), int(q[1]), fac.ctypes.data_as(fp), summ.ctypes.data_as(fp), out.ctypes.data_as(fp))
        assert np.array_equal(out, expected)
    for q, expected in zip(functions["normal_q"], functions["normal_out"]):
        L.oracle_decode_normal(int(q[0]), int(q[1]), out.ctypes.data_as(fp))
        assert np.array_equal(out, expected)


def test_brdf(functions):
    import ctypes as C
    L = oracle.lib()
    fp = C.POINTER(C.c_float)
    out = np.zeros(3, np.float32)
    for sd, wi, expected in zip(functions["brdf_sd"], functions["brdf_wi"], functions["brdf_out"]):
        sd, wi = sd.copy(), wi.copy()
        for j, (dif, spec) in enumerate([(1, 1), (1, 0), (0, 1), (0, 0)]):
            L.oracle_evaluate_brdf(sd.ctypes.data_as(fp), wi.ctypes.data_as(fp), dif, spec, out.ctypes.data_as(fp))
            assert np.array_equal(out.view(np.uint32), expected[j].view(np.uint32))


def test_srgb_transfer(functions):
    lin = functions["srgb_in"]
    rgba = np.zeros((len(lin), 4), np.float32)
    rgba[:, 0] = lin
    got = oracle.encode_srgb8(rgba)[:, 0]
    expected = (np.clip(functions["srgb_out"][:, 0], 0, 1) * 255.0 + 0.5).astype(np.uint8)
    assert np.array_equal(got, expected)
