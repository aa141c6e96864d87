"""Size-independent properties of the sampling code (SURVEY.md 8c substitutes for
the golden vectors the reference does not ship), checked on the CPU oracle."""
import tempfile

import numpy as np
import pytest

import golden_cases
import oracle


def lambert_form_factor(polygon):
    """Projected solid angle of a polygon above the horizon (Lambert's formula, float64)."""
    d = polygon.astype(np.float64)
    d /= np.linalg.norm(d, axis=1)[:, None]
    total = 0.0
    n = len(d)
    for i in range(n):
        a, b = d[i], d[(i + 1) % n]
        c = np.cross(a, b)
        norm = np.linalg.norm(c)
        total += np.arctan2(norm, np.dot(a, b)) * c[2] / norm
    return 0.5 * total


def clipped_cases(count, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < count:
        n = int(rng.integers(3, 8))
        poly = golden_cases.random_polygon(rng, n)
        for p in (poly, poly[::-1].copy()):
            c, buf = oracle.clip_polygon(p, max_count=n + 1)
            if c == 0:
                continue
            ref = lambert_form_factor(buf[:c])
            if ref > 1e-3:
                out.append((n, c, buf, ref))
                break
    return out


def test_projected_solid_angle_equals_lambert_form_factor():
    central = decentral = 0
    for n, c, buf, ref in clipped_cases(800, 11):
        state, cap = oracle.psa_prepare(buf, count=c, max_count=n + 1)
        assert abs(state[48] - ref) <= 2e-4 * ref + 2e-6
        assert 0.0 < state[48] <= np.pi * (1 + 1e-6)
        if state[37] > 0:
            central += 1
        else:
            decentral += 1
    assert central > 20 and decentral > 200


def test_wrong_winding_has_zero_projected_solid_angle():
    rng = np.random.default_rng(5)
    zeros = 0
    for _ in range(100):
        n = int(rng.integers(3, 8))
        poly = golden_cases.random_polygon(rng, n)
        c, buf = oracle.clip_polygon(poly, max_count=n + 1)
        if c == 0 or abs(lambert_form_factor(buf[:c])) < 1e-3:
            continue
        if lambert_form_factor(buf[:c]) < 0:
            state, _ = oracle.psa_prepare(buf, count=c, max_count=n + 1)
            assert state[48] == 0.0
            zeros += 1
    assert zeros > 10


def test_sampling_backward_error_and_bounds():
    """compute_projected_solid_angle_polygon_sampling_error: two iterations give
    practically no bias (reference polygon_sampling.glsl:642-643, acceptable error 1e-5 :705)."""
    rng = np.random.default_rng(17)
    errors = []
    for n, c, buf, ref in clipped_cases(300, 13):
        state, cap = oracle.psa_prepare(buf, count=c, max_count=n + 1)
        for _ in range(6):
            u0, u1 = rng.uniform(0, 1, 2)
            d = oracle.psa_sample(state, cap, u0, u1)
            assert abs(np.linalg.norm(d) - 1.0) < 2e-6 and d[2] >= 0.0
            if state[37] <= 0:
                e = oracle.psa_error(state, cap, u0, u1, d)
                if np.isfinite(e[0]):
                    errors.append(abs(e[0]))
    errors = np.array(errors)
    assert len(errors) > 1000
    assert np.median(errors) < 1e-6 and np.percentile(errors, 99) < 5e-5


def inside_polygon(direction, polygon):
    """Is the direction inside the spherical polygon (clockwise winding seen from the origin)?"""
    n = len(polygon)
    for i in range(n):
        if np.dot(direction.astype(np.float64), np.cross(polygon[i].astype(np.float64), polygon[(i + 1) % n].astype(np.float64))) < -1e-5:
            return False
    return True


def test_samples_are_uniform_in_projected_solid_angle():
    """Chi-square style check: equal-probability strata in u map to equal projected
    solid angle.  Splitting u0 at 0.5 must split the disk area in half."""
    rng = np.random.default_rng(23)
    for n, c, buf, ref in clipped_cases(12, 29):
        state, cap = oracle.psa_prepare(buf, count=c, max_count=n + 1)
        us = rng.uniform(0, 1, (4000, 2))
        dirs = np.array([oracle.psa_sample(state, cap, a, b) for a, b in us])
        inside = np.array([inside_polygon(d, buf[:c]) for d in dirs])
        assert inside.mean() > 0.995
        # E[z / density] = PSA for the density z / PSA is trivial; instead check the first moment of
        # the sample positions against numerical integration over the projected polygon
        xy = dirs[:, :2].astype(np.float64)
        # Monte Carlo reference by rejection in the unit disk
        pts = rng.uniform(-1, 1, (200000, 2))
        r2 = (pts ** 2).sum(axis=1)
        pts = pts[r2 < 1]
        z = np.sqrt(1 - (pts ** 2).sum(axis=1))
        d3 = np.concatenate([pts, z[:, None]], axis=1)
        mask = np.ones(len(d3), bool)
        for i in range(c):
            mask &= d3 @ np.cross(buf[i].astype(np.float64), buf[(i + 1) % c].astype(np.float64)) >= 0
        if mask.sum() < 2000:
            continue
        centroid = pts[mask].mean(axis=0)
        spread = pts[mask].std(axis=0).max()
        assert np.all(np.abs(xy.mean(axis=0) - centroid) < 6 * spread / np.sqrt(len(xy)) + 6 * spread / np.sqrt(mask.sum()))


def test_clip_polygon_invariants():
    rng = np.random.default_rng(31)
    for _ in range(500):
        n = int(rng.integers(3, 8))
        poly = golden_cases.random_polygon(rng, n)
        c, buf = oracle.clip_polygon(poly, max_count=n + 1)
        assert c == 0 or 3 <= c <= n + 1
        if c:
            assert np.all(buf[:c, 2] >= 0.0)
            if c < n + 1:
                assert np.array_equal(buf[c], buf[0])
            # clipping keeps the projected solid angle of the upper part (checked via Lambert on both)
            above = (poly[:, 2] > 0).sum()
            if above == n:
                assert np.array_equal(buf[:n], poly)


def test_solid_angle_of_octant_and_sampling_inside():
    octant = np.array([[1, 0, 0], [0, 0, 1], [0, 1, 0]], np.float32)  # clockwise seen from the origin
    sa, d = oracle.solid_angle_sample(octant, [0, 0, 0], 0.3, 0.6, max_count=3)
    assert abs(sa - np.pi / 2) < 2e-6
    rng = np.random.default_rng(41)
    for _ in range(200):
        u = rng.uniform(0, 1, 2)
        sa, d = oracle.solid_angle_sample(octant, [0, 0, 0], u[0], u[1], max_count=3)
        assert abs(np.linalg.norm(d) - 1) < 1e-5 and np.all(d > -1e-5)


def test_deterministic_math_is_accurate():
    """The functions of math mode 1 - the polynomial arctangent shared with the kernels' "exact" mode, the C
    library for the rest - against float64."""
    L = oracle.lib()
    oracle.set_math_mode(1)
    try:
        rng = np.random.default_rng(43)
        xs = np.concatenate([rng.normal(size=20000) * 2, rng.normal(size=5000) * 100, 10.0 ** rng.uniform(-20, 20, 5000)]).astype(np.float32)
        got = np.array([L.oracle_atan(float(x)) for x in xs], np.float32)
        ref = np.arctan(xs.astype(np.float64))
        ulp = np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)
        assert np.max(np.abs(got - ref) / ulp) <= 2.0
        xs = rng.uniform(0, 1, 20000).astype(np.float32)
        got = np.array([L.oracle_acos_unit(float(x)) for x in xs], np.float32)
        ref = np.arccos(xs.astype(np.float64))
        ulp = np.spacing(np.maximum(np.abs(ref), 1e-3).astype(np.float32)).astype(np.float64)
        assert np.max(np.abs(got - ref) / ulp) <= 3.0
        import ctypes as C
        s, c = C.c_float(), C.c_float()
        worst = 0.0
        for x in rng.uniform(-20, 20, 20000).astype(np.float32):
            L.oracle_sincos(float(x), C.byref(s), C.byref(c))
            worst = max(worst, abs(s.value - np.sin(np.float64(x))), abs(c.value - np.cos(np.float64(x))))
        assert worst < 2.5e-7
        # inversesqrt: 1 / sqrt, two correctly rounded operations (until round 3: integer seed + Newton, 0.85 ulp)
        xs = np.concatenate([rng.uniform(1e-3, 4.0, 20000), np.exp(rng.uniform(-80, 80, 20000))]).astype(np.float32)
        got = np.array([L.oracle_rsqrt(float(x)) for x in xs], np.float32).astype(np.float64)
        ref = 1.0 / np.sqrt(xs.astype(np.float64))
        ulp = np.spacing(ref.astype(np.float32)).astype(np.float64)
        assert np.max(np.abs(got - ref) / ulp) <= 1.5
        # log2 of the range the error display feeds it (1 .. 1e5)
        xs = np.exp(rng.uniform(0.0, np.log(1.0e5), 20000)).astype(np.float32)
        got = np.array([L.oracle_log2(float(x)) for x in xs], np.float64)
        assert np.max(np.abs(got - np.log2(xs.astype(np.float64)))) <= 2.5e-6
    finally:
        oracle.set_math_mode(0)
    # fast_positive_atan: documented maximal error 1.16e-5 (polygon_sampling.glsl:79-82)
    xs = np.concatenate([np.linspace(-50, 50, 20001), 10.0 ** np.linspace(-6, 6, 2000)]).astype(np.float32)
    got = np.array([L.oracle_fast_positive_atan(float(x)) for x in xs])
    ref = np.arctan(xs.astype(np.float64)) + np.where(xs < 0, np.pi, 0.0)
    assert np.max(np.abs(got - ref)) < 1.3e-5


def test_kahan_determinant_error_bound():
    L = oracle.lib()
    rng = np.random.default_rng(47)
    for _ in range(3000):
        a, b = rng.normal(size=2).astype(np.float32)
        c, d = (np.float32(a * (1 + rng.normal() * 1e-5)), np.float32(b * (1 + rng.normal() * 1e-5)))
        got = L.oracle_kahan(float(a), float(b), float(c), float(d))
        ref = np.float64(a) * np.float64(b) - np.float64(c) * np.float64(d)
        assert abs(got - ref) <= 1.5 * np.spacing(np.float32(abs(ref))) + 1e-45


def test_bvh_any_hit_equals_brute_force(dataset):
    from vulkan_renderer_amd import renderer
    hs = renderer.HostScene()
    hs.load_scene(dataset["scene"], dataset["textures"])
    inputs_q = np.ctypeslib.as_array(hs.app.scene.mesh.host_positions, (hs.app.scene.mesh.triangle_count * 3, 2)).copy()
    fac = np.array(hs.app.scene.mesh.dequantization_factor[:], np.float32)
    summ = np.array(hs.app.scene.mesh.dequantization_summand[:], np.float32)
    bvh = oracle.Bvh(inputs_q, fac, summ)
    rng = np.random.default_rng(53)
    hits = 0
    for _ in range(3000):
        o = np.array([rng.uniform(-8, 8), rng.uniform(-8, 8), rng.uniform(0.001, 2.0)], np.float32)
        d = rng.normal(size=3).astype(np.float32)
        d /= np.linalg.norm(d)
        t_max = float(rng.uniform(0.1, 20))
        a = bvh.any_hit(o, d, 1e-3, t_max)
        b = bvh.any_hit(o, d, 1e-3, t_max, brute_force=True)
        assert a == b
        hits += a
    assert 300 < hits < 2900
    hs.close()


def test_noise_stream_order(dataset):
    """get_noise_2 hands out .xy then .zw of one texel and fetches with the permuted
    frame randoms on every second call (reference noise_utility.glsl:63-103)."""
    import ctypes as C
    from vulkan_renderer_amd import renderer
    hs = renderer.HostScene()
    renderer.setup_config(hs, 1, dataset, width=32, height=32)
    inputs = hs.host_inputs(np.zeros((32, 32), np.uint32))
    frame = oracle.make_frame(inputs, hs.oracle_settings())
    out = np.zeros(16, np.float32)
    oracle.lib().oracle_noise_stream(C.byref(frame), 5, 9, 8, out.ctypes.data_as(C.POINTER(C.c_float)))
    noise = inputs["noise"]
    rnd = [0, 0x123456, 0x2468AC, 0x369D02]
    expect = []
    for s in range(4):
        r = rnd[2:] + rnd[:2] if s & 2 else list(rnd)
        if s & 1:
            r = r[1:] + [r[3]]
        shift = (s & 124) >> 2
        x = (5 + (r[0] >> shift)) & 255
        y = (9 + (r[1] >> shift)) & 255
        layer = (r[2] + s) & 63
        expect += list(noise[layer, y, x].astype(np.float32) / np.float32(65535.0))
    assert np.array_equal(out, np.array(expect, np.float32))
    hs.close()


def test_tiling_invariance_of_the_oracle(dataset):
    """Each pixel depends only on its own coordinates: shading rows separately equals one pass."""
    from vulkan_renderer_amd import renderer
    from helpers import oracle_render
    hs = renderer.HostScene()
    renderer.setup_config(hs, 3, dataset, width=48, height=32, sample_count=1)
    full, inputs, bvh = oracle_render(hs, math_mode=0)
    frame = oracle.make_frame(inputs, hs.oracle_settings(), bvh)
    parts = np.zeros_like(full)
    for y0 in range(0, 32, 5):
        part = oracle.shade(frame, y0, min(32, y0 + 5))
        parts[y0:y0 + 5] = part[y0:y0 + 5]
    assert np.array_equal(full, parts)
    hs.close()


def test_estimators_converge_to_the_same_image(dataset):
    """Unbiasedness (reference experiment_list.c:95-100,251-262): projected solid angle
    sampling with MIS and plain solid angle sampling estimate the same integral."""
    from vulkan_renderer_amd import renderer
    from helpers import oracle_render
    images = []
    for technique, strategy, spp in (("projected_solid_angle", "diffuse_specular_mis", 48), ("solid_angle", "diffuse_only", 192)):
        hs = renderer.HostScene()
        renderer.setup_config(hs, 2, dataset, width=40, height=24, sample_count=spp, polygon_technique=technique,
                              sampling_strategies=strategy, mis_heuristic="balance", trace_shadow_rays=False)
        image, _, _ = oracle_render(hs, math_mode=0)
        images.append(image[..., :3].astype(np.float64))
        hs.close()
    a, b = images
    lit = b.mean(axis=-1) > 0.02
    assert lit.sum() > 100
    rel = np.abs(a[lit] - b[lit]).mean() / b[lit].mean()
    assert rel < 0.08, rel


def test_deterministic_math_mode_renders_the_same_frames_as_libm():
    """Math mode 1 (math mode 0 with a polynomial arctangent: what the kernels' "exact" mode mirrors
    bit for bit) against math mode 0 (libm / IEEE: what is pinned against the reference and what the
    kernels' default mode mirrors bit for bit): the same frames to well within the stated tolerance."""
    from vulkan_renderer_amd import renderer, synthetic
    import golden_cases
    with tempfile.TemporaryDirectory() as d:
        dataset = synthetic.write_dataset(d, **golden_cases.DATASET)
        for case in golden_cases.FRAME_CASES:
            if case.get("error_display") or not case.get("output_linear_rgb", True):
                continue  # the error display shows rounding errors themselves (see test_gpu_golden.py)
            if "biquadratic" in case.get("technique", ""):
                # Hart's biquadratic warp evaluates inversesqrt(2 (1 - (v0 . v0)^2)) for the unit vector
                # v0 (polygon_sampling_related_work.glsl:431-435): the argument is rounding noise of
                # either sign, so one of its nine density samples - and with it every warped sample -
                # changes with the last bit of normalize().  Not comparable between arithmetic modes.
                continue
            hs, frame, _ = golden_cases.build_frame(case, dataset)
            libm = oracle.shade(frame)
            oracle.set_math_mode(1)
            try:
                deterministic = oracle.shade(frame)
            finally:
                oracle.set_math_mode(0)
            hs.close()
            difference = deterministic[..., :3].astype(np.float64) - libm[..., :3]
            # pixels that hit the shader's NaN guard in one mode only are counted, not averaged
            guard = (np.abs(difference).max(axis=-1) > 0.1)
            assert guard.sum() <= (8 if "arvo" in case.get("technique", "") else 2), case["key"]
            # Arvo's samplers are "less stable" in the reference's own words (polygon_sampling_related_work.glsl:219, :520)
            unstable = "arvo" in case.get("technique", "")
            assert np.sqrt((difference[~guard] ** 2).mean()) <= (1.0e-3 if unstable else 1.0e-4), case["key"]  # the stated tolerance
