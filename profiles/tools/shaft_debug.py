#!/usr/bin/env python3
"""Finds the pixel(s) of a sweep case whose frame differs with light shafts on / off and explains them: the verdicts of the
pixel's patch, and for every light that was called clear the rays (to 400 points of the light) that a brute-force test over
all triangles finds blocked, with the blocking triangle.   python profiles/tools/shaft_debug.py <seed>"""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_cases  # noqa: E402
import oracle  # noqa: E402
import test_gpu_sweep  # noqa: E402
from vulkan_renderer_amd import renderer, synthetic  # noqa: E402


def render(case, dataset, shafts, W, H):
    os.environ["VKR_LIGHT_SHAFTS"] = str(shafts)
    r = renderer.Renderer()
    golden_cases.apply_case(r, case, dataset, W, H)
    cam = dict(synthetic.DEFAULT_CAMERA, **case["camera"])
    r.set_camera(cam["position"], cam["rotation_x"], cam["rotation_z"], cam["vertical_fov"], cam["near"], cam["far"])
    r.set_settings(roughness_factor=case["roughness_factor"], exposure_factor=case["exposure_factor"], mis_visibility_estimate=case["mis_visibility_estimate"])
    r.create_targets()
    r.create_pass()
    r.render_visibility()
    r.render()
    return r, r.read_radiance()


def main():
    seed = int(sys.argv[1])
    W, H = 160, 96
    case = test_gpu_sweep.random_case(seed)
    with tempfile.TemporaryDirectory() as d:
        dataset = synthetic.write_dataset(d, **golden_cases.DATASET)
        r0, off = render(case, dataset, 0, W, H)
        r0.close()
        r, on = render(case, dataset, 1, W, H)
        stats = r.light_shaft_statistics()
        L = stats["lights"]
        words = np.zeros(stats["pairs"], np.uint32)
        r.lib.read_back_light_shafts(C.byref(r.app), words.ctypes.data, words.size)
        verdict = words.reshape(-1, L)
        visibility = r.read_visibility()
        inputs = r.host_inputs(visibility)
        settings = r.oracle_settings()
        spec = r.app.scene_specification
        lights = [np.array([[spec.polygonal_lights[i].vertices_world_space[4 * j + k] for k in range(3)] for j in range(spec.polygonal_lights[i].vertex_count)]) for i in range(L)]
        r.close()
    differing = np.argwhere((on.view(np.uint32) != off.view(np.uint32)).any(axis=-1))
    print("pixels that differ:", differing.tolist(), "on", [on[y, x].tolist() for y, x in differing], "off", [off[y, x].tolist() for y, x in differing])
    bvh = oracle.Bvh(inputs["quantized_positions"], inputs["dequantization_factor"], inputs["dequantization_summand"])
    frame = oracle.make_frame(inputs, settings, bvh)
    q = inputs["quantized_positions"].reshape(-1, 3, 2)
    fx = (q[..., 0] & 0x1FFFFF).astype(np.float64)
    fy = (((q[..., 0] & 0xFFE00000) >> 21) | ((q[..., 1] & 0x3FF) << 11)).astype(np.float64)
    fz = ((q[..., 1] & 0x7FFFFC00) >> 10).astype(np.float64)
    triangles = np.stack([fx, fy, fz], -1) * inputs["dequantization_factor"].astype(np.float64) + inputs["dequantization_summand"].astype(np.float64)
    blocks_x = (W + 15) // 16
    data = np.zeros(17, np.float32)
    fp = C.POINTER(C.c_float)
    for y, x in differing:
        block = (y // 16) * blocks_x + x // 16
        wave = ((y % 16) // 8) * 2 + (x % 16) // 8
        b = ((block >> 3) << 5) | (wave << 3) | (block & 7)
        print("pixel", (x, y), "patch", b, "verdicts", [hex(int(v)) for v in verdict[b]])
        origins = []
        for yy in range(y // 8 * 8, y // 8 * 8 + 8):
            for xx in range(x // 8 * 8, x // 8 * 8 + 8):
                if yy < H and xx < W and visibility[yy, xx] != 0xFFFFFFFF:
                    oracle.lib().oracle_shading_data(C.byref(frame), int(xx), int(yy), data.ctypes.data_as(fp))
                    origins.append((xx, yy, data[0:3].astype(np.float64).copy(), data[3:6].astype(np.float64).copy(), int(visibility[yy, xx])))
        print("patch origins: %d, box %s .. %s, primitives %s" % (len(origins), np.min([o[2] for o in origins], 0).round(4).tolist(), np.max([o[2] for o in origins], 0).round(4).tolist(), sorted(set(o[4] for o in origins))))
        rng = np.random.default_rng(3)
        for i, v in enumerate(lights):
            if (int(verdict[b, i]) & 0xFF) != 1:
                continue
            print(" light", i, "called clear; vertices", v.round(3).tolist())
            found = 0
            for (xx, yy, p, n, primitive) in origins:
                for _ in range(400):
                    w = rng.dirichlet(np.ones(len(v)))
                    target = (w[:, None] * v).sum(0)
                    dvec = target - p
                    dist = np.linalg.norm(dvec)
                    dvec /= dist
                    # brute force over all triangles (double precision Moeller-Trumbore)
                    e1, e2 = triangles[:, 1] - triangles[:, 0], triangles[:, 2] - triangles[:, 0]
                    pv = np.cross(dvec, e2)
                    det = (e1 * pv).sum(-1)
                    ok = np.abs(det) > 1e-20
                    s = p - triangles[:, 0]
                    u = (s * pv).sum(-1) / np.where(ok, det, 1)
                    qv = np.cross(s, e1)
                    vv = (dvec * qv).sum(-1) / np.where(ok, det, 1)
                    t = (e2 * qv).sum(-1) / np.where(ok, det, 1)
                    hit = ok & (u >= 0) & (vv >= 0) & (u + vv <= 1) & (t >= 1e-3) & (t <= dist)
                    if hit.any() and found < 6:
                        k = int(np.argmax(hit))
                        print("   ray from pixel", (xx, yy), "origin", p.round(4).tolist(), "normal", n.round(3).tolist(), "n.d %.4f" % float(n @ dvec), "to", target.round(3).tolist(), "blocked by triangle", k, triangles[k].round(4).tolist(), "t %.5f of %.4f" % (t[k], dist))
                        found += 1
            if not found:
                print("   no blocked ray found among the samples")


if __name__ == "__main__":
    main()
