#!/usr/bin/env python3
"""Where are light shafts clear, and where could they be?  Renders config 3 of the benchmark scene at 1920x1080, reads the
verdict of every (8x8 patch, light) back and compares it with what the oracle's ray tracer says about the same patches: a
patch is "truly clear" for a light when none of 25 rays from each of 4 of its pixels to points spread over the light is
blocked (an upper bound on what a conservative test can find).  Writes gpurun_out/<tag>/shaft_map.npz and prints the table.
  VKR_SHADING_LIBRARY=... python profiles/tools/shaft_map.py r05d"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C  # noqa: E402

import oracle  # noqa: E402
from vulkan_renderer_amd import renderer, synthetic  # noqa: E402


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "shaft_map"
    config = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    out_dir = os.path.join(ROOT, "gpurun_out", tag)
    os.makedirs(out_dir, exist_ok=True)
    W, H = 1920, 1080
    with tempfile.TemporaryDirectory() as d:
        dataset = synthetic.write_dataset(d, grid=256, box_count=64, seed=1234, ltc_resolution=16, fresnel_count=8)
        r = renderer.Renderer()
        renderer.setup_config(r, config, dataset, width=W, height=H, acceleration_structure="sah_device")
        r.create_targets()
        r.create_pass()
        r.render_visibility()
        r.render()
        stats = r.light_shaft_statistics()
        L = stats["lights"]
        words = np.zeros(stats["pairs"], np.uint32)
        got = r.lib.read_back_light_shafts(C.byref(r.app), words.ctypes.data, words.size)
        assert got == words.size
        visibility = r.read_visibility()
        inputs = r.host_inputs(visibility)
        settings = r.oracle_settings()
        spec = r.app.scene_specification
        lights = []
        for i in range(spec.polygonal_light_count):
            light = spec.polygonal_lights[i]
            lights.append(np.array([[light.vertices_world_space[4 * j + k] for k in range(3)] for j in range(light.vertex_count)]))
        r.close()
    # patch b of the shading grid -> pixel block (shade_pixels' mapping, one band, one rank, 16-pixel tiles)
    groups = words.size // L
    verdict = words.reshape(groups, L)
    blocks_x = (W + 15) // 16
    patch_xy = np.zeros((groups, 2), np.int64)
    for b in range(groups):
        local_block = ((b >> 5) << 3) | (b & 7)
        wave = (b >> 3) & 3
        ty, tx = divmod(local_block, blocks_x)
        patch_xy[b] = (tx * 16 + (wave & 1) * 8, ty * 16 + (wave >> 1) * 8)
    bvh = oracle.Bvh(inputs["quantized_positions"], inputs["dequantization_factor"], inputs["dequantization_summand"])
    frame = oracle.make_frame(inputs, settings, bvh)
    lib = oracle.lib()
    fp = C.POINTER(C.c_float)
    data = np.zeros(17, np.float32)
    rng = np.random.default_rng(1)
    sample = rng.choice(groups, size=min(groups, 1200), replace=False)
    table = {}
    details = []
    for b in sample:
        x0, y0 = patch_xy[b]
        if y0 >= H or x0 >= W:
            continue
        pixels = [(x0 + dx, y0 + dy) for dx, dy in ((0, 0), (7, 0), (0, 7), (7, 7)) if x0 + dx < W and y0 + dy < H and visibility[y0 + dy, x0 + dx] != 0xFFFFFFFF]
        if not pixels:
            continue
        for i, v in enumerate(lights):
            centroid = v.mean(axis=0)
            points = [centroid + (1.0 - 1e-3) * (a * (v[j] - centroid) + (1 - a) * c * (v[(j + 1) % len(v)] - centroid)) for j in range(len(v)) for a in (1.0, 0.5) for c in (1.0, 0.5)] + [centroid]
            facing, blocked = False, False
            for (x, y) in pixels:
                lib.oracle_shading_data(C.byref(frame), int(x), int(y), data.ctypes.data_as(fp))
                p, n = data[0:3].astype(np.float64), data[3:6].astype(np.float64)
                for q in points:
                    dvec = q - p
                    dist = np.linalg.norm(dvec)
                    dvec /= dist
                    if np.dot(dvec, n) <= 0:
                        continue
                    facing = True
                    if bvh.any_hit(p, dvec, 1e-3, dist):
                        blocked = True
                        break
                if blocked:
                    break
            kind = "back-facing" if not facing else ("blocked" if blocked else "truly clear")
            key = (kind, int(verdict[b, i]) & 0xFF)
            table[key] = table.get(key, 0) + 1
            if kind == "truly clear" and (int(verdict[b, i]) & 0xFF) == 20:
                # the triangle that the kernel found in the way, relative to the patch
                primitive = int(verdict[b, i]) >> 8
                q = inputs["quantized_positions"][3 * primitive:3 * primitive + 3]
                fx = (q[:, 0] & 0x1FFFFF).astype(np.float64)
                fy = (((q[:, 0] & 0xFFE00000) >> 21) | ((q[:, 1] & 0x3FF) << 11)).astype(np.float64)
                fz = ((q[:, 1] & 0x7FFFFC00) >> 10).astype(np.float64)
                tri = np.stack([fx, fy, fz], -1) * inputs["dequantization_factor"].astype(np.float64) + inputs["dequantization_summand"].astype(np.float64)
                origins = []
                for (x, y) in pixels:
                    lib.oracle_shading_data(C.byref(frame), int(x), int(y), data.ctypes.data_as(fp))
                    origins.append(data[0:3].astype(np.float64))
                origins = np.array(origins)
                normal = np.cross(tri[1] - tri[0], tri[2] - tri[0])
                normal /= max(np.linalg.norm(normal), 1e-30)
                heights_origins = (origins - tri[0]) @ normal
                heights_light = (v - tri[0]) @ normal
                details.append({"patch": (int(x0), int(y0)), "light": i, "triangle_centre": tri.mean(axis=0).round(3).tolist(), "triangle_normal": normal.round(3).tolist(),
                                "origin_heights": heights_origins.round(6).tolist(), "light_heights": heights_light.round(3).tolist(),
                                "origin_centre": origins.mean(axis=0).round(3).tolist(), "distance": float(np.linalg.norm(tri.mean(axis=0) - origins.mean(axis=0)).round(3))})
    names = {1: "clear", 16: "no pixel", 17: "no shaft", 18: "walk too long", 19: "queue full", 20: "triangle"}
    print("sampled %d patches x %d lights" % (len(sample), L))
    for key in sorted(table):
        print("%-12s kernel says %-14s %6d" % (key[0], names.get(key[1], key[1]), table[key]))
    for record in details[:40]:
        print(record)
    np.savez_compressed(os.path.join(out_dir, "shaft_map.npz"), verdict=verdict, patch_xy=patch_xy)


if __name__ == "__main__":
    main()
