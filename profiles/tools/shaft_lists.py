#!/usr/bin/env python3
"""What the shaft walks end with: per (8x8 patch, light) pair of one frame the verdict (clear / a list of n triangles /
trace, light_shafts.h) - how many pairs could be decided against a handful of triangles instead of a walk of the tree.
  VKR_SHADING_LIBRARY=... python profiles/tools/shaft_lists.py [config] [large]"""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vulkan_renderer_amd import renderer, synthetic  # noqa: E402


def main():
    config = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    large = len(sys.argv) > 2 and sys.argv[2] == "large"
    width, height = (3840, 2160) if config == 4 else (1920, 1080)
    with tempfile.TemporaryDirectory() as d:
        if large:
            dataset = synthetic.write_dataset(d, seed=4321, ltc_resolution=16, fresnel_count=8, large={})
        else:
            dataset = synthetic.write_dataset(d, grid=256, box_count=64, seed=1234, ltc_resolution=16, fresnel_count=8)
        r = renderer.Renderer()
        renderer.setup_config(r, config, dataset, width=width, height=height, acceleration_structure="sah_device")
        r.create_targets()
        r.create_pass()
        r.render_visibility()
        r.render()
        stats = r.light_shaft_statistics()
        words = np.zeros(stats["pairs"], np.uint32)
        got = r.lib.read_back_light_shafts(C.byref(r.app), words.ctypes.data, words.size)
        rays = r.last_ray_count()
        r.close()
    assert got == words.size
    code = words & 0xFF
    names = {1: "clear", 2: "list", 16: "no shaded pixel", 17: "no shaft", 18: "walk too long", 19: "queue full", 20: "triangle(s) in the way"}
    print("config", config, "large" if large else "bench", "pairs", words.size, "rays traced", rays)
    for value in sorted(set(code.tolist())):
        print("  %-24s %8d  %.3f" % (names.get(value, value), int((code == value).sum()), float((code == value).mean())))
    listed = (words[code == 2] >> 8) & 0x1F
    if listed.size:
        print("  list lengths:", {int(n): int((listed == n).sum()) for n in sorted(set(listed.tolist()))}, "mean %.2f" % listed.mean())


if __name__ == "__main__":
    main()
