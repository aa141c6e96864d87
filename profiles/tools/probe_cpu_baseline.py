import sys, tempfile, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import oracle
from vulkan_renderer_amd import renderer, synthetic
d = tempfile.mkdtemp()
ds = synthetic.write_dataset(d)
hs = renderer.HostScene()
renderer.setup_config(hs, 2, ds, width=1920, height=1080)
inputs = hs.host_inputs()
bvh = oracle.Bvh(inputs["quantized_positions"], inputs["dequantization_factor"], inputs["dequantization_summand"])
cam = synthetic.DEFAULT_CAMERA
inputs["visibility"] = oracle.primary_visibility(inputs["constants"], bvh, 1920, 1080, cam["near"], cam["far"])
frame = oracle.make_frame(inputs, hs.oracle_settings(), bvh)
oracle.set_math_mode(1)
times = []
for i in range(24):
    t = time.perf_counter(); oracle.shade(frame, 400, 656); times.append(time.perf_counter() - t)
    if i == 11: time.sleep(1.0)
print("per call ms:", " ".join("%.1f" % (t * 1e3) for t in times))
out = np.zeros((1080, 1920, 4), np.float32)
L = oracle.lib()
import ctypes as C
times = []
for i in range(12):
    t = time.perf_counter(); L.oracle_shade_rows(C.byref(frame), out.ctypes.data, 400, 656, 0); times.append(time.perf_counter() - t)
print("same buffer ms:", " ".join("%.1f" % (t * 1e3) for t in times), "threads", os.cpu_count())
