import sys, tempfile, json
sys.path.insert(0, '/root/repo')
from vulkan_renderer_amd import renderer, synthetic
with tempfile.TemporaryDirectory() as tmp:
    dataset = synthetic.write_dataset(tmp, grid=256, box_count=64, seed=1234, ltc_resolution=16, fresnel_count=8)
    out = {}
    for builder in ("sah_device", "sah_device", "lbvh_device"):
        r = renderer.Renderer()
        renderer.setup_config(r, 3, dataset, width=256, height=144, acceleration_structure=builder)
        out.setdefault(builder, []).append(round(float(r.app.scene.acceleration_structure.build_milliseconds), 3))
        r.close()
    print(json.dumps(out))
