export VKR_SHADING_LIBRARY=$(pwd)/vulkan_renderer_amd/libvkr_mini_cur.so
mkdir -p gpurun_out/r05q
for S in 1 0; do
  VKR_LIGHT_SHAFTS=$S timeout 200 python bench.py --config target --no-extra --no-secondary --no-other-modes --no-cpu-baseline > gpurun_out/r05q/target_$S.json 2>/dev/null
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r05q/target_$S.json") if l.startswith("{")][-1])
print("target shafts $S:", d["ms_per_step"], d["latency_ms"], d["shadow_rays_per_frame"], d["roofline"].get("light_shaft_kernel_ms"), d["light_shafts"]["clear_fraction"])
PY
done
