set -u
O=gpurun_out/r05l; mkdir -p $O
export VKR_SHADING_LIBRARY=$(pwd)/vulkan_renderer_amd/libvkr_mini_split.so
for SPLIT in 1 0; do
  for SCENE in large bench; do
    VKR_BVH_SPLIT_TRIANGLES=$SPLIT timeout 300 python bench.py --scene $SCENE --config 3 --no-extra --no-secondary --no-other-modes --steps 100 --warmup 10 > $O/${SCENE}_split$SPLIT.json 2> $O/${SCENE}_split$SPLIT.err
    python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/${SCENE}_split$SPLIT.json") if l.startswith("{")][-1])
    w = d["traversal"]["wide"]
    print("$SCENE split $SPLIT: %.4f ms/step, rays %d, build %.1f ms, node bytes %d, stack need %d, fetches/ray %.2f (blocked %.2f, visible %.2f), boxes %.1f, triangle tests %.2f, beyond lds %d, parity %s" % (
        d["ms_per_step"], d["shadow_rays_per_frame"], d["setup"]["bvh_build_ms"], d["setup"]["bvh_node_bytes"], d["setup"]["bvh_stack_need"], w["fetches_per_ray"], w["fetches_per_blocked_ray"], w["fetches_per_visible_ray"], w["boxes_tested_per_ray"], w["triangle_tests_per_ray"], w["rays_beyond_lds_stack"], d["parity"]["vs_libm_oracle"]["pixels_differing_in_bits"]))
except Exception as e:
    print("$SCENE split $SPLIT failed", e); print(open("$O/${SCENE}_split$SPLIT.err").read()[-800:])
PY
  done
done
