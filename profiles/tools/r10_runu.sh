# trace_shadow_rays_wide: a ray is first tested against the triangle that blocked the lane's last blocked ray (libvkr_shading.so) against the kernel without that (libvkr_nocache.so), in turn
O=gpurun_out/r10u; mkdir -p $O
Q="--no-extra --no-secondary --no-other-modes --no-host-frames --no-live-pmc"
for round in 1 2; do for lib in shading pix; do for w in "3 bench" "3 large" "target bench" "2 bench" "4 bench"; do set -- $w
S="--steps 200 --warmup 20"; [ $1 = 4 ] && S="--steps 20 --warmup 4"; [ $2 = large ] && S="--steps 60 --warmup 10"; [ $1 = 2 ] && S="--steps 1000 --warmup 100"
VKR_SHADING_LIBRARY=vulkan_renderer_amd/libvkr_$lib.so python bench.py --config $1 --scene $2 $Q $S --details $O/t.json > $O/t.log 2>&1
python - <<PY
import json
d=json.load(open("$O/t.json"))
print(json.dumps({"library": "$lib", "config": "$1", "scene": "$2", "round": $round, "ms_per_step": d["ms_per_step"], "kernel_ms_alone": d["roofline"]["kernel_ms"], "pass_alone_ms": d["roofline"]["pass_alone_ms"], "pixels_differing": (d.get("parity") or {}).get("pixels_differing"), "rays": d.get("shadow_rays_per_frame")}))
PY
done; done; done | tee $O/blocker_cache_by_pixel.jsonl
