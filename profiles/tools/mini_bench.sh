#!/bin/bash
# Quick A/B on the GPU box with the small builds of profiles/tools/ab_mini.sh (strategy 3, libm only):
#   gpurun --timeout 600 -- 'bash profiles/tools/mini_bench.sh r05c s1 base'
# prints value / ms per step / rays / shaft statistics / parity of config 3 and config 4 for every tag.
set -u
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
for LIB in "$@"; do
	for CFG in ${CFGS:-3 4}; do
		EXTRA="--no-secondary --no-extra --no-other-modes"
		[ $CFG = 4 ] && EXTRA="$EXTRA --no-cpu-baseline --steps 40 --warmup 5"
		VKR_SHADING_LIBRARY=$(pwd)/vulkan_renderer_amd/libvkr_mini_$LIB.so timeout 300 python bench.py --config $CFG $EXTRA > $O/${LIB}_cfg$CFG.json 2> $O/${LIB}_cfg$CFG.err
		python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/${LIB}_cfg$CFG.json") if l.startswith("{")][-1])
    s = d["light_shafts"]
    print("$LIB config $CFG: %.1f Msamples/s, %.4f ms/step, alone %.4f ms (shade %.4f, shafts %s), rays %d, clear %.3f of %d pairs, not clear %s, work %s, parity %s" % (d["value"], d["ms_per_step"], d["latency_ms"], d["roofline"]["kernel_ms"], d["roofline"].get("light_shaft_kernel_ms"), d["shadow_rays_per_frame"], s["clear_fraction"], s["patch_light_pairs"], s["not_clear"], s.get("work"), (d.get("parity") or {}).get("vs_libm_oracle", {}).get("pixels_differing_in_bits")))
except Exception as error:
    print("$LIB config $CFG failed:", error)
PY
	done
done
