#!/bin/bash
# occluder lists (light_shafts.h) where the shafts used to lose: the large scene, the target shape - shafts forced on / off
# with the small build named by $1 (profiles/tools/ab_mini.sh)
set -u
LIB=$1; O=gpurun_out/${2:-r05zh}; mkdir -p $O
export VKR_SHADING_LIBRARY=$(pwd)/vulkan_renderer_amd/libvkr_mini_$LIB.so
for CASE in "large:--config 3 --scene large" "target:--config target"; do
	NAME=${CASE%%:*}; ARGS=${CASE#*:}
	for SHAFTS in 0 1; do
		VKR_LIGHT_SHAFTS=$SHAFTS timeout 300 python bench.py $ARGS --no-secondary --no-extra --no-other-modes --no-cpu-baseline > $O/${NAME}_$SHAFTS.json 2> $O/${NAME}_$SHAFTS.err
		python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/${NAME}_$SHAFTS.json") if l.startswith("{")][-1])
    print("$NAME shafts $SHAFTS: %.4f ms/step, alone %.4f, shade %.4f, shafts %s, rays %d, parity %s" % (d["ms_per_step"], d["latency_ms"], d["roofline"]["kernel_ms"], d["roofline"].get("light_shaft_kernel_ms"), d["shadow_rays_per_frame"], (d.get("parity") or {}).get("vs_libm_oracle", {}).get("pixels_differing_in_bits")))
except Exception as error:
    print("$NAME $SHAFTS failed:", error)
PY
	done
done
