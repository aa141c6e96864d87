#!/bin/bash
# A/B on the GPU box: bench.py with each alternative library (see ab_build.sh), config 3 by default.
#   gpurun --timeout 600 -- 'bash profiles/tools/ab_run.sh r02c "shading b3 b4"'
set -u
TAG=$1; LIBS=$2; ARGS=${3:---no-secondary --no-cpu-baseline}
O=gpurun_out/$TAG; mkdir -p $O
for L in $LIBS; do
	VKR_SHADING_LIBRARY=$PWD/vulkan_renderer_amd/libvkr_$L.so timeout 300 python bench.py $ARGS > $O/ab_$L.json 2> $O/ab_$L.err
	python - <<P
import json
try:
    d = json.loads([l for l in open("$O/ab_$L.json") if l.startswith("{")][-1])
    print("$L", "ms/step", d["ms_per_step"], "shade alone", d["roofline"]["kernel_ms"], "pass alone", d["roofline"]["pass_alone_ms"], "value", d["value"])
except Exception as e:
    print("$L FAILED", e)
P
done
