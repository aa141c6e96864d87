O=gpurun_out/r06h; mkdir -p $O
export VKR_SHADING_LIBRARY=$(pwd)/vulkan_renderer_amd/libvkr_mini_cur.so
for W in 0 1 2 4 0; do
VKR_TRACE_WAVES=$W timeout 200 python bench.py --config 3 --no-secondary --no-extra --no-other-modes --no-cpu-baseline > $O/tw$W.json 2> $O/tw$W.err
python - <<PY
import json
d = json.loads([l for l in open("$O/tw$W.json") if l.startswith("{")][-1])
print("trace waves $W", d["value"], d["ms_per_step"], d["latency_ms"])
PY
done
