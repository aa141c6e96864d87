O=gpurun_out/r05z; mkdir -p $O
for F in 2 3 4; do
VKR_SHADING_LIBRARY=$(pwd)/vulkan_renderer_amd/libvkr_mini_base.so timeout 200 python bench.py --config 3 --no-secondary --no-extra --no-other-modes --no-cpu-baseline --frames-in-flight $F > $O/fif$F.json 2> $O/fif$F.err
python - <<PY
import json
d = json.loads([l for l in open("$O/fif$F.json") if l.startswith("{")][-1])
print("fif $F", d["value"], d["ms_per_step"], d["latency_ms"])
PY
done
