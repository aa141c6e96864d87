# how the V = 7 shading kernel depends on its resident waves: config 4 with 9 (product), 8, 7, 6 waves
# per CU, by asking for LDS that is not used (VKR_EXPERIMENT_EXTRA_LDS); frames unchanged
O=gpurun_out/r10f; mkdir -p $O
Q="--config 4 --no-extra --no-secondary --no-other-modes --no-cpu-baseline --steps 20 --warmup 4"
for round in 1 2; do for extra in 0 1280 3840 8960; do
VKR_EXPERIMENT_EXTRA_LDS=$extra VKR_SHADING_LIBRARY=vulkan_renderer_amd/libvkr_mini_base.so python bench.py $Q --details $O/c4_lds_${extra}_$round.json > $O/c4_lds_${extra}_$round.log 2>&1
python - <<PY
import json
d=json.load(open("$O/c4_lds_${extra}_$round.json"))
print(json.dumps({"extra_lds": $extra, "round": $round, "ms_per_step": d["ms_per_step"], "kernel_ms_alone": d["roofline"]["kernel_ms"], "pass_alone_ms": d["roofline"]["pass_alone_ms"]}))
PY
done; done | tee $O/waves_per_cu.jsonl
