# config 3 (V = 5) and config 4 (V = 7) with the second polygon table in device memory from V = 5 on and with four waves per
# SIMD asked of the register allocator (128 VGPRs, the rest in scratch memory) against the product kernels, in turn
O=gpurun_out/r10m; mkdir -p $O
Q="--no-extra --no-secondary --no-other-modes --no-cpu-baseline --no-live-pmc --no-host-frames"
for round in 1 2; do for t in base mem5w4 mem5w5; do for c in 3 target; do
S="--steps 200 --warmup 20"; true
VKR_SHADING_LIBRARY=vulkan_renderer_amd/libvkr_mini_$t.so python bench.py --config $c $Q $S --details $O/c${c}_${t}_$round.json > $O/c${c}_${t}_$round.log 2>&1
python - <<PY
import json
d=json.load(open("$O/c${c}_${t}_$round.json"))
print(json.dumps({"library": "$t", "config": "$c", "round": $round, "ms_per_step": d["ms_per_step"], "kernel_ms_alone": d["roofline"]["kernel_ms"], "pass_alone_ms": d["roofline"]["pass_alone_ms"]}))
PY
done; done; done | tee $O/waves5.jsonl
