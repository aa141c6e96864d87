# occupancy proxy for VERDICT round 5 item 4: config 4 (V = 7) with both polygon tables aliased into one (wrong frames, the
# LDS and resident waves of a one-table-at-a-time scheme) against the product kernel, same box, in turn
O=gpurun_out/r10f; mkdir -p $O
Q="--config 4 --no-extra --no-secondary --no-other-modes --no-cpu-baseline --steps 20 --warmup 4"
for round in 1 2; do for t in base alias; do
VKR_SHADING_LIBRARY=vulkan_renderer_amd/libvkr_mini_$t.so python bench.py $Q --details $O/c4_${t}_$round.json > $O/c4_${t}_$round.log 2>&1
python - <<PY
import json
d=json.load(open("$O/c4_${t}_$round.json"))
print("$t", $round, "ms_per_step", d["ms_per_step"], "kernel_ms alone", d["roofline"]["kernel_ms"], "pass alone", d["roofline"]["pass_alone_ms"])
PY
done; done
