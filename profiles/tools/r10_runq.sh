# XCD-local tiles (VKR_XCD_TILES=1, launch_block_of_workgroup): every XCD works on one tile at a time.  Config 3 on both scenes, the
# target shape and config 4: frame period (three in flight), the pass alone, the shading kernel alone; parity bits from the bench's own check
O=gpurun_out/r10q; mkdir -p $O
Q="--no-extra --no-secondary --no-other-modes --no-host-frames --no-live-pmc"
for round in 1 2; do for mode in "16 0" "64 0" "64 1" "128 0" "128 1" "256 1"; do set -- $mode; for w in "3 bench" "3 large" "target bench" "4 bench"; do set -- $1 $2 $w
S="--steps 200 --warmup 20"; [ $3 = 4 ] && S="--steps 20 --warmup 4"; [ $4 = large ] && S="--steps 60 --warmup 10"
VKR_BENCH_TILE=$1 VKR_XCD_TILES=$2 python bench.py --config $3 --scene $4 $Q $S --details $O/t.json > $O/t.log 2>&1
python - <<PY
import json
d=json.load(open("$O/t.json"))
print(json.dumps({"tile": $1, "xcd_tiles": $2, "config": "$3", "scene": "$4", "round": $round, "ms_per_step": d["ms_per_step"], "kernel_ms_alone": d["roofline"]["kernel_ms"], "pass_alone_ms": d["roofline"]["pass_alone_ms"], "pixels_differing": (d.get("parity") or {}).get("pixels_differing"), "rays": d.get("shadow_rays_per_frame")}))
PY
done; done; done | tee $O/xcd_tiles.jsonl
