set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r07c; mkdir -p $O; export TMPDIR=/tmp
cd $R
ls -la vulkan_renderer_amd/*.so > $O/library_sizes.txt
profiles/tools/lds_granule.bin > $O/lds_granule.txt 2>&1
# correctness of everything that changed (the whole GPU suite)
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
AB="python profiles/tools/ab_knobs.py"
# the large scene: rays handed to idle lanes, resting shaft pairs
$AB --scene large --config 3 --steps 100 --rounds 2 --set VKR_WIDE_REFILL=0,VKR_SHAFT_REST=0 --set VKR_WIDE_REFILL=16,VKR_SHAFT_REST=0 --set VKR_WIDE_REFILL=0 --set VKR_WIDE_REFILL=4 --set VKR_WIDE_REFILL=8 --set VKR_WIDE_REFILL=16 --set VKR_WIDE_REFILL=24 --set VKR_WIDE_REFILL=32 --set VKR_WIDE_REFILL=48 --set VKR_WIDE_REFILL=16,VKR_LIGHT_SHAFTS=0 --set VKR_WIDE_REFILL=16,VKR_LEAF_BATCH=8 --set VKR_WIDE_REFILL=16,VKR_LEAF_BATCH=32 > $O/ab_large.jsonl 2>&1
# the benchmark scene
$AB --config 3 --steps 400 --rounds 3 --set VKR_WIDE_REFILL=0 --set VKR_WIDE_REFILL=16 --set VKR_WIDE_REFILL=32 --set VKR_WIDE_REFILL=16,VKR_SHAFT_REST=0 --set fif=4 --set fif=6 > $O/ab_config3.jsonl 2>&1
$AB --config 4 --steps 16 --rounds 2 --set VKR_WIDE_REFILL=0 --set VKR_WIDE_REFILL=16 --set VKR_WIDE_REFILL=32 > $O/ab_config4.jsonl 2>&1
$AB --config 2 --steps 1000 --rounds 3 --set VKR_WIDE_REFILL=0 --set VKR_WIDE_REFILL=16 --set fif=4 --set fif=6 > $O/ab_config2.jsonl 2>&1
$AB --config target --steps 600 --rounds 3 --set VKR_WIDE_REFILL=0 --set VKR_WIDE_REFILL=16 --set fif=4 --set fif=6 --set fif=6,VKR_SHAFT_MAX_STEPS=20 > $O/ab_target.jsonl 2>&1
# a rank's slab at N = 8 (and 4): pipeline depth, step limit of the shaft walks
$AB --config 3 --ranks 8 --steps 400 --rounds 3 --set fif=3 --set fif=4 --set fif=6 --set fif=8 --set fif=6,VKR_SHAFT_MAX_STEPS=20 --set fif=6,VKR_SHAFT_MAX_STEPS=10 --set fif=8,VKR_SHAFT_MAX_STEPS=20 --set fif=6,VKR_LIGHT_SHAFTS=0 > $O/ab_slab_c3_n8.jsonl 2>&1
$AB --config 3 --ranks 4 --steps 400 --rounds 2 --set fif=3 --set fif=4 --set fif=6 > $O/ab_slab_c3_n4.jsonl 2>&1
$AB --config target --ranks 8 --steps 600 --rounds 3 --set fif=3 --set fif=6 --set fif=8 --set fif=8,VKR_SHAFT_MAX_STEPS=20 --set fif=8,VKR_LIGHT_SHAFTS=0 > $O/ab_slab_target_n8.jsonl 2>&1
$AB --config 4 --ranks 8 --steps 60 --rounds 2 --set fif=3 --set fif=4 --set fif=6 > $O/ab_slab_c4_n8.jsonl 2>&1
# the line as the driver will see it
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.log 2> $O/bench_driver.err; cp gpurun_out/bench_details.json $O/bench_driver_details.json
tail -n 1 $O/bench_driver.log | cut -c 1-1500
grep -h setting $O/ab_*.jsonl | cut -c 1-220
