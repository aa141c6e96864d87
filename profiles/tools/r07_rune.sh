set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r07e; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_large_scene.py tests/test_gpu_bvh_and_exchange.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_subset.log
tail -2 $O/pytest_subset.log
AB="python profiles/tools/ab_knobs.py"
SETS="--set VKR_WIDE_REFILL=0 --set VKR_WIDE_REFILL=16 --set VKR_WIDE_REFILL=16,VKR_WIDE_REFILL_BELOW=256 --set VKR_WIDE_REFILL=16,VKR_WIDE_REFILL_BELOW=190"
$AB --scene large --config 3 --steps 100 --rounds 2 $SETS --set VKR_WIDE_REFILL=16,VKR_WIDE_REFILL_BELOW=140 > $O/ab_large.jsonl 2>&1
$AB --config 2 --steps 1000 --rounds 3 $SETS > $O/ab_config2.jsonl 2>&1
$AB --config 3 --steps 400 --rounds 3 $SETS > $O/ab_config3.jsonl 2>&1
$AB --config 4 --steps 16 --rounds 2 $SETS > $O/ab_config4.jsonl 2>&1
$AB --config target --steps 600 --rounds 3 $SETS > $O/ab_target.jsonl 2>&1
grep -h setting $O/ab_*.jsonl | cut -c 1-200
