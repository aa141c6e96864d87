set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r07f; mkdir -p $O; export TMPDIR=/tmp
cd $R
AB="python profiles/tools/ab_knobs.py"
SETS="--set fif=4 --set fif=4,VKR_SHAFT_MAX_STEPS=20 --set fif=4,VKR_SHAFT_MAX_STEPS=12 --set fif=4,VKR_SHAFT_MAX_STEPS=8 --set fif=4,VKR_TRACE_WAVES=2 --set fif=4,VKR_TRACE_WAVES=4 --set fif=4,VKR_SHAFT_MAX_STEPS=12,VKR_TRACE_WAVES=2 --set fif=4,VKR_TRACE_SINGLE_WAVES=0"
$AB --config 3 --ranks 8 --rank 3 --steps 400 --rounds 3 $SETS > $O/ab_slab_c3_n8_rank3.jsonl 2>&1
$AB --config 3 --ranks 8 --rank 4 --steps 400 --rounds 3 --set fif=4 --set fif=4,VKR_SHAFT_MAX_STEPS=12 --set fif=4,VKR_SHAFT_MAX_STEPS=12,VKR_TRACE_WAVES=2 > $O/ab_slab_c3_n8_rank4.jsonl 2>&1
$AB --config 3 --ranks 4 --rank 2 --steps 400 --rounds 2 --set fif=3 --set fif=3,VKR_SHAFT_MAX_STEPS=12 --set fif=3,VKR_TRACE_WAVES=4 > $O/ab_slab_c3_n4.jsonl 2>&1
$AB --config target --ranks 8 --rank 3 --steps 600 --rounds 3 --set fif=3 --set fif=4 --set fif=3,VKR_SHAFT_MAX_STEPS=12 --set fif=3,VKR_TRACE_WAVES=2 --set fif=3,VKR_SHAFT_MAX_STEPS=12,VKR_TRACE_WAVES=2 > $O/ab_slab_target_n8.jsonl 2>&1
$AB --config 3 --steps 400 --rounds 2 --set fif=3 --set fif=3,VKR_SHAFT_MAX_STEPS=20 --set fif=3,VKR_SHAFT_MAX_STEPS=12 > $O/ab_config3_steps.jsonl 2>&1
grep -h setting $O/ab_*.jsonl | cut -c 1-200
