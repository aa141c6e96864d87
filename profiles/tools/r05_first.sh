#!/bin/bash
# Round 4, first GPU call: the whole GPU test-suite (new: large scene, division window, full-size primary visibility,
# exchange abort paths), the default bench line, a self-launched two-rank run on the one GPU, the VALU price list
# measured again, the whole config-4 frame against the oracle.
#   gpurun --timeout 2400 -- 'bash profiles/tools/r05_first.sh r05a'
set -u
TAG=${1:-r05a}
O=gpurun_out/$TAG; mkdir -p $O
timeout 1300 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee $O/pytest_rc.txt
tail -5 $O/pytest_gpu.log
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
tail -c 600 $O/bench_default.json | head -c 300; echo
VKR_BENCH_DEVICE=0 VKR_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 40 --warmup 5 --no-secondary --no-cpu-baseline > $O/bench_two_ranks_one_gpu.json 2> $O/bench_two_ranks_one_gpu.err; echo "two ranks rc $?"
grep '^{' $O/bench_two_ranks_one_gpu.json | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['value'], d['ms_per_step'], d.get('stages'), d.get('scaling_parity'))"
timeout 120 profiles/tools/valu_rate2.bin > $O/valu_rate2.txt 2>&1; echo "valu rc $?"
timeout 700 python profiles/tools/config4_whole_frame.py $TAG > $O/config4_whole_frame.log 2>&1; echo "config 4 rc $?"
tail -1 $O/config4_whole_frame.log | head -c 600; echo
