#!/bin/bash
# Round 4, second GPU call: light shafts on (whole GPU suite + A/B of the bench line), the fixed IEEE check library,
# the self-launched two-rank run again, kernels alone under rocprofv3, VALU price list with whole-kernel timing.
set -u
TAG=${1:-r05b}
R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=12 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee $O/pytest_rc.txt
tail -8 $O/pytest_gpu.log
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
VKR_LIGHT_SHAFTS=0 timeout 300 python bench.py --no-cpu-baseline --no-other-modes > $O/bench_no_shafts.json 2> $O/bench_no_shafts.err; echo "bench without shafts rc $?"
python - <<PY
import json
for name in ("bench_default", "bench_no_shafts"):
    d = json.loads([l for l in open("$O/%s.json" % name) if l.startswith("{")][-1])
    print(name, d["value"], d["ms_per_step"], d["latency_ms"], d["shadow_rays_per_frame"], d["light_shafts"]["clear_fraction"], "config 4:", d["secondary"]["ms_per_step"], d["secondary"]["shadow_rays_per_frame"],
          {k: (v["ms_per_step"], v["shadow_rays_per_frame"]) for k, v in d.get("extra_workloads", {}).items()})
PY
VKR_BENCH_DEVICE=0 VKR_BENCH_BACKEND=gloo timeout 240 python bench.py --gpus 2 --steps 40 --warmup 5 --no-secondary --no-cpu-baseline --no-extra > $O/bench_two_ranks_one_gpu.json 2> $O/bench_two_ranks_one_gpu.err; echo "two ranks rc $?"
grep '^{' $O/bench_two_ranks_one_gpu.json | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['value'], d['ms_per_step'], d.get('stages'), d.get('scaling_parity'))"
timeout 120 profiles/tools/valu_rate2.bin > $O/valu_rate2.txt 2>&1; echo "valu rc $?"
cd /tmp; export TMPDIR=/tmp
for CFG in 3 4; do
	timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg${CFG}_serial -o trace -- python $R/bench.py --config $CFG --no-cpu-baseline --no-secondary --no-other-modes --no-extra --frames-in-flight 1 --steps 60 --warmup 10 > $O/cfg${CFG}_serial.log 2>&1
	echo "rocprof config $CFG rc $?"
	python - <<PY
import csv, glob
for path in glob.glob("$O/cfg${CFG}_serial/**/*kernel_stats.csv", recursive=True):
    for row in list(csv.DictReader(open(path)))[:6]:
        print(row["Name"][:70], row["Calls"], row["AverageNs"], row["Percentage"])
PY
done
