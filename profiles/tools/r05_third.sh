#!/bin/bash
# Round 4, third full GPU call: whole GPU suite with the final light shafts, the default bench line, child order A/B on
# both scenes, the VALU price list with whole-kernel timing, instruction-cache counters of the shading kernels.
set -u
TAG=${1:-r05i}
R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee $O/pytest_rc.txt
tail -6 $O/pytest_gpu.log
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
for ORDER in 1 0; do
	VKR_WIDE_CHILD_ORDER=$ORDER timeout 200 python bench.py --scene large --no-extra --no-secondary --no-cpu-baseline --no-other-modes --steps 100 --warmup 10 > $O/large_order$ORDER.json 2> $O/large_order$ORDER.err
	VKR_WIDE_CHILD_ORDER=$ORDER VKR_LIGHT_SHAFTS=0 timeout 200 python bench.py --no-extra --no-secondary --no-cpu-baseline --no-other-modes > $O/bench_order${ORDER}_no_shafts.json 2> $O/bench_order${ORDER}_no_shafts.err
done
python - <<PY
import json
def line(name):
    return json.loads([l for l in open("$O/%s.json" % name) if l.startswith("{")][-1])
d = line("bench_default")
print("default", d["value"], d["ms_per_step"], d["latency_ms"], d["shadow_rays_per_frame"], d["light_shafts"]["clear_fraction"], "config 4:", d["secondary"]["ms_per_step"], d["secondary"]["shadow_rays_per_frame"],
      {k: (v["ms_per_step"], v["shadow_rays_per_frame"], v["parity"]["pixels_differing_in_bits"]) for k, v in d.get("extra_workloads", {}).items()}, "parity", d["parity"]["vs_libm_oracle"]["pixels_differing_in_bits"])
for name in ("large_order1", "large_order0", "bench_order1_no_shafts", "bench_order0_no_shafts"):
    d = line(name)
    w = d["traversal"]["wide"]
    print(name, d["ms_per_step"], d["shadow_rays_per_frame"], "fetches/ray", w["fetches_per_ray"], "blocked", w["fetches_per_blocked_ray"], "visible", w["fetches_per_visible_ray"], "triangle tests", w["triangle_tests_per_ray"])
PY
timeout 120 profiles/tools/valu_rate2.bin > $O/valu_rate2.txt 2>&1; echo "valu rc $?"
cd /tmp; export TMPDIR=/tmp
rocprofv3-avail list 2>/dev/null | grep -i -E "icache|ifetch|inst_cache|SQC_" | head -40 > $O/counters_icache.txt
FILTER="--kernel-include-regex shade_pixels|trace_shadow_rays|resolve_shadow|light_shafts --output-format csv"
for CFG in 3 4; do
	B="python $R/bench.py --config $CFG --steps 6 --warmup 2 --prewarm-frames 8 --no-cpu-baseline --no-secondary --no-other-modes --no-extra"
	timeout 150 rocprofv3 --kernel-trace $FILTER --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES -d $O/cfg${CFG}_icache -o pmc -- $B > $O/cfg${CFG}_icache.log 2>&1; echo "icache pmc config $CFG rc $?"
	timeout 150 rocprofv3 --kernel-trace $FILTER --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_WAVE_CYCLES -d $O/cfg${CFG}_ifetch -o pmc -- $B > $O/cfg${CFG}_ifetch.log 2>&1; echo "ifetch pmc config $CFG rc $?"
done
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/cfg*_i*")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            acc[row["Kernel_Name"].split("(")[0][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for kernel, counters in acc.items():
        print(d.split("/")[-1], kernel, {c: round(sum(v) / len(v)) for c, v in counters.items()})
PY
