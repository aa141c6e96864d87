# Front-end counters of the shading kernel.  gpurun -- 'CFG=3 bash profiles/tools/pmc_shade.sh'
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
B="python $R/bench.py --config ${CFG:-3} --steps 4 --warmup 2 --no-cpu-baseline ${EXTRA:-}"
F="--kernel-include-regex ${KERNEL:-shade_pixels} --output-format csv"
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_WAVES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_IFETCH_LEVEL SQ_LEVEL_WAVES SQ_THREAD_CYCLES_VALU"; do
	i=$((i+1))
	timeout 75 rocprofv3 --kernel-trace $F --pmc $set -d /tmp/ps$i -o pmc -- $B > /tmp/ps$i.log 2>&1 || { echo "set $i failed or timed out: $set"; tail -2 /tmp/ps$i.log; }
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/ps*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, n), v in per.items():
        acc[n].append(v)
for n, v in sorted(acc.items()):
    print("%-42s %16.0f  (mean of %d launches)" % (n, sum(v) / len(v), len(v)))
PY
