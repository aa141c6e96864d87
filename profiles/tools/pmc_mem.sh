# Memory-pipeline counters of the trace kernel (config 3).  Run on the GPU box:
#   gpurun -- 'bash profiles/tools/pmc_mem.sh'
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
B="python $R/bench.py --config ${CFG:-3} --steps 4 --warmup 2 --no-cpu-baseline"
F="--kernel-include-regex ${KERNEL:-trace_shadow_rays} --output-format csv"
i=0
for set in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
	i=$((i+1))
	timeout 75 rocprofv3 --kernel-trace $F --pmc $set -d /tmp/pm$i -o pmc -- $B > /tmp/pm$i.log 2>&1 || { echo "set $i failed or timed out: $set"; tail -2 /tmp/pm$i.log; }
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/pm*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, n), v in per.items():
        acc[n].append(v)
for n, v in sorted(acc.items()):
    print("%-42s %16.0f  (mean of %d launches)" % (n, sum(v) / len(v), len(v)))
PY
