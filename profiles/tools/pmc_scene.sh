# Memory-pipeline and issue counters of the tracing kernel (or KERNEL=...) for one workload.  Run on the GPU box:
#   gpurun -- 'SCENE=large CFG=3 OUT=gpurun_out/r07b/large bash profiles/tools/pmc_scene.sh'
# Counters in their own passes (never combined with other trace domains), one workload per pass.
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp
OUT=$R/${OUT:-gpurun_out/pmc_scene}; mkdir -p $OUT
B="python $R/bench.py --config ${CFG:-3} --scene ${SCENE:-bench} --steps 4 --warmup 2 --prewarm-frames 8 --no-cpu-baseline --no-secondary --no-extra --no-other-modes $EXTRA"
F="--kernel-include-regex ${KERNEL:-trace_shadow_rays|shade_pixels|light_shafts|resolve_shadow} --output-format csv"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" \
           "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_EA0_RD_UNCACHED_32B_sum"; do
	i=$((i+1))
	timeout 120 rocprofv3 --kernel-trace $F --pmc $set -d $OUT/pm$i -o pmc -- $B > $OUT/pm$i.log 2>&1 || { echo "set $i failed or timed out: $set"; tail -2 $OUT/pm$i.log; }
done
python - <<PY > $OUT/counters.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pm*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        per[(r["Kernel_Name"].split("(")[0][-48:], r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (k, d, n), v in per.items():
        acc[k][n].append(v)
for k in sorted(acc):
    print(k)
    for n, v in sorted(acc[k].items()):
        print("  %-42s %18.0f  (mean of %d launches)" % (n, sum(v) / len(v), len(v)))
PY
cat $OUT/counters.txt
find $OUT -name "*.csv" -size +2M -delete
