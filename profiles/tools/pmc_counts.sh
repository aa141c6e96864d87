# VALU/SALU instruction counts of the shading kernel per arithmetic mode (no rays).
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for mode in exact fast; do
	B="python $R/bench.py --config ${CFG:-3} --steps 3 --warmup 1 --no-cpu-baseline --no-rays --mode $mode"
	timeout 90 rocprofv3 --kernel-trace --kernel-include-regex shade_pixels --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD -d /tmp/pc_$mode -o pmc -- $B > /tmp/pc_$mode.log 2>&1 || echo "failed $mode"
	python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/pc_$mode/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, n), v in per.items():
        acc[n].append(v)
print("$mode", {n: int(sum(v) / len(v)) for n, v in sorted(acc.items())})
PY
done
