# Instruction mix of the shading kernel (no rays): FMA / MUL / ADD / TRANS / INT / CVT
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for mode in ${MODES:-exact fast}; do
	B="python $R/bench.py --config ${CFG:-3} --steps 3 --warmup 1 --no-cpu-baseline --no-rays --mode $mode"
	timeout 90 rocprofv3 --kernel-trace --kernel-include-regex shade_pixels --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU -d /tmp/pm_$mode -o pmc -- $B > /tmp/pm_$mode.log 2>&1 || echo "failed $mode"
	python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/pm_$mode/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, n), v in per.items():
        acc[n].append(v)
print("$mode", {n.replace("SQ_INSTS_", ""): round(sum(v) / len(v) / 1e6, 1) for n, v in sorted(acc.items())}, "(millions of wave instructions per launch)")
PY
done
