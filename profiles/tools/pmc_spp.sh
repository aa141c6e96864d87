# VALU instructions of the shading kernel as a function of the sample count (config 3, no rays):
# instructions per pixel = fixed + lights * (prepare + spp * per_sample_pair)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for spp in 1 2 4 8; do
	B="python $R/bench.py --config ${CFG:-3} --steps 3 --warmup 1 --no-cpu-baseline --no-rays --spp $spp"
	timeout 90 rocprofv3 --kernel-trace --kernel-include-regex shade_pixels --output-format csv --pmc SQ_INSTS_VALU SQ_WAVES -d /tmp/ps_$spp -o pmc -- $B > /tmp/ps_$spp.log 2>&1 || echo "failed $spp"
	python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/ps_$spp/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, n), v in per.items():
        acc[n].append(v)
valu = sum(acc["SQ_INSTS_VALU"]) / len(acc["SQ_INSTS_VALU"]); waves = sum(acc["SQ_WAVES"]) / len(acc["SQ_WAVES"])
print("spp $spp: %.0f VALU instructions per wave (one pixel per lane)" % (valu / waves))
PY
done
