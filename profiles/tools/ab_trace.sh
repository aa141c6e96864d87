export TMPDIR=/tmp; cd /tmp; R=$GRAFT_REPO_ROOT
for q in 0 1; do for t in 0 24 44 56; do
export VKR_QUEUE_MODE=$q VKR_REFILL_THRESHOLD=$t
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_${q}_$t -o t -- python $R/bench.py --config 3 --steps 8 --warmup 2 --no-cpu-baseline > /tmp/ab.log 2>&1
python - <<PY
import csv,glob
rows=list(csv.DictReader(open(glob.glob("/tmp/ab_${q}_$t/*kernel_stats.csv")[0])))
d={r["Name"].split("(")[0][-22:]: float(r["AverageNs"])/1e6 for r in rows[:4]}
print("queue_mode=$q threshold=$t", {k: round(v,3) for k,v in d.items() if "shade" in k or "trace" in k or "resolve" in k})
PY
done; done
