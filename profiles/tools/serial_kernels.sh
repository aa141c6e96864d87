set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/serial
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for MODE in exact fast; do
for CFG in 2 3; do
	STEPS=40; [ $CFG = 3 ] && STEPS=10
	timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${MODE}_cfg${CFG} -o t -- python $R/bench.py --config $CFG --steps $STEPS --warmup 3 --no-cpu-baseline --frames-in-flight 1 --mode $MODE > $O/${MODE}_cfg${CFG}.log 2>&1
	echo "== $MODE cfg$CFG"
	f=$(find $O/${MODE}_cfg${CFG} -name "*kernel_stats.csv" | head -1)
	head -5 "$f" | cut -d, -f1-6 | cut -c1-150
done
done
