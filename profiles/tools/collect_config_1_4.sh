export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01f_c14; mkdir -p $O; cd /tmp
for CFG in 1 4; do
	timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg${CFG}_trace -o trace -- python $R/bench.py --config $CFG --no-cpu-baseline > $O/cfg${CFG}_trace.log 2>&1
	timeout 300 python $R/bench.py --config $CFG > $O/cfg${CFG}_bench.json 2>/dev/null
done
