set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r09z; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 1400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -2 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.log 2> $O/bench_driver.err; cp gpurun_out/bench_details.json $O/bench_driver_details.json
python bench.py > $O/bench_default.log 2> $O/bench_default.err; cp gpurun_out/bench_details.json $O/bench_default_details.json
tail -n 1 $O/bench_default.log | cut -c 1-2700
