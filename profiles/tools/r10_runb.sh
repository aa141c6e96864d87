set -x
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r10b; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"
tail -5 $O/pytest_gpu.log
python bench.py --steps 20 --warmup 5 > $O/bench_driver.log 2>&1; cp gpurun_out/bench_details.json $O/bench_driver_details.json
tail -c 1500 $O/bench_driver.log
Q="--no-extra --no-secondary --no-cpu-baseline --no-other-modes"
python bench.py $Q --details $O/bench_500_details.json > $O/bench_500.log 2>&1; tail -c 600 $O/bench_500.log
python bench.py $Q --steps 200 --warmup 20 --force-distributed --details $O/bench_fd_ondemand_details.json > $O/bench_fd_ondemand.log 2>&1; tail -c 600 $O/bench_fd_ondemand.log
python bench.py $Q --steps 200 --warmup 20 --force-distributed --assemble every-frame --details $O/bench_fd_everyframe_details.json > $O/bench_fd_everyframe.log 2>&1; tail -c 600 $O/bench_fd_everyframe.log
python bench.py $Q --steps 200 --warmup 20 --force-distributed --exchange none --details $O/bench_fd_none_details.json > $O/bench_fd_none.log 2>&1; tail -c 600 $O/bench_fd_none.log
