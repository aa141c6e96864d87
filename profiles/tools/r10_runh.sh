set -x
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r10n; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"
tail -5 $O/pytest_gpu.log
python bench.py --steps 20 --warmup 5 > $O/bench_driver.log 2>&1; cp gpurun_out/bench_details.json $O/bench_driver_details.json
tail -c 2500 $O/bench_driver.log
