set -x
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r10a
python -m pytest tests -m gpu -x -q > gpurun_out/r10a/pytest_gpu.log 2>&1; echo "pytest rc $?" 
tail -5 gpurun_out/r10a/pytest_gpu.log
for fif in 1 2 3 4; do python profiles/tools/frame_periods.py --config 3 --frames 240 --fif $fif; done > gpurun_out/r10a/frame_periods.jsonl 2>&1
python profiles/tools/frame_periods.py --config target --frames 240 --fif 3 >> gpurun_out/r10a/frame_periods.jsonl 2>&1
python profiles/tools/frame_periods.py --config 2 --frames 240 --fif 3 >> gpurun_out/r10a/frame_periods.jsonl 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r10a/bench_driver.log 2>&1; cp gpurun_out/bench_details.json gpurun_out/r10a/bench_driver_details.json
tail -c 3000 gpurun_out/r10a/bench_driver.log
