set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r07j; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 1400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log; grep FAILED $O/pytest_gpu.log | head
python bench.py > $O/bench_default.log 2> $O/bench_default.err; cp gpurun_out/bench_details.json $O/bench_default_details.json
tail -n 1 $O/bench_default.log | wc -c
python -c "
import json; d=json.load(open('$O/bench_default_details.json')); print(json.dumps(d.get('timing_matrix_cells'))[:700])"
python profiles/tools/whole_frame_parity.py --config 3 --scene large --out $O/large_scene_whole_frame.json > $O/large_whole.log 2>&1; tail -1 $O/large_whole.log | cut -c 1-500
python profiles/tools/whole_frame_parity.py --config target --scene bench --out $O/target_whole_frame.json > $O/target_whole.log 2>&1; tail -1 $O/target_whole.log | cut -c 1-300
