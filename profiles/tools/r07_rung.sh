set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r07g; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_light_shafts.py tests/test_gpu_golden.py tests/test_gpu_experiments.py -m gpu -x -q > $O/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_subset.log
tail -3 $O/pytest_subset.log
python profiles/tools/predict_scaling.py --configs 3 target 4 --tiles 32 --out $O/predicted_scaling > $O/predict.log 2>&1
cat $O/predicted_scaling.md
