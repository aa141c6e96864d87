#!/bin/bash
# Round 4, fourth full GPU call: everything with the kernels as they stand - the whole GPU suite (both libraries), the
# default bench line, the rocprofv3 collection for profiles/, the predicted scaling with the current kernels.
set -u
TAG=${1:-r05m}
R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee $O/pytest_rc.txt
tail -6 $O/pytest_gpu.log
timeout 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
python - <<PY
import json
d = json.loads([l for l in open("$O/bench_default.json") if l.startswith("{")][-1])
print("default", d["value"], d["ms_per_step"], d["latency_ms"], d["shadow_rays_per_frame"], d["light_shafts"]["clear_fraction"], d["roofline"]["kernel_ms"], d["roofline"].get("light_shaft_kernel_ms"), "config 4:", d["secondary"]["ms_per_step"], d["secondary"]["shadow_rays_per_frame"],
      {k: (v["ms_per_step"], v["shadow_rays_per_frame"], v["parity"]["pixels_differing_in_bits"]) for k, v in d.get("extra_workloads", {}).items()}, "parity", d["parity"]["vs_libm_oracle"]["pixels_differing_in_bits"])
PY
bash profiles/collect.sh $TAG > $O/collect.log 2>&1; echo "collect rc $?"
cd $R
timeout 600 python profiles/tools/predict_scaling.py --tiles 32 --configs 4 --out gpurun_out/$TAG/predicted_scaling > $O/predict.log 2>&1; echo "predict rc $?"; tail -12 $O/predict.log
