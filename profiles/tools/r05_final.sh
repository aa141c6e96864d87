#!/bin/bash
# Round 4, closing GPU call: the whole GPU suite with the libraries as committed, then the default bench line.
set -u
TAG=${1:-r05u}
R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee $O/pytest_rc.txt
tail -14 $O/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log
timeout 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
python - <<PY
import json
d = json.loads([l for l in open("$O/bench_default.json") if l.startswith("{")][-1])
print("default", d["value"], d["ms_per_step"], d["latency_ms"], d["shadow_rays_per_frame"], d["light_shafts"]["clear_fraction"], d["roofline"]["kernel_ms"], d["roofline"].get("traffic"), "config 4:", d["secondary"]["ms_per_step"],
      {k: (v["ms_per_step"], v["parity"]["pixels_differing_in_bits"]) for k, v in d.get("extra_workloads", {}).items()}, "parity", d["parity"]["vs_libm_oracle"]["pixels_differing_in_bits"])
PY
# second argument "collect": the rocprofv3 passes for profiles/ as well (kernel sources changed)
if [ "${2:-}" = collect ]; then
	bash profiles/collect.sh $TAG > $O/collect.log 2>&1; echo "collect rc $?"
fi
