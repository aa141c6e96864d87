#!/bin/bash
# Dynamic VALU instruction mix of the kernels of the pass (per-class SQ counters), config 2 and 3.
#   gpurun --timeout 600 -- 'bash profiles/tools/pmc_valu_mix.sh r02b'
set -u
TAG=${1:-r02b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --list-avail > $O/avail.txt 2>&1
FILTER="--kernel-include-regex shade_pixels|trace_shadow_rays|resolve_shadow --output-format csv"
for CFG in ${CONFIGS:-3}; do
	B="python $R/bench.py --config $CFG --steps 6 --warmup 2 --prewarm-frames 8 --no-cpu-baseline --no-secondary ${BENCH_ARGS:-}"
	timeout 150 rocprofv3 --kernel-trace $FILTER --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 -d $O/cfg${CFG}_mix1 -o pmc -- $B > $O/cfg${CFG}_mix1.log 2>&1
	timeout 150 rocprofv3 --kernel-trace $FILTER --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F16 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_SMEM -d $O/cfg${CFG}_mix2 -o pmc -- $B > $O/cfg${CFG}_mix2.log 2>&1
	timeout 150 rocprofv3 --kernel-trace $FILTER --pmc SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $O/cfg${CFG}_mix3 -o pmc -- $B > $O/cfg${CFG}_mix3.log 2>&1
done
python3 - <<P
import csv, glob, collections
for d in sorted(glob.glob("$O/cfg*_mix*")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"].split("(")[0][-60:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        print(d.split("/")[-1], k, {c: "%.4g" % (sum(v) / len(v)) for c, v in cs.items()})
P
