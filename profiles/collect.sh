#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box:
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash profiles/collect.sh r07 libm'
# Kernel-trace statistics and PMC counters are taken in SEPARATE runs (counters perturb timing; the guide's HBM section
# asks for separate --pmc passes, and gpurun refuses --pmc together with other trace domains).
# Raw output lands in gpurun_out/<tag>/; profiles/summarize.py turns it into the committed summaries.
# Workloads (WORKLOADS="2 3 ..."): BASELINE configs 2, 3, 4, the north_star target shape and config 3 on the large scene
# ("3_large").  Configs 2 and 3 get the full set of passes; the others the passes that the bench line's roofline block
# needs (kernel durations alone, instruction classes, HBM bytes, waits).
set -u
TAG=${1:-r01}
# arithmetic mode of the kernels (bench.py --mode): libm is the default of the pass
MODE=${2:-libm}
WORKLOADS=${WORKLOADS:-"2 3 4 target 3_large"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
export VKR_BENCH_DATASET_CACHE=/tmp/vkr_bench_datasets
cd /tmp
FILTER="--kernel-include-regex shade_pixels|trace_shadow_rays|resolve_shadow|light_shafts --output-format csv"
for W in $WORKLOADS; do
	CFG=${W%%_*}
	SCENE=bench; [ "$W" != "$CFG" ] && SCENE=${W#*_}
	P=$O/cfg${W}
	QUIET="--scene $SCENE --mode $MODE --no-cpu-baseline --no-secondary --no-other-modes --no-extra --no-live-pmc --no-host-frames"
	FULL=0; [ "$W" = "2" ] || [ "$W" = "3" ] && FULL=1
	STEPS="--steps 6 --warmup 2 --prewarm-frames 8"; [ "$CFG" = "4" ] && STEPS="--steps 3 --warmup 1 --prewarm-frames 3"
	# config 4: one launch per frame like the pass that bench.py times the kernel alone with (three bands by default), so
	# that "per dispatch" is per frame
	unset VKR_BAND_COUNT; [ "$CFG" = "4" ] && export VKR_BAND_COUNT=1
	# the bench command itself (default steps / warm-up, three frames in flight) ...
	[ $FULL = 1 ] && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d ${P}_trace -o trace -- python $R/bench.py --config $CFG $QUIET > ${P}_trace.log 2>&1
	# ... and with one frame at a time: every kernel alone on the GPU
	SERIAL="--steps 200 --warmup 50"; [ "$CFG" = "4" ] && SERIAL="--steps 12 --warmup 3"; [ "$SCENE" = "large" ] && SERIAL="--steps 60 --warmup 10"
	timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d ${P}_serial -o trace -- python $R/bench.py --config $CFG $QUIET --frames-in-flight 1 $SERIAL > ${P}_serial.log 2>&1
	B="python $R/bench.py --config $CFG $QUIET $STEPS"
	timeout 150 rocprofv3 --kernel-trace $FILTER --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU -d ${P}_pmc1 -o pmc -- $B > ${P}_pmc1.log 2>&1
	timeout 150 rocprofv3 --kernel-trace $FILTER --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_FLAT SQ_INSTS_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE -d ${P}_pmc2 -o pmc -- $B > ${P}_pmc2.log 2>&1
	timeout 150 rocprofv3 --kernel-trace $FILTER --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d ${P}_pmc3 -o pmc -- $B > ${P}_pmc3.log 2>&1
	timeout 150 rocprofv3 --kernel-trace $FILTER --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d ${P}_pmc4 -o pmc -- $B > ${P}_pmc4.log 2>&1
	[ $FULL = 1 ] && timeout 150 rocprofv3 --kernel-trace $FILTER --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS -d ${P}_pmc5 -o pmc -- $B > ${P}_pmc5.log 2>&1
	# instruction classes: the VALU issue cost differs by class (profiles/tools/valu_rate.hip)
	timeout 150 rocprofv3 --kernel-trace $FILTER --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 -d ${P}_pmc6 -o pmc -- $B > ${P}_pmc6.log 2>&1
	[ $FULL = 1 ] && timeout 150 rocprofv3 --kernel-trace $FILTER --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 SQ_INSTS_SMEM -d ${P}_pmc7 -o pmc -- $B > ${P}_pmc7.log 2>&1
	# the bench line itself, un-profiled, for the record
	unset VKR_BAND_COUNT
	timeout 300 python $R/bench.py --config $CFG --scene $SCENE --mode $MODE --no-secondary --no-extra --no-other-modes --details ${P}_bench_details.json > ${P}_bench.json 2> ${P}_bench.err
done
# (raw traces are large: the summaries need the statistics and the counter files only)
find $O -name "*kernel_trace.csv" -size +3M -delete
find $O -name "*.csv" | wc -l
