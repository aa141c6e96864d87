#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box:
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash profiles/collect.sh r01'
# Kernel-trace statistics and PMC counters are taken in SEPARATE runs (counters
# perturb timing; the guide's HBM section asks for separate --pmc passes).
# Raw output lands in gpurun_out/<tag>/; profiles/summarize.py turns it into the
# committed summaries.
set -u
TAG=${1:-r01}
# arithmetic mode of the kernels (bench.py --mode): libm is the default of the pass
MODE=${2:-libm}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
FILTER="--kernel-include-regex shade_pixels|trace_shadow_rays|resolve_shadow|light_shafts --output-format csv"
for CFG in 2 3; do
	# the bench command itself (default steps / warm-up, two frames in flight) ...
	B="python $R/bench.py --config $CFG --mode $MODE --no-cpu-baseline --no-secondary --no-other-modes --no-extra"
	timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg${CFG}_trace -o trace -- $B > $O/cfg${CFG}_trace.log 2>&1
	# ... and with one frame at a time: every kernel alone on the GPU
	timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg${CFG}_serial -o trace -- $B --frames-in-flight 1 --steps 200 --warmup 50 > $O/cfg${CFG}_serial.log 2>&1
	B="python $R/bench.py --config $CFG --mode $MODE --steps 6 --warmup 2 --prewarm-frames 8 --no-cpu-baseline --no-secondary --no-other-modes --no-extra"
	timeout 150 rocprofv3 --kernel-trace $FILTER --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU -d $O/cfg${CFG}_pmc1 -o pmc -- $B > $O/cfg${CFG}_pmc1.log 2>&1
	timeout 150 rocprofv3 --kernel-trace $FILTER --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_FLAT SQ_INSTS_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE -d $O/cfg${CFG}_pmc2 -o pmc -- $B > $O/cfg${CFG}_pmc2.log 2>&1
	timeout 150 rocprofv3 --kernel-trace $FILTER --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $O/cfg${CFG}_pmc3 -o pmc -- $B > $O/cfg${CFG}_pmc3.log 2>&1
	timeout 150 rocprofv3 --kernel-trace $FILTER --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/cfg${CFG}_pmc4 -o pmc -- $B > $O/cfg${CFG}_pmc4.log 2>&1
	timeout 150 rocprofv3 --kernel-trace $FILTER --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS -d $O/cfg${CFG}_pmc5 -o pmc -- $B > $O/cfg${CFG}_pmc5.log 2>&1
	# instruction classes: the VALU issue cost differs by class (profiles/tools/valu_rate.hip)
	timeout 150 rocprofv3 --kernel-trace $FILTER --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 -d $O/cfg${CFG}_pmc6 -o pmc -- $B > $O/cfg${CFG}_pmc6.log 2>&1
	timeout 150 rocprofv3 --kernel-trace $FILTER --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 SQ_INSTS_SMEM -d $O/cfg${CFG}_pmc7 -o pmc -- $B > $O/cfg${CFG}_pmc7.log 2>&1
	# the bench line itself, un-profiled, for the record
	timeout 300 python $R/bench.py --config $CFG --mode $MODE --no-secondary > $O/cfg${CFG}_bench.json 2> $O/cfg${CFG}_bench.err
done
find $O -name "*.csv" | wc -l
