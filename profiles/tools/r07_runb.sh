set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r07b; mkdir -p $O; export TMPDIR=/tmp
cd $R
# 1. one rank's slab of config 3 at N = 8: plain, then under a kernel trace, three and four frames in flight; tile 16
for FIF in 3 4; do
  python profiles/tools/slab_probe.py --config 3 --ranks 8 --tile 32 --frames-in-flight $FIF --whole > $O/slab_c3_n8_t32_fif$FIF.json 2>&1
done
python profiles/tools/slab_probe.py --config 3 --ranks 8 --tile 16 --frames-in-flight 3 > $O/slab_c3_n8_t16_fif3.json 2>&1
python profiles/tools/slab_probe.py --config target --ranks 8 --tile 32 --frames-in-flight 3 --whole > $O/slab_target_n8_t32_fif3.json 2>&1
VKR_LIGHT_SHAFTS=0 python profiles/tools/slab_probe.py --config 3 --ranks 8 --tile 32 --frames-in-flight 3 --whole > $O/slab_c3_n8_t32_fif3_noshafts.json 2>&1
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/slab_trace -o trace -- python $R/profiles/tools/slab_probe.py --config 3 --ranks 8 --tile 32 --frames-in-flight 3 --steps 100 > $O/slab_trace.log 2>&1)
python profiles/tools/timeline_overlap.py $O/slab_trace 2.0 48 > $O/slab_timeline.txt 2>&1
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/whole_trace -o trace -- python $R/profiles/tools/slab_probe.py --config 3 --ranks 1 --tile 16 --frames-in-flight 3 --steps 100 > $O/whole_trace.log 2>&1)
python profiles/tools/timeline_overlap.py $O/whole_trace 6.0 24 > $O/whole_timeline.txt 2>&1
find $O -name "*.csv" -size +4M -delete
# 2. large scene: counters of the tracing kernel
SCENE=large CFG=3 OUT=gpurun_out/r07b/large KERNEL="trace_shadow_rays" bash profiles/tools/pmc_scene.sh > $O/large_pmc.log 2>&1
cat $O/slab_*.json; cat $O/slab_timeline.txt | head -30
