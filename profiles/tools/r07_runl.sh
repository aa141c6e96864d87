set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r07l; mkdir -p $O; export TMPDIR=/tmp
cd $R
AB="python profiles/tools/ab_knobs.py"
for ROUND in 1 2; do
  for LIB in shading stack6; do
    VKR_SHADING_LIBRARY=$R/vulkan_renderer_amd/libvkr_$LIB.so $AB --config 4 --steps 16 --rounds 2 --set fif=3 > $O/c4_${LIB}_$ROUND.jsonl 2>&1
  done
done
for LIB in shading stack6; do
  VKR_SHADING_LIBRARY=$R/vulkan_renderer_amd/libvkr_$LIB.so $AB --config 3 --steps 400 --rounds 2 --set fif=3 > $O/c3_${LIB}.jsonl 2>&1
  VKR_SHADING_LIBRARY=$R/vulkan_renderer_amd/libvkr_$LIB.so $AB --config 4 --ranks 8 --steps 60 --rounds 2 --set fif=4 > $O/c4n8_${LIB}.jsonl 2>&1
done
for f in $O/*.jsonl; do echo $f; grep -h setting $f | cut -c 1-170; done
