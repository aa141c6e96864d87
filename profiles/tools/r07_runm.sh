set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r07m2; mkdir -p $O; export TMPDIR=/tmp
cd $R
AB="python profiles/tools/ab_knobs.py"
for ROUND in 1 2; do
  for LIB in shading vc; do
    VKR_SHADING_LIBRARY=$R/vulkan_renderer_amd/libvkr_$LIB.so $AB --config 3 --steps 600 --rounds 2 --set fif=3 --set fif=1 > $O/c3_${LIB}_$ROUND.jsonl 2>&1
    VKR_SHADING_LIBRARY=$R/vulkan_renderer_amd/libvkr_$LIB.so $AB --config 4 --steps 16 --rounds 1 --set fif=3 > $O/c4_${LIB}_$ROUND.jsonl 2>&1
    VKR_SHADING_LIBRARY=$R/vulkan_renderer_amd/libvkr_$LIB.so $AB --config target --steps 600 --rounds 2 --set fif=3 > $O/t_${LIB}_$ROUND.jsonl 2>&1
  done
done
VKR_SHADING_LIBRARY=$R/vulkan_renderer_amd/libvkr_vc.so timeout 600 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -2
for f in $O/*.jsonl; do echo $(basename $f); grep -h setting $f | cut -c 60-190; done
