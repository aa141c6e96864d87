#!/bin/bash
# Like ab_mini.sh, with the extra flags given to shading_pass.hip as well (knobs that the host code shares with the kernels,
# e.g. -DVKR_PSA_MEMORY_FROM=5): vulkan_renderer_amd/libvkr_mini_<tag>.so from host code, shading_pass, the BVH builder and ONE
# shading unit.
#   profiles/tools/ab_mini_pair.sh <tag> shade_libm_3 "-DVKR_PSA_MEMORY_FROM=5 -DVKR_SHADE_MIN_WAVES=4"
set -e
TAG=$1; UNIT=$2; EXTRA=${3:-}
cd "$(dirname "$0")/../../vulkan_renderer_amd/csrc"
case $UNIT in
	shade_libm_*) DEFS="-ffp-contract=off -DVKR_MATH_MODE=2 -DVKR_STRATEGY=${UNIT##*_}";;
	shade_exact_*) DEFS="-ffp-contract=off -DVKR_MATH_MODE=0 -DVKR_STRATEGY=${UNIT##*_}";;
	*) echo "unknown unit $UNIT"; exit 1;;
esac
mkdir -p build/ab
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-result -I../../include -I. -I/opt/rocm/include -fno-slp-vectorize"
/opt/rocm/bin/hipcc $FLAGS $DEFS $EXTRA -c shading_variants.hip -o build/ab/${UNIT}_$TAG.o &
/opt/rocm/bin/hipcc $FLAGS -ffp-contract=off -DVKR_MATH_MODE=0 $EXTRA -c shading_pass.hip -o build/ab/shading_pass_$TAG.o &
wait
[ -f build/ab/stubs_$UNIT.o ] || bash ../../profiles/tools/ab_mini.sh stubs_only $UNIT > /dev/null 2>&1 || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvkr_mini_$TAG.so build/host_*.o build/ab/shading_pass_$TAG.o build/lbvh_build.o build/ab/${UNIT}_$TAG.o build/ab/stubs_$UNIT.o -lm -ldl
ls -la ../libvkr_mini_$TAG.so
