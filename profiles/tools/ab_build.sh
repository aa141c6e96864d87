#!/bin/bash
# Builds vulkan_renderer_amd/libvkr_<tag>.so: the library with ONE translation unit recompiled with
# extra flags, for A/B measurements on the GPU box (VKR_SHADING_LIBRARY=... python bench.py ...).
#   profiles/tools/ab_build.sh b3 shade_exact_3 "-DVKR_SHADE_BOUNDS=256,3"
set -e
TAG=$1; UNIT=$2; EXTRA=$3
cd "$(dirname "$0")/../../vulkan_renderer_amd/csrc"
make -s -j 8 all > /dev/null
SRC=shading_variants.hip; DEFS=""
case $UNIT in
	shade_libm_*) DEFS="-ffp-contract=off -DVKR_MATH_MODE=2 -DVKR_STRATEGY=${UNIT##*_}";;
	shade_exact_*) DEFS="-ffp-contract=off -DVKR_MATH_MODE=0 -DVKR_STRATEGY=${UNIT##*_}";;
	shade_fast_*) DEFS="-ffp-contract=fast -DVKR_MATH_MODE=1 -DVKR_STRATEGY=${UNIT##*_}";;
	shading_pass) SRC=shading_pass.hip; DEFS="-ffp-contract=off -DVKR_MATH_MODE=0";;
	*) echo "unknown unit $UNIT"; exit 1;;
esac
mkdir -p build/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-result --offload-compress -I../../include -I. -I/opt/rocm/include -fno-slp-vectorize $DEFS $EXTRA -c $SRC -o build/ab/${UNIT}_$TAG.o
OBJ=$(ls build/*.o | grep -v "build/$UNIT.o" | grep -v "build/ieee_")
/opt/rocm/bin/hipcc --offload-arch=gfx950 --offload-compress -shared -fPIC -o ../libvkr_$TAG.so $OBJ build/ab/${UNIT}_$TAG.o -lm -ldl
echo built vulkan_renderer_amd/libvkr_$TAG.so
