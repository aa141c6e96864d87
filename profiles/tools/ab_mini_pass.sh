#!/bin/bash
# mini library with shading_pass.o rebuilt with extra flags: profiles/tools/ab_mini_pass.sh <tag> "<flags>"
set -e
TAG=$1; EXTRA=${2:-}
cd "$(dirname "$0")/../../vulkan_renderer_amd/csrc"
mkdir -p build/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-result -I../../include -I. -I/opt/rocm/include -fno-slp-vectorize -ffp-contract=off -DVKR_MATH_MODE=0 $EXTRA -c shading_pass.hip -o build/ab/shading_pass_$TAG.o
[ -f build/ab/stubs_shade_libm_3.o ] || ../../profiles/tools/ab_mini.sh base shade_libm_3 "" > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvkr_mini_$TAG.so build/host_*.o build/ab/shading_pass_$TAG.o build/lbvh_build.o build/shade_libm_3.o build/ab/stubs_shade_libm_3.o -lm -ldl
ls -la ../libvkr_mini_$TAG.so
