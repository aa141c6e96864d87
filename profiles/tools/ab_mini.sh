#!/bin/bash
# A small build of the library for A/B measurements on the GPU box: host code, shading_pass, the BVH builder and ONE
# shading translation unit compiled with extra flags; every other launcher is a stub that reports "not built".
# 8 MB instead of 120 (what is pushed to the GPU box is paid for in GPU time).
#   profiles/tools/ab_mini.sh <tag> shade_libm_3 "-mllvm -some-flag"     -> vulkan_renderer_amd/libvkr_mini_<tag>.so
#   VKR_SHADING_LIBRARY=vulkan_renderer_amd/libvkr_mini_<tag>.so python bench.py --config 3 --no-extra --no-secondary --no-other-modes ...
set -e
TAG=$1; UNIT=$2; EXTRA=${3:-}; OPT=${4:--O3}
cd "$(dirname "$0")/../../vulkan_renderer_amd/csrc"
SRC=shading_variants.hip
case $UNIT in
	shade_libm_*) DEFS="-ffp-contract=off -DVKR_MATH_MODE=2 -DVKR_STRATEGY=${UNIT##*_}";;
	shade_exact_*) DEFS="-ffp-contract=off -DVKR_MATH_MODE=0 -DVKR_STRATEGY=${UNIT##*_}";;
	*) echo "unknown unit $UNIT"; exit 1;;
esac
mkdir -p build/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 $OPT -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-result -I../../include -I. -I/opt/rocm/include -fno-slp-vectorize $DEFS $EXTRA -c $SRC -o build/ab/${UNIT}_$TAG.o
python3 - > build/ab/stubs_$UNIT.c <<PY
unit = "$UNIT"
names = ["vkr_launch_shade_%s%s_%d" % (t, m, s) for t in ("", "textured_") for m in ("libm", "fast", "exact") for s in range(5)]
names = [n for n in names if n != "vkr_launch_" + unit]
print("/* launchers that this small build does not contain */")
for n in names:
    print("int %s(int technique, int capacity, int rays, const void* p, unsigned int grid_x, void* stream) { (void) technique; (void) capacity; (void) rays; (void) p; (void) grid_x; (void) stream; return -1; }" % n)
for m in ("libm", "fast", "exact"):
    print("int vkr_launch_error_display_%s(int a, int b, int c, int d, const void* p, unsigned int g, void* s) { (void) a; (void) b; (void) c; (void) d; (void) p; (void) g; (void) s; return -1; }" % m)
    print("int vkr_launch_resolve_materials_%s(const void* p, float* m, void* s) { (void) p; (void) m; (void) s; return -1; }" % m)
PY
gcc -std=gnu99 -O1 -fPIC -fvisibility=hidden -c build/ab/stubs_$UNIT.c -o build/ab/stubs_$UNIT.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvkr_mini_$TAG.so build/host_*.o build/shading_pass.o build/lbvh_build.o build/ab/${UNIT}_$TAG.o build/ab/stubs_$UNIT.o -lm -ldl
ls -la ../libvkr_mini_$TAG.so
