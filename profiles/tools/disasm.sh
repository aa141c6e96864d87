#!/bin/bash
# Disassembly of one kernel of a host object with embedded gfx950 code:
#   profiles/tools/disasm.sh vulkan_renderer_amd/csrc/build/shading_pass.o trace_shadow_rays_wide > /tmp/wide.s
set -e
input=$1; name=$2
work=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin="$work/fat.bin" "$input" "$work/copy"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input="$work/fat.bin" --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output="$work/device.co"
/opt/rocm/lib/llvm/bin/llvm-objdump -d "$work/device.co" | awk -v n="$name" '/^[0-9a-f]+ </{p = index($0, n) > 0} p' | sed 's/ *\/\/.*//'
rm -rf "$work"
