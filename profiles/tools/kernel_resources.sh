#!/bin/bash
# Prints VGPRs / SGPRs / scratch bytes / LDS bytes of every kernel in a host object or shared
# library with embedded gfx950 code objects (no GPU needed).
#   profiles/tools/kernel_resources.sh vulkan_renderer_amd/csrc/build/shade_exact_3.o [name filter]
set -e
input=$1; filter=${2:-.}
work=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin="$work/fat.bin" "$input" "$work/copy"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input="$work/fat.bin" \
  --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output="$work/device.co"
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$work/device.co" | python3 -c '
import sys, re, subprocess
text = sys.stdin.read()
for block in text.split("- .agpr_count")[1:]:
    get = lambda key: (re.search(r"\." + key + r":\s*(\S+)", block) or [None, "?"])[1]
    name = get("name")
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        pass
    print("%-4s vgpr %-4s sgpr %-6s scratch %-6s lds  %s" % (get("vgpr_count"), get("sgpr_count"), get("private_segment_fixed_size"), get("group_segment_fixed_size"), name))
' | grep -E "$filter"
rm -rf "$work"
