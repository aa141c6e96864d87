"""Constants of the roofline, the identity of the kernel sources a counter measurement belongs to, and probes of the host."""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the file the driver runs (child runs and self-launched ranks start it again)
BENCH_PY = os.path.join(ROOT, "bench.py")

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
FP32_VECTOR_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: FP32 vector (no matrix cores on this path)


def kernel_source_hash():
    """Identifies the kernel sources a PMC measurement under profiles/ belongs to: numbers that were not
    measured in this run are only attached to the line if the kernels have not changed since."""
    import hashlib
    base = os.path.join(ROOT, "vulkan_renderer_amd", "csrc")
    h = hashlib.sha256()
    # (what the shading, tracing and resolve kernels are compiled from, and the flags; host code - host/*.c,
    # shading_pass.hip around the kernels it instantiates - and the BVH builder do not change what they execute)
    for name in ("shading_kernel.h", "polygon_sampling.h", "related_work.h", "device_math.h", "glibc_math.h", "lbvh.h", "clip_cases.inc",
                 "wavefront_kernels.h", "light_shafts.h", "shading_variants.hip", "Makefile"):
        h.update(name.encode())
        h.update(open(os.path.join(base, name), "rb").read())
    return h.hexdigest()[:16]


def pmc_entry_for(table, config, mode, scene, width, height, world, csrc_hash):
    """The entry of profiles/pmc_traffic.json (rocprofv3 --pmc passes, profiles/collect.sh + summarize.py) that belongs to a
    workload, or None.  Entries belong to one configuration, arithmetic mode, frame size AND scene ("config3_libm" = the
    benchmark scene, "config3_libm_large" = the 2.6 M-triangle one: round 4 attached the benchmark scene's counters to the
    large scene's line) and to one rank rendering the whole frame; `stale` says that the kernels have changed since."""
    key = "config%s_%s%s" % (config, mode, "" if scene == "bench" else "_" + scene)
    entry = table.get(key)
    if not entry or world != 1 or width != entry.get("width") or height != entry.get("height") or entry.get("scene", "bench") != scene:
        return None
    pmc = dict(entry)
    pmc["valu_floor_us"] = table.get(key + "_valu_floor_us")
    pmc["stale"] = entry.get("csrc_hash") != csrc_hash
    return pmc


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def algorithmic_bytes_per_pixel(light_count, sample_count, techniques):
    """visibility id + 3 vertices (positions, normals/uv) + material id
    + 4 LTC texels + noise texels + RGBA32F out."""
    noise_fetches = math.ceil(light_count * sample_count * techniques / 2)
    return 4 + 49 + 48 + 8 * noise_fetches + 16


def available_cpus():
    """Host threads this process may really use: affinity mask and the cgroup's CPU quota (a
    container sees all cores of the machine in os.cpu_count() but is throttled to its quota:
    256 threads on a quota of a few cores ran in bursts of 100 ms periods)."""
    count = os.cpu_count() or 1
    try:
        count = min(count, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        fields = open("/sys/fs/cgroup/cpu.max").read().split()  # cgroup v2: "<quota|max> <period>"
        if fields and fields[0] != "max":
            quota = float(fields[0]) / float(fields[1])
    except (OSError, ValueError, IndexError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota:
        count = max(1, min(count, int(math.ceil(quota))))
    return count


def libm_identity():
    """Which C library the "libm" of the oracle is on this machine: bit-parity of the default arithmetic mode is
    parity with THIS library's float functions (csrc/glibc_math.h restates glibc 2.35's x86-64 FMA / AVX2 variants)."""
    import platform
    name, version = platform.libc_ver()
    flags = set()
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                flags = set(line.split(":", 1)[1].split())
                break
    except OSError:
        pass
    variant = "FMA + AVX2 IFUNC variants (__sinf_fma, __log2f_fma, ...)" if {"fma", "avx2"} <= flags else "baseline SSE2 variants (no FMA: differs from what the kernels restate)"
    return "%s %s, %s, %s" % (name or "libc", version or "?", platform.machine(), variant)


def parse_config(text):
    return text if text == "target" else int(text)
