#!/usr/bin/env python3
"""Benchmark of the shading pass (BASELINE.json: Msamples/s = pixels x spp / s).

  python bench.py --gpus 1 --steps K --warmup W
  python bench.py --gpus N ...                      (starts its N ranks itself: launch_ranks() below)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
  python bench.py --gpus N --dry-launch             (no GPU: the ranks only rendezvous over gloo)

A step is one pass of the shading kernels over one frame: write_constants -> upload ->
shade / trace / resolve over the rank's tiles [-> all-gather of the tile slabs -> scatter
into the frame].  Inputs (scene, BVH, LTC and noise tables, visibility buffer) are resident
in HBM before the timed region; data is synthetic (seeded generators,
vulkan_renderer_amd/synthetic.py).

Workload.  N = 1 runs BASELINE config 3 (1920x1080, 4 spp per technique, 4 polygonal lights,
diffuse + specular MIS with the clamped optimal heuristic, shadow rays), the heaviest 1080p
configuration; --config 1 | 2 | 4 | target select the others (target = north_star's
"1920x1080, 4 spp, 1 light").  N > 1 scales STRONGLY: the same fixed frame is cut into tiles,
tile t goes to rank t mod N, every rank shades its tiles into a dense slab, one RCCL
all-gather (ncclAllGather called from C behind the C-ABI, include/vkr_slab_exchange.h) per
frame delivers all slabs to all ranks and a scatter kernel reassembles the frame - all inside
the timed region, overlapped with the shading of the next frame.  The value at N = 1 is the
plain single-GPU pass (no exchange), so the per-N values of one workload are comparable.
Every run also measures BASELINE config 4 (3840x2160, 8 spp, 8 lights: the configuration
BASELINE.json tiles over 8 GPUs) as a second workload and attaches it as "secondary" to the
JSON line (--no-secondary to skip).  --scaling weak / --exchange none keep the round-1 mode
(frame height x N, slabs stay where they are) as an option.

PyTorch is plumbing here: device selection, the stream, the process group used for the
rendezvous token, barriers and the max over ranks of the timing.
"""
import argparse
import ctypes
import json
import math
import os

# The frame streams of the shading pass and the exchange stream need hardware queues of their own
# next to torch's and RCCL's streams; the HIP runtime multiplexes all streams onto 4 queues unless
# told otherwise.  Must be set before the runtime initialises (i.e. before torch is imported).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

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


def live_traffic(job, config, scene, timeout=150):
    """HBM bytes per launch of the dominant kernel, MEASURED IN THIS RUN: two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do
    not fit one pass: MI355X_MICROARCH.md, PMC slots) around a short child run of this file on the same workload - one frame at a
    time, one launch per frame -, counters averaged over the dispatches of shade_pixels.  FETCH_SIZE counts wide coalesced reads
    at half their bytes on gfx950 (the guide's HBM section): doubled.  None if rocprofv3 is not usable here (the caller then falls
    back to the committed passes of profiles/pmc_traffic.json and says so)."""
    import csv
    import glob
    import shutil
    import subprocess
    args = job.args
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof) or any(k.startswith(("ROCPROFILER_", "ROCP_TOOL")) for k in os.environ):
        return None
    out = tempfile.mkdtemp(prefix="vkr_bench_pmc_")
    child = [sys.executable, os.path.abspath(__file__), "--config", str(config), "--scene", scene, "--mode", args.mode, "--bvh", args.bvh, "--ltc-resolution", str(args.ltc_resolution),
             "--no-cpu-baseline", "--no-secondary", "--no-other-modes", "--no-extra", "--no-live-pmc", "--no-host-frames", "--frames-in-flight", "1", "--steps", "6", "--warmup", "2", "--prewarm-frames", "8",
             "--details", os.path.join(out, "child_details.json")]
    env = dict(os.environ, TMPDIR="/tmp", VKR_BENCH_DATASET_CACHE=job.dataset_cache or job.tmp.name, VKR_BAND_COUNT="1")
    for key in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)
    sums = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            directory = os.path.join(out, counter)
            command = [rocprof, "--kernel-trace", "--kernel-include-regex", "shade_pixels", "--output-format", "csv", "--pmc", counter, "-d", directory, "-o", "pmc", "--"] + child
            done = subprocess.run(command, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout)
            values = []
            for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(path)):
                    if "shade_pixels" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                        values.append(float(row["Counter_Value"]))
            if done.returncode != 0 or not values:
                return None
            sums[counter] = (sum(values) / len(values), len(values))
    except Exception:
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)
    fetch_kib, write_kib = sums["FETCH_SIZE"][0], sums["WRITE_SIZE"][0]
    return {"hbm_bytes_per_launch": int((2.0 * fetch_kib + write_kib) * 1024), "fetch_size_kib": round(fetch_kib, 1), "write_size_kib": round(write_kib, 1),
            "dispatches": [sums["FETCH_SIZE"][1], sums["WRITE_SIZE"][1]],
            "source": "measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over %d / %d dispatches of the kernel (one frame at a time), 2 x FETCH_SIZE + WRITE_SIZE (the guide's correction for wide reads on gfx950)" % (sums["FETCH_SIZE"][1], sums["WRITE_SIZE"][1])}


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


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(rank_count, argv):
    """`python bench.py --gpus N` without a launcher around it: starts N copies of this script, one per
    GPU, with the environment torch.distributed.run would give them (RANK, LOCAL_RANK, WORLD_SIZE,
    MASTER_ADDR = 127.0.0.1, a free MASTER_PORT) and waits for them.  The ranks inherit stdout, so the
    one JSON line rank 0 prints is the last line of this process's output too.  If a rank fails, the
    others are stopped (by PID) and its exit code is returned."""
    import signal
    import subprocess
    port = free_port()
    children = []
    for rank in range(rank_count):
        env = dict(os.environ)
        env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(rank_count), "LOCAL_WORLD_SIZE": str(rank_count),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "VKR_BENCH_SELF_LAUNCHED": "1"})
        # dmabuf IPC is the only kind the host driver supports (RCCL fails with hipIpcGetMemHandle otherwise)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("GPU_MAX_HW_QUEUES", "8")
        children.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    exit_code = 0
    pending = set(range(rank_count))
    try:
        while pending:
            for rank in sorted(pending):
                code = children[rank].poll()
                if code is None:
                    continue
                pending.discard(rank)
                if code != 0 and exit_code == 0:
                    exit_code = code if code > 0 else 1
                    print("bench.py: rank %d exited with %d; stopping the other ranks" % (rank, code), file=sys.stderr, flush=True)
                    for other in pending:
                        children[other].send_signal(signal.SIGTERM)
            time.sleep(0.05)
    except KeyboardInterrupt:
        for rank in pending:
            children[rank].send_signal(signal.SIGTERM)
        exit_code = 130
    for child in children:
        try:
            child.wait(timeout=10)
        except Exception:
            child.kill()
    return exit_code


def dry_launch(args):
    """--dry-launch: what every rank does before it touches a GPU - join the process group (gloo, CPU),
    carry rank 0's 128-byte rendezvous token to all ranks, a barrier and a max over ranks - and one
    JSON line from rank 0.  Proves that the launch path of `--gpus N` works on a machine without GPUs."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    t0 = time.perf_counter()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    token = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        token.copy_(torch.arange(128, dtype=torch.uint8) * 3 + 1)
    seen = torch.tensor([1.0], dtype=torch.float64)
    slowest = torch.tensor([float(rank)], dtype=torch.float64)
    if world > 1:
        dist.broadcast(token, src=0)
        dist.all_reduce(seen, op=dist.ReduceOp.SUM)
        dist.all_reduce(slowest, op=dist.ReduceOp.MAX)
        dist.barrier()
    token_ok = bool((token == torch.arange(128, dtype=torch.uint8) * 3 + 1).all())
    if world > 1:
        dist.destroy_process_group()
    if not token_ok:
        raise SystemExit("rank %d did not receive rank 0's token" % rank)
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": world, "ranks_seen": int(seen.item()), "highest_rank": int(slowest.item()), "token_ok": token_ok,
                          "self_launched": os.environ.get("VKR_BENCH_SELF_LAUNCHED") == "1", "backend": "gloo",
                          "rendezvous_ms": round((time.perf_counter() - t0) * 1e3, 1), "master_port": int(os.environ.get("MASTER_PORT", "0"))}), flush=True)


class Job:
    """What all workloads of one bench.py run share: ranks, torch handles, the dataset."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.args = torch, dist, args
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus and not (self.world == 1 and args.gpus == 1):
            raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, self.world))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X; there is no CPU fallback for the shading pass")
        # VKR_BENCH_DEVICE / VKR_BENCH_BACKEND=gloo: several ranks on ONE GPU with a CPU process group, to
        # exercise the N > 1 code paths of this file on a single-GPU box (profiles/tools/two_ranks_one_gpu.sh)
        if os.environ.get("VKR_BENCH_DEVICE"):
            self.local_rank = int(os.environ["VKR_BENCH_DEVICE"])
        self.backend = os.environ.get("VKR_BENCH_BACKEND", "nccl")
        self.collective_device = "cuda" if self.backend == "nccl" else "cpu"
        torch.cuda.set_device(self.local_rank)
        self.process_group = self.world > 1
        if self.process_group:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            if self.backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group(self.backend, rank=self.rank, world_size=self.world)
        self.tmp = tempfile.TemporaryDirectory(prefix="vkr_bench_%d_" % self.rank)
        self.datasets = {}
        self.dataset_cache = None
        self.dataset = self.dataset_of(args.scene)
        self.stream = torch.cuda.current_stream()

    def dataset_of(self, scene):
        """"bench": SURVEY.md 8(d), ground plane of 2 x 256^2 triangles + 64 boxes; "large": 2.6 M triangles with stacked
        occluders, long thin triangles, deep occlusion, eight materials (synthetic.make_large_scene_geometry).
        LTC tables with R = 64, 51 layers either way."""
        from vulkan_renderer_amd import synthetic
        if scene not in self.datasets:
            t = time.perf_counter()
            # VKR_BENCH_DATASET_CACHE=<directory>: the generated files are kept there and found again by later runs of one
            # profiling session (profiles/collect.sh starts bench.py dozens of times; the large scene takes 25 s to generate)
            cache = os.environ.get("VKR_BENCH_DATASET_CACHE")
            self.dataset_cache = cache
            # (the same layout in the run's own temporary directory: the child run of live_traffic() finds the files there)
            directory = os.path.join(cache or self.tmp.name, "%s_R%d_rank%d" % (scene, self.args.ltc_resolution, self.rank))
            marker = os.path.join(directory, "dataset.json")
            if os.path.exists(marker):
                self.datasets[scene] = json.load(open(marker))
            else:
                if scene == "large":
                    self.datasets[scene] = synthetic.write_dataset(directory, seed=4321, ltc_resolution=self.args.ltc_resolution, fresnel_count=51, large={})
                else:
                    self.datasets[scene] = synthetic.write_dataset(directory, grid=256, box_count=64, seed=1234, ltc_resolution=self.args.ltc_resolution, fresnel_count=51)
                json.dump(self.datasets[scene], open(marker, "w"))
            self.datasets[scene]["generate_seconds"] = round(time.perf_counter() - t, 2)
        return self.datasets[scene]

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.process_group:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def max_over_ranks(self, value):
        if not self.process_group:
            return float(value)
        t = self.torch.tensor([value], dtype=self.torch.float64, device=self.collective_device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather_floats(self, values):
        """-> list over ranks of lists"""
        if not self.process_group:
            return [list(values)]
        t = self.torch.tensor(list(values), dtype=self.torch.float64, device=self.collective_device)
        out = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [[float(v) for v in o.tolist()] for o in out]

    def broadcast_bytes(self, payload, count):
        """rank 0's bytes on every rank (the rendezvous token of the C-side communicator)"""
        if not self.process_group:
            return payload
        t = self.torch.zeros(count, dtype=self.torch.uint8, device=self.collective_device)
        if self.rank == 0:
            t.copy_(self.torch.frombuffer(bytearray(payload), dtype=self.torch.uint8))
        self.dist.broadcast(t, src=0)
        return bytes(t.cpu().numpy().tobytes())

    def host_staged_gather(self):
        """A slab_gather_function_t for a CPU process group: wait for the slab, copy it to the host, all-gather
        there, copy the gathered slabs back.  Slow and synchronous - it exists so that the N-rank schedule of
        this file can run end to end on a box with one GPU, never for a number."""
        hip = ctypes.CDLL("libamdhip64.so")
        torch, dist = self.torch, self.dist

        def gather(rank, buffer_set, send, gathered, send_bytes, stream):
            mine = torch.empty(send_bytes, dtype=torch.uint8)
            everyone = torch.empty(send_bytes * self.world, dtype=torch.uint8)
            if hip.hipStreamSynchronize(ctypes.c_void_p(stream)):
                return 1
            if hip.hipMemcpy(ctypes.c_void_p(mine.data_ptr()), ctypes.c_void_p(send), ctypes.c_size_t(send_bytes), 2):
                return 1
            dist.all_gather_into_tensor(everyone, mine)
            return int(hip.hipMemcpy(ctypes.c_void_p(gathered), ctypes.c_void_p(everyone.data_ptr()), ctypes.c_size_t(send_bytes * self.world), 1) != 0)
        return gather

    def close(self):
        if self.process_group:
            self.dist.destroy_process_group()
        self.tmp.cleanup()


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


def protocol_window(frames_in_flight, least=8):
    """Frames between two timing brackets of a pipelined run: the smallest multiple of the frames in flight that is at
    least `least`.  Frames in flight finish in BURSTS - n shading kernels share the GPU and end together, then nothing ends
    for n frame times (profiles/r10a/frame_periods.jsonl: periods of 0.2, 0.2, 3.0 ms with three in flight) - so the time
    between the ends of two frames that are k frames apart is a whole number of bursts, and its median over windows of k
    frames is biased unless n divides k: with k = 8 and n = 3 two windows in three span three bursts, one spans two, and
    the median sits 10 % above the mean (rounds 4 and 5 reported exactly that gap between `value` and `value_from_median`)."""
    n = max(1, int(frames_in_flight))
    return ((max(1, int(least)) + n - 1) // n) * n


def run_workload(job, config, role, scene=None):
    """Sets one BASELINE configuration up, times it and returns the dict that describes the run.
    role: "primary" (the headline: CPU baseline, parity, other arithmetic modes), "extra" (a short run of another
    1920x1080 configuration with parity bits and roofline, attached to the headline line) or "secondary" (config 4)."""
    primary = role == "primary"
    from vulkan_renderer_amd import renderer, synthetic
    args, torch = job.args, job.torch
    scene = scene or args.scene
    dataset = job.dataset_of(scene)
    rank, world = job.rank, job.world
    settings = dict(synthetic.CONFIG_SETTINGS[config])
    strong = args.scaling == "strong"
    exchange = args.exchange if (world > 1 or args.force_distributed) else "none"
    distributed = world > 1 or args.force_distributed
    width = args.width or settings["width"]
    height = (args.height or settings["height"]) * (1 if strong else world)
    if args.spp:
        settings["sample_count"] = args.spp
    sample_count = settings["sample_count"]
    if args.no_rays:
        settings["trace_shadow_rays"] = False
    steps = args.steps if args.steps is not None else {1: 2000, 2: 2000, 3: 500, 4: 100, "target": 1000}[config]
    warmup = args.warmup if args.warmup is not None else max(steps // 10, 1)
    if role == "secondary":
        steps, warmup = max(4, min(steps, 25)), max(1, min(warmup, 5))
    elif role == "extra":
        # at least 100 timed frames, so that the reference's protocol (median of >= 100 frame times) applies
        steps, warmup = 200, 20
    frames_in_flight_requested = args.frames_in_flight or renderer.frames_in_flight_for(world if (world > 1 or args.force_distributed) else 1)
    # frames between two timing brackets: a multiple of the frames in flight (protocol_window())
    window = protocol_window(frames_in_flight_requested, args.timing_stride)
    timing_stride = 1 if steps < 4 * window else window

    # ---- set-up (untimed, reported separately: BASELINE.md section 3) -------------------------
    r = renderer.Renderer(hip_device=job.local_rank, stream=job.stream.cuda_stream, arithmetic=args.mode, inline_rays=args.inline_rays,
                          timing_stride=timing_stride, frames_in_flight=frames_in_flight_requested, binary_traversal=args.binary_traversal)
    t = time.perf_counter()
    renderer.setup_config(r, config, dataset, width=width, height=height, sample_count=sample_count, acceleration_structure=args.bvh,
                          trace_shadow_rays=settings["trace_shadow_rays"])
    r.sync()
    load_ms = (time.perf_counter() - t) * 1e3
    structure = r.app.scene.acceleration_structure
    r.set_tiles(args.tile_size if distributed else 16, rank, world if distributed else 1, slab_layout=distributed)
    r.create_targets()
    r.create_pass()
    t = time.perf_counter()
    r.render_visibility()
    r.sync()
    first_visibility_ms = (time.perf_counter() - t) * 1e3
    # the first launch pays for code-object loading and buffer creation: the cost per frame is that of the later ones
    t = time.perf_counter()
    for _ in range(4):
        r.render_visibility()
    r.sync()
    visibility_ms = (time.perf_counter() - t) * 1e3 / 4
    light_count = r.app.scene_specification.polygonal_light_count
    techniques = 1 if settings["sampling_strategies"] == "diffuse_only" else 2
    total_pixels = width * height

    slab = None
    if exchange != "none":
        # the rendezvous token comes from rank 0 (ncclGetUniqueId behind the C-ABI) over the process group
        if job.backend == "nccl":
            token = job.broadcast_bytes(r.exchange_id() if rank == 0 else b"", 128)
            r.create_exchange(token, exchange)
        else:
            # CPU process group (VKR_BENCH_BACKEND=gloo: several ranks on ONE GPU, where RCCL refuses to form a
            # communicator): the same schedule with the collective staged through the host
            r.create_exchange_with_gather(job.host_staged_gather(), exchange)

        # the gathered slabs, tile-major, are the frame every rank holds; rows are made when somebody reads (the fences
        # of this run, once each) unless --assemble every-frame asks for the scatter kernel behind every all-gather
        r.assemble_on_demand(args.assemble == "on-demand")

        def step():
            r.render_and_exchange(None)

        def drain():
            r.finish_exchange()
    elif distributed:
        slab = torch.zeros((r.slab_pixel_count(0), 4), dtype=torch.float32, device="cuda")

        def step():
            r.render(slab.data_ptr())

        def drain():
            r.finish_frames()
    else:
        def step():
            r.render()

        def drain():
            r.finish_frames()

    def fence():
        drain()
        job.barrier()

    # clocks and the frame pipeline reach their steady state only after a few hundred frames (config 2:
    # 100 frames are 14 ms); the driver's --warmup 5 alone would time a cold GPU
    # Every rank must submit the SAME number of frames (each frame is one collective): the ranks agree on when the
    # prewarm ends - after a chunk of frames the slowest rank's clock decides for all.  (Until round 3 every rank
    # looked at its own clock, and a rank that fitted one frame more into the time than its peers left the job hanging
    # in its last all-gather: found by the first self-launched two-rank run, profiles/r05a/.)
    prewarm = 0
    t0 = time.perf_counter()
    prewarm_seconds = args.prewarm_seconds if role != "extra" else min(args.prewarm_seconds, 0.5)
    while prewarm < args.prewarm_frames:
        for _ in range(min(8 if prewarm == 0 else 16, args.prewarm_frames - prewarm)):
            step()
            prewarm += 1
        drain()
        torch.cuda.synchronize()
        if job.max_over_ranks(time.perf_counter() - t0) >= prewarm_seconds:
            break
    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    issue_seconds = time.perf_counter() - t0  # host time to queue the steps (a bound if the host cannot keep up)
    fence()
    elapsed = job.max_over_ranks(time.perf_counter() - t0)
    ms_per_step = elapsed / steps * 1e3
    value = total_pixels * sample_count / (elapsed / steps) / 1e6

    # ---- what the timed region looked like from the inside ---------------------------------------
    timed_frames = max(1, min(steps // max(timing_stride, 1), 256))
    overlapped_kernel_ms = r.shading_kernel_ms(timed_frames)
    period_ms = r.frame_period_ms(max(1, timed_frames - 1))
    launch_ms = r.dispatch_ms(timed_frames)
    pipelined = bool(r.app.shading_pass.last_frame_in_flight)
    bands_per_frame = int(r.app.shading_pass.last_band_count)
    frames_in_flight = int(r.app.shading_pass.last_frame_in_flight) if pipelined else 1
    rays = r.last_ray_count()
    shafts = r.light_shaft_statistics()
    # The reference's protocol is the median of at least 100 frame times (src/frame_timer.c:24,47-72, main.c:1958-1959).  A
    # run with fewer timed steps (the driver's --steps 20) renders 128 more frames behind the timed region for it; `value`
    # and `ms_per_step` stay those of the K timed steps.
    protocol_periods = None
    if primary and steps < 100:
        # sixteen more windows of `window` frames each (>= 128 frames), bracketed like the windows of a long run
        r.app.shading_pass.timing_stride = window
        for _ in range(17 * window):
            step()
        fence()
        protocol_periods = r.frame_period_ms(15)
        r.app.shading_pass.timing_stride = timing_stride
    # ---- every frame to the host (PCIe-inclusive; never `value`): a ring of targets, each read back through pinned staging
    # on the pass's copy stream while the next frames render (begin_read_back / end_read_back, include/vkr_shading_pass.h)
    with_readback = None
    if primary and not distributed and not args.no_host_frames:
        ring = [torch.empty((height, width, 4), dtype=torch.float32, device="cuda") for _ in range(frames_in_flight_requested + 1)]
        frame_bytes = width * height * 16

        def host_frames(count):
            for i in range(count):
                slot = i % len(ring)
                if i >= len(ring):
                    r.end_read_back(slot)  # the consumer takes frame i - len(ring) before its target and staging are reused
                r.render(ring[slot].data_ptr())
                r.begin_read_back(slot, ring[slot].data_ptr(), frame_bytes)
            for slot in range(min(count, len(ring))):
                r.end_read_back(slot)
        host_frames(2 * len(ring))  # (the first use of a slot allocates its pinned memory)
        frames_to_host = max(32, min(steps, 200))
        fence()
        t = time.perf_counter()
        host_frames(frames_to_host)
        fence()
        host_ms = (time.perf_counter() - t) / frames_to_host * 1e3
        with_readback = {"ms_per_frame": round(host_ms, 4), "value": round(total_pixels * sample_count / (host_ms * 1e-3) / 1e6, 3), "unit": "Msamples/s", "frames": frames_to_host,
                         "bytes_per_frame": frame_bytes, "GB_per_s": round(frame_bytes / (host_ms * 1e-3) / 1e9, 2), "targets": len(ring),
                         "note": "every frame lands in pinned host memory (RGBA32F, whole frame): render into a ring of device targets, begin_read_back() behind each frame on the pass's copy stream, end_read_back() when the ring comes round; PCIe-inclusive, never `value`"}
        del ring
    stages = None
    if exchange != "none":
        mine = r.exchange_ms() or [float("nan")] * 3
        per_rank = job.gather_floats(mine)
        stages = {"shade_ms": [round(v[0], 4) for v in per_rank], "all_gather_ms": [round(v[1], 4) for v in per_rank],
                  "scatter_ms": [round(v[2], 4) for v in per_rank],
                  "note": "per rank, HIP events of the most recent timed frame: shading (+ encoding) of the rank's slab on its frame stream, ncclAllGather and scatter on the exchange stream; they overlap the next frame, so they do not add up to ms_per_step"}
    # the assembled frame against a single-GPU render of the whole frame (rank 0)
    scaling_parity = None
    assembled = None
    if exchange != "none" and rank == 0:
        if exchange == "rgba32f":
            assembled = r.read_radiance()
        else:
            assembled = np.zeros((height, width, 4), np.uint8)
            r.lib.read_back_encoded(ctypes.byref(r.app), assembled.ctypes.data)
    if exchange != "none":
        r.destroy_exchange()
    visibility = r.read_visibility()
    own_pixels = r.slab_pixel_count(rank) if distributed else total_pixels
    if distributed:
        # shaded fraction of the pixels this rank owns
        xy = np.zeros((own_pixels, 2), np.uint32)
        slots = r.lib.get_slab_pixel_coordinates(ctypes.byref(r.app), rank, xy.ctypes.data, own_pixels)
        valid = xy[:slots, 0] != 0xFFFFFFFF
        own_visibility = visibility[xy[:slots][valid, 1], xy[:slots][valid, 0]]
    else:
        own_visibility = visibility.ravel()
    shaded = int((own_visibility != 0xFFFFFFFF).sum())
    background = int(own_visibility.size - shaded)
    bytes_per_launch = shaded * algorithmic_bytes_per_pixel(light_count, sample_count, techniques) + background * 20

    # ---- the dominant kernel alone: a short pass with one frame at a time, every frame timed -------
    # (one launch per frame, so that the events around the shading kernel bracket that kernel and nothing else)
    r.frames_in_flight, r.timing_stride, r.band_count = 1, 1, 1
    r.create_pass()
    target = slab.data_ptr() if slab is not None else None
    alone_frames = max(4, min(steps, 16))
    for _ in range(alone_frames + 3):
        r.render(target)
    r.sync()
    kernel_alone_ms = r.shading_kernel_ms(alone_frames)
    shaft_alone_ms = r.light_shaft_ms(alone_frames)
    pass_alone_ms = r.dispatch_ms(alone_frames)
    kernel_ms = float(np.mean(kernel_alone_ms)) if kernel_alone_ms else float("nan")
    traversal = None
    if (args.traversal_stats or primary or role == "extra") and rays and not args.inline_rays and world == 1:
        traversal = {}
        for wide in ([True, False] if structure.wide_nodes else [False]):
            s = r.traversal_statistics(wide)
            traversal[s["tree"]] = {"fetches_per_ray": round(s["node_visits"] / max(s["rays"], 1), 2), "boxes_tested_per_ray": round(s["boxes_tested"] / max(s["rays"], 1), 2),
                                    "triangle_tests_per_ray": round(s["triangle_tests"] / max(s["rays"], 1), 2), "lane_use": round(s["node_visits"] / max(64 * s["wave_steps"], 1), 3),
                                    "longest_ray_fetches": s["longest_ray_visits"], "blocked_fraction": round(s["blocked_rays"] / max(s["rays"], 1), 4)}
            if wide:
                visible_rays = max(s["rays"] - s["blocked_rays"], 1)
                traversal[s["tree"]].update({"fetches_per_blocked_ray": round(s["node_visits_of_blocked_rays"] / max(s["blocked_rays"], 1), 2),
                                             "fetches_per_visible_ray": round((s["node_visits"] - s["node_visits_of_blocked_rays"]) / visible_rays, 2),
                                             "deepest_stack": s["deepest_stack"], "rays_beyond_lds_stack": s["rays_beyond_lds_stack"]})
        traversal["walked"] = "wide" if (structure.wide_nodes and not args.binary_traversal) else "binary"
        traversal["note"] = ("replayed by a statistics kernel that walks the queued rays in batches of 64, each to its end: lane_use is what share of the lanes of such a batch "
                             "is busy per step - the figure by which a tracing wave decides to hand rays to idle lanes instead (below 0.65, csrc/wavefront_kernels.h)")
    if assembled is not None:
        # single-GPU render of the whole frame with the same pass settings
        r.set_tiles(16, 0, 1, slab_layout=False)
        r.render()
        single = r.read_radiance() if exchange == "rgba32f" else r.read_encoded(False, 0)
        differing = int((assembled.view(np.uint32) != single.view(np.uint32)).any(axis=-1).sum()) if exchange == "rgba32f" else int((assembled != single).any(axis=-1).sum())
        scaling_parity = {"pixels_differing_from_single_gpu_frame": differing, "pixels": total_pixels, "format": exchange}
        r.set_tiles(args.tile_size, rank, world, slab_layout=True)
    # PCIe-inclusive figures (never part of `value`): the frame to the host, a visibility buffer from the host
    t = time.perf_counter()
    gpu_image = r.read_radiance()
    pageable_readback_ms = (time.perf_counter() - t) * 1e3
    # ... and through the pinned staging of begin_read_back() / end_read_back() (second use of the slot: the first one allocates)
    r.begin_read_back(0)
    r.end_read_back(0)
    t = time.perf_counter()
    r.begin_read_back(0)
    r.end_read_back(0)
    readback_ms = (time.perf_counter() - t) * 1e3
    t = time.perf_counter()
    r.upload_visibility(visibility)
    r.sync()
    upload_ms = (time.perf_counter() - t) * 1e3

    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
    pass_ms = float(np.mean(period_ms if (pipelined and period_ms) else launch_ms)) if launch_ms else float("nan")
    # the reference's protocol (src/frame_timer.c:24,47-72, main.c:1958-1959): the median of at least 100 frame
    # times; here of the periods between the ends of consecutive timed frames inside the timed region
    median_ms = None
    if steps >= 100:
        slowest = job.max_over_ranks(float(np.median(period_ms)) if (period_ms and len(period_ms) >= 8) else -1.0)
        median_ms = slowest if slowest > 0.0 else None
    median_frames = (len(period_ms) * window) if median_ms else None
    if primary and steps < 100:
        # (every rank takes part in the reduction whatever it measured: a collective behind a local condition would hang)
        local = float(np.median(protocol_periods)) if (protocol_periods and len(protocol_periods) >= 8) else -1.0
        slowest = job.max_over_ranks(local)
        if slowest > 0.0:
            median_ms, median_frames = slowest, len(protocol_periods) * window
    pmc = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path):
        try:
            pmc = pmc_entry_for(json.load(open(pmc_path)), config, args.mode, scene, width, height, world, kernel_source_hash())
        except Exception:
            pmc = None
    # bound: what binds the dominant kernel - the issue of its VALU instructions (valu_issue below).  achieved /
    # peak / frac are the NOMINAL HBM figures SURVEY.md 8(d) prescribes for the metric (algorithmic bytes over the kernel's
    # duration against 8 TB/s): nominal_bound says so.
    live = None
    if primary and world == 1 and not distributed and not args.no_live_pmc and rank == 0:
        r.sync()
        live = live_traffic(job, config, scene)
    roofline = {"bound": "valu_issue", "nominal_bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 6),
                "traffic": live["hbm_bytes_per_launch"] if live else (pmc["hbm_bytes_per_launch"] if (pmc and not pmc["stale"]) else None),
                "traffic_live": live,
                "traffic_source": live["source"] if live else (("%s: rocprofv3 --pmc passes of this configuration and arithmetic mode (profiles/collect.sh), kernel sources %s" % (pmc.get("source", "profiles/pmc_traffic.json"), pmc.get("csrc_hash")))
                                   if not pmc["stale"] else "profiles/pmc_traffic.json has an entry, but for other kernel sources (%s, now %s): not attached" % (pmc.get("csrc_hash"), kernel_source_hash())) if pmc else None,
                "kernel": "shade_pixels<%s, V=%d, rays=%d, %s>" % (settings["sampling_strategies"], r.app.shading_pass.max_polygon_vertex_count, int(r.app.shading_pass.use_ray_tracing), args.mode),
                "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes_per_launch": bytes_per_launch,
                "kernel_ms_source": "HIP events around the kernel on its stream, %d frames with one frame at a time (nothing else on the GPU), run right after the timed region" % alone_frames,
                "pass_alone_ms": round(float(np.mean(pass_alone_ms)), 4) if pass_alone_ms else None,
                "light_shaft_kernel_ms": round(float(np.mean(shaft_alone_ms)), 4) if (shaft_alone_ms and shafts["pairs"]) else None,
                "overlapped": {"frames_in_flight": frames_in_flight, "frame_period_ms": round(pass_ms, 4),
                               "kernel_bracket_ms": round(float(np.mean(overlapped_kernel_ms)), 4) if overlapped_kernel_ms else None,
                               "note": "inside the timed region %d frames share the GPU: the bracket of one frame's shade_pixels then spans time in which the other frames' trace and resolve kernels run too, so it can exceed ms_per_step; it is not used for `achieved`" % frames_in_flight},
                "note": "nominal roofline (SURVEY.md 8d): the pass is bound by FP32 VALU issue and BVH latency, not by HBM"}
    if pmc and not pmc["stale"] and pmc.get("fp32_flop_per_launch"):
        # FP32 arithmetic of the dominant kernel: (ADD + MUL + 2 FMA) wave instructions x 64 lanes from the PMC passes over its
        # duration alone, against the FP32 vector peak
        tflops = pmc["fp32_flop_per_launch"] / (kernel_ms * 1e-3) / 1e12
        roofline["flops"] = {"achieved": round(tflops, 2), "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tflops / FP32_VECTOR_PEAK_TFLOPS, 4),
                             "fp32_flop_per_launch": pmc["fp32_flop_per_launch"], "source": "SQ_INSTS_VALU_{ADD,MUL,FMA}_F32 of %s (all 64 lanes counted), kernel sources %s; time live" % (pmc.get("source"), pmc.get("csrc_hash"))}
    if pmc and not pmc["stale"] and pmc.get("valu_floor_us"):
        # the bound that actually holds: wave64 VALU instructions per class, counted by the PMC passes in
        # profiles/, priced with the issue cost measured per class on this GPU (profiles/tools/valu_rate.hip:
        # 2.5 clocks add / mul / fma, 8.2 transcendental, 2.5 - 4.3 the rest; the mid-point is used), on 1024 SIMDs at 2.4 GHz
        floors = {k: v for k, v in pmc["valu_floor_us"].items() if isinstance(v, (int, float))}
        roofline["valu_issue"] = {"shade_pixels_floor_ms": round(floors.get("shade_pixels", 0.0) * 1e-3, 4),
                                  "shade_pixels_frac": round(floors.get("shade_pixels", 0.0) * 1e-3 / kernel_ms, 4),
                                  "floor_ms_per_pass": round(sum(floors.values()) * 1e-3, 4), "frac_of_ms_per_step": round(sum(floors.values()) * 1e-3 / ms_per_step, 4),
                                  "source": "per-class instruction counts from %s (rocprofv3 --pmc, not measured in this run) x issue clocks per class from profiles/r02d_valu_rate.txt; times live" % pmc.get("source", "profiles/pmc_traffic.json")}

    shaded_fraction = float(job.max_over_ranks(shaded / max(own_visibility.size, 1))) if distributed else shaded / max(own_visibility.size, 1)
    result = {
        "metric": "Msamples/s (pixels x spp / s), shading pass", "value": round(value, 3), "unit": "Msamples/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(ms_per_step, 4),
        "median_frame_period_ms": round(median_ms, 4) if median_ms else None, "median_over_frames": median_frames, "median_window_frames": window if median_ms else None,
        "value_from_median": round(total_pixels * sample_count / (median_ms * 1e-3) / 1e6, 3) if median_ms else None,
        "latency_ms": round(float(np.mean(pass_alone_ms)), 4) if pass_alone_ms else None,
        "value_single_frame": round(total_pixels * sample_count / (float(np.mean(pass_alone_ms)) * 1e-3) / 1e6, 3) if pass_alone_ms else None,
        "shaded_fraction": round(shaded_fraction, 4), "value_shaded_only": round(value * shaded_fraction, 3),
        "value_note": "value = W x H x spp / time over ALL pixels of the frame (SURVEY.md 8d), background included, with config.frames_in_flight frames queued like the reference's frame queue (main.h:374-390); latency_ms / value_single_frame = one frame at a time (roofline.pass_alone_ms); value_shaded_only counts the pixels that see geometry"
                      + ("; median_frame_period_ms = median over windows of median_window_frames frames - a multiple of the frames in flight, which finish in bursts - of the time between the ends of the window's first and last frame, per frame (the reference's protocol: median of >= 100 frame times, src/frame_timer.c:47-72)" if median_ms else "; the median of frame periods is reported from 100 steps on"),
        "higher_is_better": True, "scaling": args.scaling if world > 1 else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE config %s: %dx%d, %d spp per technique, %d polygonal light(s), %s + %s, %s, %s arithmetic"
                               % ("%s%s" % (config, "" if scene == "bench" else " on the large scene"), width, height, sample_count, light_count, settings["sampling_strategies"], settings["polygon_technique"],
                                  ("shadow rays through the %s BVH (%s)" % ("four-wide" if (structure.wide_nodes and not args.binary_traversal and not args.inline_rays) else "binary",
                                                                            renderer.BVH_BUILDER_NAME[int(structure.builder)])) if settings["trace_shadow_rays"] else "no shadow rays", args.mode),
                   "width": width, "height": height, "spp": sample_count, "lights": light_count, "techniques": techniques,
                   "parallelism": ("tiles %dx%d round-robin over %d rank(s), %s" % (args.tile_size, args.tile_size, world,
                                   ("RCCL all-gather of %s slabs (ncclAllGather from C, in place) %s inside the timed region, overlapped with the next frame" % (exchange, "+ scatter per frame" if args.assemble == "every-frame" else "per frame, un-tiled when read")) if exchange != "none" else "every rank keeps its slab of the frame (no data-path collective)")) if distributed else "one GPU, whole frame",
                   "scene": scene, "scene_triangles": int(r.app.scene.mesh.triangle_count), "scene_materials": int(r.app.scene.materials.material_count), "ltc_resolution": int(r.app.ltc_table.roughness_count),
                   "arithmetic": args.mode, "bands_per_frame": bands_per_frame, "frames_in_flight": frames_in_flight},
        "prewarm_frames": prewarm, "host_issue_ms_per_step": round(issue_seconds / steps * 1e3, 4),
        "shadow_rays_per_frame": rays, "Mrays_per_s": round(rays / (ms_per_step * 1e-3) / 1e6, 2) if rays else 0.0,
        "light_shafts": {"patch_light_pairs": shafts["pairs"], "clear_pairs": shafts["clear_pairs"], "clear_fraction": round(shafts["clear_pairs"] / max(shafts["pairs"], 1), 4), "occluder_list_pairs": shafts["list_pairs"], "triangles_per_occluder_list": round(shafts["listed_triangles"] / max(shafts["list_pairs"], 1), 2),
                         "not_clear": shafts["not_clear"], "work": shafts["work"],
                         "note": "csrc/light_shafts.h: one conservative BVH walk per (8x8 pixel patch, light) pair of the last launch; clear_pairs: no shadow ray of the pair can be blocked, none is queued; occluder_list_pairs: its rays can only meet the (at most 12) triangles of a list, and the shading kernel decides them against that list with the tracing kernel's triangle test; not_clear: the rays are queued and traced - shadow_rays_per_frame counts those; VKR_LIGHT_SHAFTS=0 traces all rays, VKR_SHAFT_LISTS=0 all but the clear pairs'; frames are bit-identical either way (tests/test_gpu_light_shafts.py)"},
        "setup": {"load_and_upload_ms": round(load_ms - structure.build_milliseconds, 2), "bvh_build_ms": round(float(structure.build_milliseconds), 3),
                  "bvh_builder": renderer.BVH_BUILDER_NAME[int(structure.builder)], "bvh_node_bytes": 16 * int(structure.node_count) + 64 * int(structure.wide_node_count),
                  "bvh_wide_nodes": int(structure.wide_node_count), "bvh_stack_need": int(structure.wide_stack_need), "visibility_pass_ms": round(visibility_ms, 3), "first_visibility_pass_ms": round(first_visibility_ms, 3),
                  "readback_ms": round(readback_ms, 3), "pageable_readback_ms": round(pageable_readback_ms, 3), "upload_ms": round(upload_ms, 3),
                  "note": "untimed set-up, once per scene (load = parse .vks / LTC fits / noise + copies to the device); visibility_pass_ms = primary visibility per frame (mean of 4 launches after the first, host clock around a synchronised device), first_visibility_pass_ms includes one-time costs; readback = RGBA32F frame into pinned staging (begin_read_back + end_read_back, nothing else running), pageable_readback = read_back_radiance() into pageable memory, upload = a visibility buffer from the host; with_readback (top level) = frames per second when EVERY frame goes to the host while the next ones render; never part of value"},
        "roofline": roofline,
    }
    if with_readback:
        result["with_readback"] = with_readback
    if stages:
        result["stages"] = stages
    if scaling_parity:
        result["scaling_parity"] = scaling_parity
    if traversal:
        result["traversal"] = traversal

    # ---- CPU baseline and parity (rank 0, N = 1, first workload only) ------------------------------
    if primary and rank == 0 and world == 1 and not distributed and not args.no_cpu_baseline:
        import oracle
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from helpers import classify_outliers
        inputs = r.host_inputs(visibility)
        bvh = oracle.Bvh(inputs["quantized_positions"], inputs["dequantization_factor"], inputs["dequantization_summand"])
        frame_o = oracle.make_frame(inputs, r.oracle_settings(), bvh)
        cores = available_cpus()
        # The oracle deals 64-pixel chunks to all host threads; bands of at least `cores` rows keep
        # the per-call overhead (256 threads waking up) small next to the work.  Calibrate on one
        # band in the middle (after a call that starts the thread pool), then spread bands over
        # the frame for ~12 s of CPU time.
        band = int(min(height, max(24, cores)))
        mid = max(0, height // 2 - band // 2)
        # The oracle in its libm mode is the arithmetic that is pinned bit for bit against the reference's
        # shader source (tests/test_reference_live.py) - what the default mode of the kernels reproduces
        # and what every mode is measured against; the polynomial "exact" mode is also compared with the
        # oracle's matching polynomial mode (bit-comparable).
        matching_mode = renderer.ORACLE_MATH_MODE[args.mode]
        oracle.set_math_mode(matching_mode)
        oracle.shade(frame_o, mid, mid + band, cores)
        t = time.perf_counter()
        oracle.shade(frame_o, mid, mid + band, cores)
        per_row = max((time.perf_counter() - t) / band, 1e-7)
        rows_budget = int(min(height, max(band, 12.0 / per_row)))
        bands = max(1, rows_budget // band)
        starts = sorted(set(int(i * (height - band) / max(bands - 1, 1)) for i in range(bands)))
        cpu_time = 0.0
        covered = np.zeros(height, bool)
        cpu_frames = {matching_mode: np.zeros((height, width, 4), np.float32)}
        for y0 in starts:
            t = time.perf_counter()
            cpu = oracle.shade(frame_o, y0, y0 + band, cores)
            cpu_time += time.perf_counter() - t
            cpu_frames[matching_mode][y0:y0 + band] = cpu[y0:y0 + band]
            covered[y0:y0 + band] = True
        sample_pixels = int(covered.sum()) * width
        timed_pixels = len(starts) * band * width
        # cheap configurations: repeat the sample until about ten seconds of CPU work are timed
        passes = 1
        while cpu_time < 10.0 and passes < 4096:
            t = time.perf_counter()
            for y0 in starts:
                oracle.shade(frame_o, y0, y0 + band, cores)
            cpu_time += time.perf_counter() - t
            passes += 1
        result["cpu_baseline"] = {"value": round(passes * timed_pixels * sample_count / cpu_time / 1e6, 4), "unit": "Msamples/s", "cores": cores, "cpu": cpu_model(),
                                  "kind": "port", "seconds": round(cpu_time, 2),
                                  "sample": "%d passes over %d bands of %d rows: %d of %d pixels of the frame" % (passes, len(starts), band, sample_pixels, total_pixels),
                                  "implementation": "CPU oracle (C99 restatement of the reference GLSL, OpenMP over 64-pixel chunks, %s)" % ("deterministic polynomial math" if matching_mode == 1 else "libm math: the mode that is bit-identical to the reference's shader source compiled as C++")}
        result["speedup_vs_cpu"] = round(value / result["cpu_baseline"]["value"], 1)
        # the other oracle mode over the same rows (untimed)
        if matching_mode != 0:
            oracle.set_math_mode(0)
            cpu_frames[0] = np.zeros((height, width, 4), np.float32)
            for y0 in starts:
                cpu_frames[0][y0:y0 + band] = oracle.shade(frame_o, y0, y0 + band, cores)[y0:y0 + band]
        oracle.set_math_mode(0)

        def against(cpu_frame):
            g, c = gpu_image[covered], cpu_frame[covered]
            stats = classify_outliers(g, c)
            stats.pop("other_coordinates", None)
            stats["pixels_differing_in_bits"] = int((g[..., :3].view(np.uint32) != c[..., :3].view(np.uint32)).any(axis=-1).sum())
            stats["max_abs"] = float(np.abs(np.nan_to_num(g[..., :3].astype(np.float64) - c[..., :3], nan=1e3)).max())
            return stats

        libm = against(cpu_frames[0])
        result["parity"] = {
            "tolerance_rmse": 1e-4, "sample_pixels": sample_pixels, "nan": int(np.isnan(gpu_image).sum()),
            "vs_libm_oracle": libm,
            "rmse_vs_libm_oracle": libm["rmse"], "pixels_over_1e-2": libm["pixels_over_threshold"], "guard_pixels": libm["guard_pixels"],
            "libm_oracle": "oracle math mode 0: C library transcendentals, IEEE division / sqrt; bit-identical to the reference's GLSL compiled as C++ (tests/test_reference_live.py, tests/test_oracle_golden.py); what --mode libm reproduces bit for bit",
            "libm": oracle.libm_description() + "; this machine: " + libm_identity(),
            "rule": "RMSE <= 1e-4 over all pixels that do not sit on a discontinuity of the shader; every pixel that differs by more than 1e-2 is a NaN-guard pixel (shading_pass.frag.glsl:861-864) "
                    "or a shadow-ray silhouette, else it counts as `other_pixels` and the run is out of tolerance (tests/helpers.py classify_outliers; silhouettes need the frames without rays: tests/test_gpu_full_size.py)",
            "within_tolerance": bool(libm["rmse_without_outliers"] <= 1e-4 and (libm["pixels_over_threshold"] == libm["guard_pixels"])),
        }
        if matching_mode != 0:
            result["parity"]["vs_polynomial_oracle"] = against(cpu_frames[matching_mode])
            result["parity"]["polynomial_oracle"] = "oracle math mode 1: the polynomial transcendentals that --mode exact mirrors operation for operation (bit-comparable)"
        # (kept for readers of earlier rounds' lines: the comparison with the oracle mode that matches --mode)
        matched = result["parity"]["vs_polynomial_oracle"] if matching_mode != 0 else libm
        result["parity"]["rmse_vs_oracle"] = matched["rmse"]
        result["parity"]["pixels_differing"] = matched["pixels_differing_in_bits"]
    if role == "extra" and rank == 0 and world == 1 and not distributed and not args.no_cpu_baseline:
        # three bands of the frame against the oracle in the arithmetic the mode reproduces (libm: every bit)
        import oracle
        inputs = r.host_inputs(visibility)
        bvh = oracle.Bvh(inputs["quantized_positions"], inputs["dequantization_factor"], inputs["dequantization_summand"])
        frame_o = oracle.make_frame(inputs, r.oracle_settings(), bvh)
        cores = available_cpus()
        band = int(min(height, 24))
        starts = sorted(set(int(f * (height - band)) for f in (1 / 6, 1 / 2, 5 / 6)))
        if config == 1:
            # BASELINE configs[0] is the CPU-runnable case: the WHOLE frame against the oracle, and the oracle timed beside it below
            band, starts = height, [0]
        oracle.set_math_mode(renderer.ORACLE_MATH_MODE[args.mode])
        differing, compared, worst = 0, 0, 0.0
        for y0 in starts:
            cpu = oracle.shade(frame_o, y0, y0 + band, cores)[y0:y0 + band]
            g = gpu_image[y0:y0 + band]
            differing += int((g[..., :3].view(np.uint32) != cpu[..., :3].view(np.uint32)).any(axis=-1).sum())
            compared += band * width
            worst = max(worst, float(np.abs(np.nan_to_num(g[..., :3].astype(np.float64) - cpu[..., :3], nan=1e3)).max()))
        if config == 1:
            # "CPU C reference of polygon_sampling math (plumbing, no GPU)": the same frame on the host cores, about two seconds of it
            passes, cpu_time = 0, 0.0
            while cpu_time < 2.0 and passes < 4096:
                t = time.perf_counter()
                oracle.shade(frame_o, 0, height, cores)
                cpu_time += time.perf_counter() - t
                passes += 1
            result["cpu_only"] = {"value": round(passes * total_pixels * sample_count / cpu_time / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port", "seconds": round(cpu_time, 2),
                                  "ms_per_frame": round(cpu_time / passes * 1e3, 3), "sample": "%d passes over the whole %dx%d frame" % (passes, width, height)}
        oracle.set_math_mode(0)
        result["parity"] = {"sample_pixels": compared, "sample": "%d bands of %d rows" % (len(starts), band), "pixels_differing_in_bits": differing, "max_abs": worst,
                            "nan": int(np.isnan(gpu_image).sum()), "oracle_math_mode": renderer.ORACLE_MATH_MODE[args.mode],
                            "within_tolerance": bool(differing == 0) if args.mode != "fast" else None}
    if primary and rank == 0 and world == 1 and not distributed and args.mode != "fast" and not args.no_other_modes and not args.inline_rays and not args.no_rays:
        r.close()
        result["other_modes"] = {}
        for other in ("exact",) if args.no_fast_mode else ("exact", "fast"):
            if other != args.mode:
                result["other_modes"][other] = mode_companion(job, config, other, gpu_image, width, height, sample_count, max(20, min(steps, 200)), frames_in_flight_requested)
        return result
    r.close()
    return result


def mode_companion(job, config, mode, headline_image, width, height, sample_count, steps, frames_in_flight):
    """The same workload in one of the cheaper arithmetic modes - exact: polynomial transcendentals, IEEE
    otherwise; fast: v_rcp / v_rsq / v_sqrt and contraction -, timed the same way and compared with the
    headline frame of this run (libm mode: the oracle's, bit for bit) under the same outlier rule.
    Reported next to the headline, never as the headline."""
    from vulkan_renderer_amd import renderer
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import classify_outliers
    args, torch = job.args, job.torch
    r = renderer.Renderer(hip_device=job.local_rank, stream=job.stream.cuda_stream, arithmetic=mode, timing_stride=protocol_window(frames_in_flight, args.timing_stride), frames_in_flight=frames_in_flight)
    renderer.setup_config(r, config, job.dataset, width=width, height=height, sample_count=sample_count, acceleration_structure=args.bvh)
    r.set_tiles(16, 0, 1, slab_layout=False)
    r.create_targets()
    r.create_pass()
    r.render_visibility()
    for _ in range(max(8, steps // 10)):
        r.render()
    r.finish_frames()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r.render()
    r.finish_frames()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    image = r.read_radiance()
    r.close()
    stats = classify_outliers(image, headline_image)
    if stats["pixels_over_threshold"] != stats["guard_pixels"]:
        # some outlier is not a NaN-guard pixel: the two modes' frames WITHOUT shadow rays tell a silhouette (a ray that passes
        # a triangle edge on the other side: the frames agree at that pixel once no ray is traced) from anything else
        without = {}
        for m in (mode, args.mode):
            q = renderer.Renderer(hip_device=job.local_rank, stream=job.stream.cuda_stream, arithmetic=m, frames_in_flight=1)
            renderer.setup_config(q, config, job.dataset, width=width, height=height, sample_count=sample_count, acceleration_structure=args.bvh, trace_shadow_rays=False)
            q.create_targets()
            q.create_pass()
            q.render_visibility()
            q.render()
            without[m] = q.read_radiance()
            q.close()
        stats = classify_outliers(image, headline_image, without[mode], without[args.mode])
    stats.pop("other_coordinates", None)
    # the rule of DESIGN.md section 2: every pixel that differs by more than 1e-2 is a NaN-guard pixel or a shadow-ray
    # silhouette, the rest is within 1e-4 RMSE
    within = bool(stats["rmse_without_outliers"] <= 1e-4 and stats["other_pixels"] == 0 and not np.isnan(image).any())
    return {"mode": mode, "within_tolerance": within, "value": round(width * height * sample_count / (ms * 1e-3) / 1e6, 3), "unit": "Msamples/s", "ms_per_step": round(ms, 4), "steps": steps,
            "rmse": stats["rmse"], "rmse_without_discontinuity_pixels": stats["rmse_without_outliers"], "guard_pixels": stats["guard_pixels"], "silhouette_pixels": stats["silhouette_pixels"], "other_pixels": stats["other_pixels"],
            "vs_headline_frame": stats, "nan": int(np.isnan(image).sum()), "tolerance_rmse": 1e-4,
            "note": "against the headline frame of this run (libm: the oracle's, bit for bit); pixels over 1e-2 are classified as NaN-guard pixels, shadow-ray silhouettes (the two modes agree there without shadow rays) or `other`, which fails the tolerance"}


LINE_LIMIT = 4096  # the driver keeps 8 KB of stdout; round 4's 25 KB line could not be parsed from that


def _pick(source, keys):
    return {k: source[k] for k in keys if source and k in source and source[k] is not None}


def _short_roofline(roofline):
    """bound / achieved / peak / unit / frac / traffic (the contract) + which kernel, its duration alone and the two
    fractions that say what really bounds it - numbers only, the sources are in the details file"""
    if not roofline:
        return None
    out = {k: roofline.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
    if roofline.get("nominal_bound"):
        out["nominal_bound"] = roofline["nominal_bound"]
    # (kernel_ms brackets shade_pixels alone; the shaft kernel that runs in front of it since round 4 is named next to it)
    out.update(_pick(roofline, ("kernel", "kernel_ms", "light_shaft_kernel_ms", "algorithmic_bytes_per_launch", "pass_alone_ms")))
    if roofline.get("flops"):
        out["flops"] = _pick(roofline["flops"], ("achieved", "peak", "unit", "frac"))
    if roofline.get("valu_issue"):
        out["valu_issue"] = _pick(roofline["valu_issue"], ("shade_pixels_frac", "frac_of_ms_per_step"))
    return out


def _short_parity(parity):
    if not parity:
        return None
    out = {"pixels_differing": parity.get("pixels_differing", parity.get("pixels_differing_in_bits")),
           "rmse": parity.get("rmse_vs_libm_oracle", parity.get("rmse_vs_oracle")), "tolerance_rmse": 1e-4}
    out.update(_pick(parity, ("sample_pixels", "nan", "within_tolerance")))
    if out["rmse"] is None:
        del out["rmse"]
    return out


def _short_workload(w):
    """a workload other than the headline: what it is, its value, its time, its roofline fraction, its parity"""
    out = {"workload": "%dx%d, %d spp, %d light(s)%s" % (w["config"]["width"], w["config"]["height"], w["config"]["spp"], w["config"]["lights"],
                                                        "" if w["config"].get("scene", "bench") == "bench" else ", %s scene" % w["config"]["scene"])}
    out.update(_pick(w, ("value", "steps", "ms_per_step", "median_frame_period_ms")))
    if w.get("roofline"):
        out["roofline"] = _pick(w["roofline"], ("frac", "traffic", "kernel_ms"))
    if w.get("parity"):
        out["parity"] = _pick(_short_parity(w["parity"]), ("pixels_differing", "sample_pixels", "within_tolerance"))
    if w.get("scaling_parity"):
        out["scaling_parity"] = _pick(w["scaling_parity"], ("pixels_differing_from_single_gpu_frame", "pixels"))
    if w.get("cpu_only"):
        out["cpu_only"] = _pick(w["cpu_only"], ("value", "cores"))
    return out


def short_line(result, details_path=None):
    """The ONE line bench.py prints: the contract's keys, numbers and short identifiers only (no prose), at most
    LINE_LIMIT characters.  Everything else the run measured - extra workloads in full, traversal and light-shaft
    statistics, set-up times, the other arithmetic mode, where each number comes from - goes to the details file."""
    line = _pick(result, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "median_frame_period_ms", "median_over_frames", "median_window_frames", "value_from_median",
                          "higher_is_better", "scaling", "dtype", "data"))
    line["vs_baseline"] = result.get("vs_baseline")
    line.update(_pick(result, ("value_shaded_only", "shaded_fraction", "latency_ms")))
    if result.get("with_readback"):
        line["value_with_readback"] = result["with_readback"].get("value")
    cfg = result.get("config", {})
    line["config"] = _pick(cfg, ("workload", "width", "height", "spp", "lights", "techniques", "scene", "scene_triangles", "arithmetic", "frames_in_flight", "parallelism"))
    for key in ("workload", "parallelism"):
        if len(str(line["config"].get(key, ""))) > 240:
            line["config"][key] = line["config"][key][:240]
    line["roofline"] = _short_roofline(result.get("roofline"))
    if result.get("cpu_baseline"):
        line["cpu_baseline"] = _pick(result["cpu_baseline"], ("value", "unit", "cores", "kind", "cpu", "seconds"))
        line["cpu_baseline"]["sample"] = str(result["cpu_baseline"].get("sample", ""))[:120]
    if result.get("parity"):
        line["parity"] = _short_parity(result["parity"])
    if result.get("scaling_parity"):
        line["scaling_parity"] = _pick(result["scaling_parity"], ("pixels_differing_from_single_gpu_frame", "pixels", "format"))
    if result.get("stages"):
        line["stages"] = {k: [round(v, 3) for v in result["stages"][k]] for k in ("shade_ms", "all_gather_ms", "scatter_ms") if k in result["stages"]}
    if result.get("north_star_target"):
        target = result["north_star_target"]
        line["north_star_target"] = _pick(target, ("shape", "target_Msamples_per_s", "value", "met"))
        if target.get("parity"):
            line["north_star_target"]["pixels_differing"] = _short_parity(target["parity"])["pixels_differing"]
    if result.get("secondary"):
        line["secondary"] = _short_workload(result["secondary"])
    extras = result.get("extra_workloads") or {}
    if extras:
        line["extra_workloads"] = {name: _pick(_short_workload(w), ("value", "ms_per_step", "parity", "cpu_only")) for name, w in extras.items()}
    if result.get("other_modes"):
        line["other_modes"] = {m: _pick(v, ("value", "ms_per_step", "within_tolerance", "rmse", "rmse_without_discontinuity_pixels", "guard_pixels", "silhouette_pixels", "other_pixels")) for m, v in result["other_modes"].items()}
    line["details"] = details_path
    # a long workload string is the first thing to go if the line ever outgrows the driver's buffer
    for drop in ("other_modes", "extra_workloads", "stages"):
        if len(json.dumps(line, separators=(",", ":"))) <= LINE_LIMIT:
            break
        line.pop(drop, None)
    if len(json.dumps(line, separators=(",", ":"))) > LINE_LIMIT:
        line["config"]["workload"] = str(line["config"].get("workload", ""))[:160]
    return line


def write_details(result, path=None):
    """Everything the run measured, with the prose: gpurun_out/bench_details.json (scratch that gpurun brings back;
    copies that are meant to be judged are committed under profiles/).  Returns the path relative to the repository."""
    path = path or os.path.join(ROOT, "gpurun_out", "bench_details.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(result, f, indent=1)
            f.write("\n")
    except OSError as e:
        print("bench.py: could not write %s: %s" % (path, e), file=sys.stderr)
        return None
    return os.path.relpath(path, ROOT)


def timing_matrix_cells(job):
    """Two cells of the reference's own timing matrix (src/experiment_list.c:366-409; all 260: profiles/tools/timing_matrix.py
    and profiles/r07h_timing_matrix.md), measured in this run by the reference's protocol - median frame time of 110 frames -
    with the matrix's settings: 1920x1080, diffuse only, projected solid angle sampling, no shadow rays, a decentral quad;
    128 lights x 1 sample and 1 light x 128 samples.  For the details file."""
    import importlib.util
    from vulkan_renderer_amd import renderer, synthetic
    spec = importlib.util.spec_from_file_location("timing_matrix", os.path.join(ROOT, "profiles", "tools", "timing_matrix.py"))
    matrix = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(matrix)
    dataset = job.dataset_of("bench")
    cells = {}
    r = renderer.Renderer(hip_device=job.local_rank, stream=job.stream.cuda_stream, frames_in_flight=1, timing_stride=1, arithmetic=job.args.mode)
    try:
        r.load_scene(dataset["scene"], dataset["textures"], acceleration_structure=True)
        r.load_ltc_table(dataset["ltc"], dataset["fresnel_count"])
        r.load_noise_table("white")
        cam = synthetic.DEFAULT_CAMERA
        r.set_camera(cam["position"], cam["rotation_x"], cam["rotation_z"], cam["vertical_fov"], cam["near"], cam["far"])
        r.set_settings(width=1920, height=1080, trace_shadow_rays=False, sampling_strategies="diffuse_only", polygon_technique="projected_solid_angle")
        r.set_lights(matrix.timing_lights(4, False, 1))
        r.create_targets()
        r.create_pass()
        r.render_visibility()
        for light_count, sample_count in ((128, 1), (1, 128)):
            r.set_settings(sample_count=sample_count)
            r.set_lights(matrix.timing_lights(4, False, light_count))
            r.create_pass()
            for _ in range(8):
                r.render()
            r.sync()
            for _ in range(110):
                r.render()
            r.sync()
            times = sorted(r.dispatch_ms(110))
            median = times[len(times) // 2]
            cells["%d_lights_x_%d_samples" % (light_count, sample_count)] = {
                "frame_ms": round(median, 4), "light_samples_per_s": round(1920 * 1080 * light_count * sample_count / (median * 1e-3), 0),
                "experiment": "timings_decentral_4%s_projected_solid_angle_ours" % ("_128" if light_count == 128 else "")}
    finally:
        # (also when a launch fails: the config-4 secondary that follows needs the memory)
        r.close()
    cells["protocol"] = "median of 110 frame times, one frame at a time (no shadow rays: a frame is one kernel), %s arithmetic; scene, noise and lights are stand-ins for the reference's downloaded assets (profiles/tools/timing_matrix.py)" % job.args.mode
    return cells


def parse_config(text):
    return text if text == "target" else int(text)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed frames (default 2000 / 2000 / 500 / 100 for configs 1-4)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed frames right before the timed ones (default: a tenth of the steps); --prewarm-frames come before them")
    ap.add_argument("--config", type=parse_config, default=3, choices=[1, 2, 3, 4, "target"], help="BASELINE.json configuration (default 3, the heaviest 1080p one); target = 1920x1080, 4 spp, 1 light")
    ap.add_argument("--scene", default="bench", choices=["bench", "large"], help="bench: the scene of SURVEY.md 8(d) (131 840 triangles); large: 2.6 M triangles with stacked occluders, thin triangles, deep occlusion, eight materials")
    ap.add_argument("--no-secondary", action="store_true", help="do not also measure BASELINE config 4 (3840x2160, 8 spp, 8 lights)")
    ap.add_argument("--mode", default="libm", choices=["libm", "exact", "fast"],
                    help="libm: IEEE arithmetic with glibc's transcendentals, bit-identical to the CPU oracle in the mode that is pinned against the reference shader (default); "
                         "exact: the libm mode with a polynomial arctangent, bit-identical to the oracle's math mode 1 (arithmetic_mode_polynomial of the C-ABI); fast: approximate reciprocals + contraction")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"], help="N > 1: strong = the fixed frame is cut into tiles (default); weak = the frame height grows with N")
    ap.add_argument("--exchange", choices=("rgba32f", "rgb8", "none"), default="rgba32f",
                    help="N > 1: all-gather of the tile slabs per frame as float radiance (default) or as packed RGB8 of the encoded output, then the scatter into the frame on every rank; none leaves every rank's slab in its HBM")
    ap.add_argument("--assemble", choices=("on-demand", "every-frame"), default="on-demand",
                    help="N > 1: on-demand (default) = a frame stays the gathered slabs (tile-major) and is un-tiled when it is read - once per fence of this run; every-frame = a scatter kernel behind every all-gather (SURVEY.md 8e names both)")
    ap.add_argument("--tile-size", type=int, default=32)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--bvh", default="sah_device", choices=["sah_device", "lbvh_device", "sah_host"], help="who builds the BVH (default: binned SAH by HIP kernels)")
    ap.add_argument("--binary-traversal", action="store_true", help="walk the binary tree in the wavefront kernel instead of the four-wide one")
    ap.add_argument("--no-rays", action="store_true", help="disable shadow rays (TRACE_SHADOW_RAYS=0) for experiments")
    ap.add_argument("--inline-rays", action="store_true", help="trace shadow rays inside the shading kernel instead of the wavefront path")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-modes", dest="no_other_modes", action="store_true", help="do not also measure the workload in the exact (polynomial arctangent) and fast arithmetic modes (reported as \"other_modes\" next to the headline)")
    ap.add_argument("--no-fast-mode", action="store_true", help="other_modes without the fast arithmetic mode (approximate reciprocals and roots, contraction, polynomial transcendentals)")
    ap.add_argument("--no-large-scene", action="store_true", help="do not also run config 3 on the large scene (2.6 M triangles) as an extra workload")
    ap.add_argument("--no-extra", action="store_true", help="do not also run the north_star target shape (1920x1080, 4 spp, 1 light) and BASELINE config 2 as short extra workloads")
    ap.add_argument("--traversal-stats", action="store_true", help="attach BVH traversal work counters to the secondary workload too (diagnostics)")
    ap.add_argument("--frames-in-flight", type=int, default=None, choices=(1, 2, 3, 4, 5, 6, 7, 8),
                    help="n >= 2: n consecutive frames overlap on the device's frame streams, like the frames of the reference's frame queue, which is as deep as "
                         "its swapchain (main.c:1498: typically 3).  Default: 3, and 4 from eight ranks on (renderer.frames_in_flight_for)")
    ap.add_argument("--timing-stride", type=int, default=8, help="bracket every n-th frame of the timed region with HIP events (rounded up to a multiple of the frames in flight: protocol_window())")
    ap.add_argument("--prewarm-frames", type=int, default=200, help="untimed frames before --warmup that bring clocks and the frame pipeline to their steady state")
    ap.add_argument("--prewarm-seconds", type=float, default=1.5, help="... but no longer than this (after the first eight)")
    ap.add_argument("--ltc-resolution", type=int, default=64, help="roughness / inclination resolution R of the generated LTC tables (SURVEY.md 8d: 64)")
    ap.add_argument("--no-host-frames", action="store_true", help="do not also measure the rate with every frame read back to the host (with_readback)")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not measure roofline.traffic in this run (two short rocprofv3 --pmc passes around a child run of the headline workload); the committed passes of profiles/pmc_traffic.json are used instead")
    ap.add_argument("--force-distributed", action="store_true", help="run the N > 1 code path (slab layout, exchange) even with one rank")
    ap.add_argument("--details", default=None, help="where the full record of the run goes (default gpurun_out/bench_details.json); the printed line is the short one")
    ap.add_argument("--dry-line", default=None, metavar="RECORD", help="no GPU work: print the short line for the full record of an earlier run (a details file, or a file whose last line is a record)")
    ap.add_argument("--dry-launch", action="store_true", help="no GPU work: the ranks join a gloo process group, exchange a token, print one JSON line and leave (checks the launch path of --gpus N)")
    args = ap.parse_args()

    if args.dry_line:
        text = open(args.dry_line).read()
        try:
            record = json.loads(text)
        except ValueError:
            record = json.loads([line for line in text.splitlines() if line.startswith("{")][-1])
        print(json.dumps(short_line(record, os.path.relpath(os.path.abspath(args.dry_line), ROOT)), separators=(",", ":")), flush=True)
        return
    # `python bench.py --gpus N` by itself: become the launcher of N ranks (one per GPU)
    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and "VKR_BENCH_SELF_LAUNCHED" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.dry_launch:
        dry_launch(args)
        return

    job = Job(args)
    result = run_workload(job, args.config, "primary")
    customised = bool(args.width or args.height or args.spp)
    if not args.no_extra and args.config == 3 and not customised and job.world == 1:
        # the other 1920x1080 shapes the contract names, on the same clock: north_star's target and BASELINE config 2
        keep = ("value", "unit", "steps", "warmup", "ms_per_step", "median_frame_period_ms", "value_from_median", "latency_ms", "value_single_frame", "shaded_fraction",
                "config", "shadow_rays_per_frame", "Mrays_per_s", "light_shafts", "roofline", "parity", "traversal", "setup", "cpu_only")
        result["extra_workloads"] = {}
        for extra, scene in (("target", args.scene), (2, args.scene), (1, args.scene)) + ((("3", "large"),) if args.scene == "bench" and not args.no_large_scene else ()):
            line = run_workload(job, int(extra) if extra != "target" else extra, "extra", scene)
            result["extra_workloads"]["config_%s%s" % (extra, "" if scene == args.scene else "_large_scene")] = {k: line[k] for k in keep if k in line}
        target_value = result["extra_workloads"]["config_target"]["value"]
        result["north_star_target"] = {"shape": "1920x1080, 4 spp, 1 polygonal light", "target_Msamples_per_s": 1000.0, "value": target_value, "met": bool(target_value >= 1000.0),
                                       "parity": result["extra_workloads"]["config_target"].get("parity")}
    if not args.no_extra and args.config == 3 and not customised and job.world == 1 and job.rank == 0:
        try:
            result["timing_matrix_cells"] = timing_matrix_cells(job)
        except Exception as error:  # (diagnostics for the details file: never a reason to lose the line)
            result["timing_matrix_cells"] = {"error": repr(error)}
    if not args.no_secondary and args.config != 4 and not customised:
        second = run_workload(job, 4, "secondary")
        keep = ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "config", "shadow_rays_per_frame", "Mrays_per_s", "light_shafts", "stages", "scaling_parity", "setup", "roofline", "traversal")
        result["secondary"] = {k: second[k] for k in keep if k in second}
    # the library reports like the reference does (printf): every rank flushes C stdio before rank 0
    # prints, so that the JSON line is the last line of the job's output
    ctypes.CDLL(None).fflush(None)
    job.barrier()
    if job.rank == 0:
        details_path = write_details(result, args.details)
        print(json.dumps(short_line(result, details_path), separators=(",", ":")), flush=True)
    job.close()


if __name__ == "__main__":
    main()
