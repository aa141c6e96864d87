"""Ranks of a bench run: the torch.distributed plumbing (Job), `python bench.py --gpus N` starting its N ranks itself, and the
rendezvous dry run that the CPU tests use."""
import ctypes
import json
import os
import sys
import tempfile
import time

from .common import BENCH_PY, ROOT


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
        children.append(subprocess.Popen([sys.executable, BENCH_PY] + list(argv), env=env))
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
