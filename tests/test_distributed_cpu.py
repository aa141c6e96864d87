"""The multi-process tile path on CPU: two gloo ranks fill their slabs (layout from
the library's own host-side description), all-gather them like bench.py does over
RCCL, and scatter them back into a frame.  No kernel runs here; the kernels'
agreement with this layout is checked on the GPU by
test_gpu_golden.py::test_tiles_of_virtual_ranks_reassemble_bit_exactly."""
import ctypes as C
import os
import socket

import numpy as np
import pytest


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _pixel_value(x, y):
    return np.stack([x * 1.0, y * 1.0, x * 1000.0 + y, np.ones_like(x, dtype=np.float64)], -1).astype(np.float32)


def _worker(rank, world, port, width, height, tile_size, result_dir):
    import torch
    import torch.distributed as dist
    from vulkan_renderer_amd import capi
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = capi.load()
    app = capi.Application()
    app.swapchain.extent.width, app.swapchain.extent.height = width, height
    app.tile_schedule.tile_size, app.tile_schedule.rank, app.tile_schedule.rank_count = tile_size, rank, world
    slab_pixels = lib.get_slab_pixel_count(C.byref(app), 0)
    assert slab_pixels >= lib.get_slab_pixel_count(C.byref(app), rank)
    xy = np.zeros((slab_pixels, 2), np.uint32)
    slots = lib.get_slab_pixel_coordinates(C.byref(app), rank, xy.ctypes.data, slab_pixels)
    slab = np.zeros((slab_pixels, 4), np.float32)
    valid = xy[:slots, 0] != 0xFFFFFFFF
    slab[:slots][valid] = _pixel_value(xy[:slots][valid, 0].astype(np.float64), xy[:slots][valid, 1].astype(np.float64))
    gathered = torch.zeros((world * slab_pixels, 4), dtype=torch.float32)
    dist.all_gather_into_tensor(gathered, torch.from_numpy(slab))
    gathered = gathered.view(world, slab_pixels, 4)
    # every rank can rebuild the frame
    frame = np.full((height, width, 4), -1.0, np.float32)
    covered = np.zeros((height, width), np.int32)
    for r in range(world):
        rxy = np.zeros((slab_pixels, 2), np.uint32)
        n = lib.get_slab_pixel_coordinates(C.byref(app), r, rxy.ctypes.data, slab_pixels)
        ok = rxy[:n, 0] != 0xFFFFFFFF
        frame[rxy[:n][ok, 1], rxy[:n][ok, 0]] = gathered[r].numpy()[:n][ok]
        np.add.at(covered, (rxy[:n][ok, 1], rxy[:n][ok, 0]), 1)
    np.save(os.path.join(result_dir, "frame_%d.npy" % rank), frame)
    np.save(os.path.join(result_dir, "covered_%d.npy" % rank), covered)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("width,height,tile_size", [(200, 120, 16), (256, 144, 32), (130, 70, 64)])
def test_two_ranks_gather_tiles_into_the_full_frame(tmp_path, width, height, tile_size):
    import torch.multiprocessing as mp
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, width, height, tile_size, str(tmp_path)), nprocs=world, join=True)
    ys, xs = np.meshgrid(np.arange(height), np.arange(width), indexing="ij")
    expected = _pixel_value(xs.astype(np.float64), ys.astype(np.float64))
    for rank in range(world):
        frame = np.load(tmp_path / ("frame_%d.npy" % rank))
        covered = np.load(tmp_path / ("covered_%d.npy" % rank))
        assert np.all(covered == 1), "every pixel must belong to exactly one rank"
        assert np.array_equal(frame, expected)


def test_slab_sizes_are_balanced():
    from vulkan_renderer_amd import capi
    lib = capi.load()
    app = capi.Application()
    app.swapchain.extent.width, app.swapchain.extent.height = 1920, 8640
    app.tile_schedule.tile_size, app.tile_schedule.rank_count = 32, 8
    sizes = []
    for rank in range(8):
        app.tile_schedule.rank = rank
        sizes.append(lib.get_slab_pixel_count(C.byref(app), rank))
    assert max(sizes) - min(sizes) <= 32 * 32
    assert sum(sizes) >= 1920 * 8640


def _run_bench(arguments, extra_env=None, timeout=240):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "VKR_BENCH_SELF_LAUNCHED")}
    env.update(extra_env or {})
    done = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + arguments, env=env, cwd=root, capture_output=True, text=True, timeout=timeout)
    lines = [line for line in done.stdout.splitlines() if line.startswith("{")]
    return done, [json.loads(line) for line in lines]


@pytest.mark.parametrize("ranks", [2, 4])
def test_bench_starts_its_own_ranks_without_a_launcher(ranks):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment (how the driver starts the N = 1 bench):
    N ranks are spawned, rendezvous on a free port and rank 0 prints exactly one JSON line."""
    done, lines = _run_bench(["--gpus", str(ranks), "--dry-launch"])
    assert done.returncode == 0, done.stderr[-2000:]
    assert len(lines) == 1, done.stdout
    line = lines[0]
    assert line["dry_launch"] and line["self_launched"] and line["token_ok"]
    assert line["n_gpus"] == ranks and line["ranks_seen"] == ranks and line["highest_rank"] == ranks - 1


def test_bench_under_a_launcher_does_not_spawn_again():
    """with RANK / WORLD_SIZE given (torch.distributed.run, or the ranks bench.py spawned) the script is a rank"""
    port = _free_port()
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    children = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.pop("VKR_BENCH_SELF_LAUNCHED", None)
        children.append(subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-launch"], env=env, cwd=root, stdout=subprocess.PIPE, text=True))
    outputs = [child.communicate(timeout=240)[0] for child in children]
    assert all(child.returncode == 0 for child in children)
    assert '"self_launched": false' in outputs[0] and "dry_launch" not in outputs[1]


def test_a_failing_rank_ends_the_launcher_with_its_exit_code():
    """a rank count that does not match: every rank refuses, the launcher reports failure instead of hanging"""
    done, lines = _run_bench(["--gpus", "2", "--dry-launch"], {"VKR_BENCH_SELF_LAUNCHED": "1", "WORLD_SIZE": "1"})
    assert done.returncode != 0 and not lines
