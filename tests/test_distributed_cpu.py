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
