"""The N > 1 paths, rehearsed with two gloo ranks on the CPU.

The GPU library has no CPU mode, so in these tests each rank's share of the
image is produced by the CPU oracle (test infrastructure) with exactly the
partition the GPU ranks use -- interleaved strips for tile mode, a strided frame
sequence with independent seeds for sample mode -- and the exchange code of
ray_tracing_amd.distributed is what is under test: the gathered or reduced image
must equal a single-process render bit for bit.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
from ray_tracing_amd import default_scene
from ray_tracing_amd.distributed import gather_strips, reduce_samples, sample_split, strip_rows

W, H, STRIP = 96, 54, 8


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_strip_rows_partition_the_canvas():
    for h, strip, world in ((54, 8, 2), (1080, 8, 8), (360, 16, 4), (7, 8, 3), (1080, 8, 1)):
        owned = [strip_rows(h, strip, r, world) for r in range(world)]
        allrows = np.concatenate(owned)
        assert sorted(allrows.tolist()) == list(range(h))
        for r, rows in enumerate(owned):
            assert all((y // strip) % world == r for y in rows) or world == 1
    assert sample_split(16, 1, 4) == (1, 4, 4) and sample_split(10, 3, 4) == (3, 4, 2)


def _worker(rank, world, port, mode, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        scene, cam = default_scene(W, H)
        if mode == "tile":
            r = O.OracleRenderer(scene, cam, trig=O.TRIG_DET, threads=1)
            r.render(3, rows=strip_rows(H, STRIP, rank, world))
            full = gather_strips(torch.from_numpy(r.acc), STRIP)
            if rank == 0:
                np.save(os.path.join(out_dir, "tile.npy"), full.numpy())
        else:
            first, stride, count = sample_split(6, rank, world)
            c = type(cam).from_buffer_copy(cam)
            c.frame = first
            r = O.OracleRenderer(scene, c, trig=O.TRIG_DET, threads=1, frame_stride=stride, independent_seeds=True)
            r.render(count)
            acc = reduce_samples(torch.from_numpy(r.acc))
            if rank == 0:
                np.save(os.path.join(out_dir, "sample.npy"), acc.numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["tile", "sample"])
def test_two_ranks_reproduce_the_single_process_image(tmp_path, mode):
    world = 2
    mp.spawn(_worker, args=(world, free_port(), mode, str(tmp_path)), nprocs=world, join=True)
    scene, cam = default_scene(W, H)
    if mode == "tile":
        want = O.OracleRenderer(scene, cam, trig=O.TRIG_DET, threads=2).render(3).acc
        got = np.load(tmp_path / "tile.npy")
        assert got.view(np.uint32).tolist() == want.view(np.uint32).tolist()  # byte-identical
    else:
        # the same 6 independent samples summed in one process, in rank-major order
        total = np.zeros((H, W, 4), np.float32)
        for rank in range(world):
            first, stride, count = sample_split(6, rank, world)
            c = type(cam).from_buffer_copy(cam)
            c.frame = first
            r = O.OracleRenderer(scene, c, trig=O.TRIG_DET, threads=2, frame_stride=stride, independent_seeds=True)
            total = total + r.render(count).acc
        got = np.load(tmp_path / "sample.npy")
        assert got.view(np.uint32).tolist() == total.view(np.uint32).tolist()
        assert np.all(got[..., 3] == 6.0)  # alpha counts the samples
