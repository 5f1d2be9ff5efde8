"""The N>1 path on CPU: row-block sharding (PtFrame.shard_index / shard_count) and the gather of
the accumulation buffer, with world_size 2 and 3 over gloo.  The CPU checker stands in for the
GPU renderer (it honours the same shard fields), so what is tested is the distributed plumbing:
each rank renders only its row blocks, rank 0 ends up with the bytes of a single-process render.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import MODELS, ROOT
from pathtracer_amd.ctypes_abi import PT_ROW_BLOCK


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, height, width, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.bindings import OracleLib
        from pathtracer_amd.distributed import gather_rows, shard_rows
        orc = OracleLib()
        scene, cam, aspect = orc.model_scene(os.path.join(MODELS, "rook.triangles"))
        frame = orc.camera_frame(cam, aspect, width, height)
        frame.sample_count = 2
        frame.shard_index, frame.shard_count = rank, world
        accum = torch.from_numpy(scene.render(frame)).reshape(height, width, 4)
        # a rank touches its own row blocks only
        mine = set(shard_rows(height, rank, world).tolist())
        for row in range(height):
            touched = bool((accum[row, :, 3] != 0).any())
            assert touched == (row in mine), (rank, row)
        from pathtracer_amd.distributed import gather_image8
        image = gather_image8(accum.clone(), 2, height, width, rank, world)
        gather_rows(accum, height, width, rank, world)
        if rank == 0:
            np.save(out_path, accum.numpy())
            np.save(out_path + ".image8.npy", image.numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,height", [(2, 48), (3, 45)])
def test_sharded_render_and_gather_equals_single_process(tmp_path, orc, world, height):
    width = 40
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(world, _free_port(), height, width, out), nprocs=world, join=True)
    gathered = np.load(out)
    scene, cam, aspect = orc.model_scene(os.path.join(MODELS, "rook.triangles"))
    frame = orc.camera_frame(cam, aspect, width, height)
    frame.sample_count = 2
    single = scene.render(frame).reshape(height, width, 4)
    assert gathered.tobytes() == single.tobytes()
    # the 8-bit gather (tone-map per rank, 4 B per pixel on the wire) gives the tone-map of the whole buffer
    image = np.load(out + ".image8.npy")
    assert image.tobytes() == orc.tonemap_bgra8(single, 2).tobytes()


def test_shard_rows_partition_the_image():
    from pathtracer_amd.distributed import shard_rows
    for height, world in [(1080, 8), (45, 3), (7, 4), (8, 8)]:
        rows = [shard_rows(height, r, world) for r in range(world)]
        merged = torch.cat(rows).sort().values
        assert torch.equal(merged, torch.arange(height))
        for r, mine in enumerate(rows):
            assert all(((int(x) // PT_ROW_BLOCK) % world) == r for x in mine)
