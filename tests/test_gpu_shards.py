"""The sharded (multi-GPU) path of the HIP kernels, on one GPU.

A rank of an N-GPU job renders `frame.shard_index = rank, frame.shard_count = N`: row blocks of
PT_ROW_BLOCK dealt round-robin, the reference's "one writer per row" (src/main.cpp:1001-1020).
Here every shard of a frame is rendered by the one GPU at hand, and the device-side packing that
the gather uses is run without a process group.  Bar: bit-exact, as everywhere.
"""
import os

import numpy as np
import pytest

from conftest import MODELS, same_bits
from pathtracer_amd import ctypes_abi as abi
from pathtracer_amd.ctypes_abi import PT_ROW_BLOCK

pytestmark = pytest.mark.gpu

SWITCHES = ("PT_KERNEL", "PT_WORKSPACE_GB")


@pytest.fixture
def switches():
    saved = {k: os.environ.pop(k, None) for k in SWITCHES}

    def use(**values):
        for k in SWITCHES:
            os.environ.pop(k, None)
        for k, v in values.items():
            os.environ[k] = str(v)

    yield use
    for k in SWITCHES:
        os.environ.pop(k, None)
    for k, v in saved.items():
        if v is not None:
            os.environ[k] = v


@pytest.fixture(scope="module")
def scenes(api, orc):
    cache = {}

    def get(name):
        if name not in cache:
            kind = {"chess32": abi.PT_SCENE_CHESS32, "mixed": abi.PT_SCENE_CHESS32_MIXED}[name]
            host = api.HostScene(kind, MODELS)
            cache[name] = (host, api.DeviceScene.from_host(host), orc.scene_create(host.arrays()))
        return cache[name]

    return get


def owner_of_rows(height, count, row_begin=0):
    """Shard that owns each row (-1 below row_begin)."""
    rows = np.arange(height)
    return np.where(rows >= row_begin, ((rows - row_begin) // PT_ROW_BLOCK) % count, -1)


def render_on_device(device, frame):
    """pt_render_device into a fresh zeroed buffer in HBM; returns the torch tensor (synchronised)."""
    import torch
    buf = torch.zeros((frame.height, frame.width, 4), dtype=torch.float64, device="cuda:0")
    device.render_device(frame, buf.data_ptr())
    torch.cuda.synchronize()
    return buf


@pytest.mark.parametrize("name,bounces", [("chess32", 50), ("mixed", 16)])
@pytest.mark.parametrize("count", [2, 3, 8])
def test_shards_assemble_the_unsharded_frame(scenes, name, bounces, count):
    """Every shard of a frame rendered into ONE device buffer gives the bytes of the unsharded render and
    of the CPU checker; a shard alone touches exactly the rows it owns."""
    import torch
    host, device, checker = scenes(name)
    w, h, spp = 200, 117, 5
    whole = host.frame(w, h, sample_count=spp, max_bounces=bounces)
    expect = checker.render(whole, threads=8).reshape(h, w, 4)
    assert same_bits(render_on_device(device, whole).cpu().numpy(), expect)
    owner = owner_of_rows(h, count)
    shared = torch.zeros((h, w, 4), dtype=torch.float64, device="cuda:0")
    for index in range(count):
        frame = host.frame(w, h, sample_count=spp, max_bounces=bounces, shard_index=index, shard_count=count)
        alone = render_on_device(device, frame).cpu().numpy()
        touched = (alone[..., 3] != 0).any(axis=1)
        assert np.array_equal(touched, owner == index)
        assert same_bits(alone[owner == index], expect[owner == index])
        device.render_device(frame, shared.data_ptr())
    torch.cuda.synchronize()
    assert same_bits(shared.cpu().numpy(), expect)


@pytest.mark.parametrize("kernel", ["wave", "pixels"])
def test_shards_of_a_row_range_that_starts_inside_a_block(scenes, switches, kernel):
    """Row blocks are counted from row_begin, not from row 0: a range that starts at row 13 deals rows
    13..20 to shard 0, 21..28 to shard 1, ...; both kernels agree with the checker, which honours the same fields."""
    import torch
    host, device, checker = scenes("chess32")
    w, h, spp, begin, end, count = 136, 99, 3, 13, 94, 3
    switches(PT_KERNEL=kernel)
    owner = owner_of_rows(h, count, begin)
    owner[end:] = -1
    shared = torch.zeros((h, w, 4), dtype=torch.float64, device="cuda:0")
    whole = checker.render(host.frame(w, h, sample_count=spp, row_begin=begin, row_end=end), threads=8).reshape(h, w, 4)
    for index in range(count):
        frame = host.frame(w, h, sample_count=spp, row_begin=begin, row_end=end, shard_index=index, shard_count=count)
        alone = render_on_device(device, frame).cpu().numpy()
        assert np.array_equal((alone[..., 3] != 0).any(axis=1), owner == index)
        assert same_bits(alone, checker.render(frame, threads=8).reshape(h, w, 4))
        device.render_device(frame, shared.data_ptr())
    torch.cuda.synchronize()
    assert same_bits(shared.cpu().numpy(), whole)
    assert (shared[:begin] == 0).all() and (shared[end:] == 0).all()


@pytest.mark.parametrize("world,height", [(2, 117), (8, 1080), (3, 45)])
def test_pack_and_unpack_rows_on_the_device(scenes, world, height):
    """The gather's device-side halves without a process group: each 'rank' renders its shard into a buffer of
    its own, packs its rows (index_select on the GPU), and 'rank 0' unpacks them (index_copy_): the result is
    the unsharded frame.  What travels between the two halves in a real job is dist.gather's business."""
    import torch
    from pathtracer_amd.distributed import pack_rows, shard_rows, unpack_rows
    host, device, _ = scenes("chess32")
    w = 64 if height > 200 else 120
    whole = render_on_device(device, host.frame(w, height, sample_count=2))
    rows_of = [shard_rows(height, r, world).to("cuda:0") for r in range(world)]
    longest = max(len(r) for r in rows_of)
    packed = []
    for r in range(world):
        mine = render_on_device(device, host.frame(w, height, sample_count=2, shard_index=r, shard_count=world))
        send = pack_rows(mine, rows_of[r], longest)
        assert send.shape == (longest, w, 4) and send.is_cuda
        assert (send[len(rows_of[r]):] == 0).all()
        packed.append(send)
        if r == 0:
            assembled = mine
    for r in range(1, world):
        unpack_rows(assembled, rows_of[r], packed[r])
    torch.cuda.synchronize()
    assert torch.equal(assembled.view(torch.int64), whole.view(torch.int64))


def test_4k_frame_in_batches_and_shards(scenes, switches):
    """3840x2160 chess set (the 8-GPU configuration's frame): a colour scratch capped below two sample groups
    makes the 40 samples run as three batches; the bytes equal two launches over split sample ranges and,
    for three rows, the CPU checker's; the same frame rendered as 8 shards into one buffer equals it too."""
    import torch
    host, device, checker = scenes("chess32")
    w, h, spp = 3840, 2160, 40
    switches(PT_WORKSPACE_GB="5")            # one 16-sample group of this frame needs 4.25 GB
    once = render_on_device(device, host.frame(w, h, sample_count=spp))
    split = torch.zeros_like(once)
    device.render_device(host.frame(w, h, sample_count=23), split.data_ptr())
    device.render_device(host.frame(w, h, sample_begin=23, sample_count=spp - 23), split.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(once.view(torch.int64), split.view(torch.int64))
    del split
    assert bool((once[..., 3] == float(spp)).all()) and bool(torch.isfinite(once).all())
    for row in (0, 1003, 2159):
        expect = checker.render(host.frame(w, h, sample_count=spp, row_begin=row, row_end=row + 1), threads=16)
        assert same_bits(once[row].cpu().numpy(), expect.reshape(h, w, 4)[row])
    switches()
    sharded = torch.zeros_like(once)
    for index in range(8):
        device.render_device(host.frame(w, h, sample_count=spp, shard_index=index, shard_count=8), sharded.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(once.view(torch.int64), sharded.view(torch.int64))


def test_one_shard_of_the_8_gpu_configuration_in_full(scenes):
    """What one of 8 GPUs renders of the 3840x2160, 1024-samples-per-pixel frame (its row blocks, every sample: 64 sample
    groups through the workspace), on three of its rows against the CPU checker; rows of other shards stay untouched."""
    import torch
    host, device, checker = scenes("chess32")
    w, h, spp, index, count = 3840, 2160, 1024, 5, 8
    image = render_on_device(device, host.frame(w, h, sample_count=spp, shard_index=index, shard_count=count))
    owner = owner_of_rows(h, count)
    mine = torch.from_numpy(owner == index).to("cuda:0")
    assert bool((image[mine][..., 3] == float(spp)).all()) and bool((image[~mine] == 0).all()) and bool(torch.isfinite(image).all())
    for row in (40, 1128, 2152):
        assert owner[row] == index
        expect = checker.render(host.frame(w, h, sample_count=spp, row_begin=row, row_end=row + 1), threads=16)
        assert same_bits(image[row].cpu().numpy(), expect.reshape(h, w, 4)[row])


def test_benchmark_frame_at_256_samples_against_the_checker(scenes):
    """The frame bench.py times (1920x1080, 256 samples per pixel: 16 sample groups dealt across draws) on
    three rows against the CPU checker, every double; count channel and finiteness on the whole frame."""
    import torch
    host, device, checker = scenes("chess32")
    w, h, spp = 1920, 1080, 256
    image = render_on_device(device, host.frame(w, h, sample_count=spp))
    assert bool((image[..., 3] == float(spp)).all()) and bool(torch.isfinite(image).all())
    for row in (3, 540, 1072):
        expect = checker.render(host.frame(w, h, sample_count=spp, row_begin=row, row_end=row + 1), threads=16)
        assert same_bits(image[row].cpu().numpy(), expect.reshape(h, w, 4)[row])


def test_two_host_threads_render_through_two_handles_at_once(api, orc):
    """Distinct handles are independent: two host threads, each with a scene handle and a stream of its own, create their scenes
    and render different frames at the same time (ctypes drops the interpreter lock during the calls); each gets the bytes a
    lone render gives."""
    import threading
    import torch
    host = api.HostScene(abi.PT_SCENE_CHESS32, MODELS)
    frames = [host.frame(320, 200, sample_count=24), host.frame(256, 144, sample_count=40, max_bounces=12)]
    checker = orc.scene_create(host.arrays())
    expect = [checker.render(f, threads=8).reshape(f.height, f.width, 4) for f in frames]
    results, errors = [None, None], []

    def work(i):
        try:
            device = api.DeviceScene.from_host(host)
            stream = torch.cuda.Stream()
            for _ in range(3):                       # several launches each, so that they do overlap
                buf = torch.zeros((frames[i].height, frames[i].width, 4), dtype=torch.float64, device="cuda:0")
                device.render_device(frames[i], buf.data_ptr(), stream.cuda_stream)
                stream.synchronize()
            results[i] = buf.cpu().numpy()
        except Exception as e:                       # noqa: BLE001 - reported below
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(2):
        assert same_bits(results[i], expect[i])
