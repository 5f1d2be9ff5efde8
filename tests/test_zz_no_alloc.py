"""pt_render_device_noalloc: the render entry point that allocates nothing (its per-path colour records and
work counter live in a workspace inside the library's code object) against pt_render_device, through the C
ABI.  Bar: the same bytes.  Two calls in a row, the second with a smaller input, because every call on a GPU
shares that GPU's workspace and it keeps the previous call's contents.
"""
import numpy as np
import pytest

from conftest import MODELS
from pathtracer_amd import ctypes_abi as abi

pytestmark = pytest.mark.gpu


def render_both(device, frame):
    import torch
    expect = torch.zeros((frame.height, frame.width, 4), dtype=torch.float64, device="cuda:0")
    device.render_device(frame, expect.data_ptr())
    torch.cuda.synchronize()
    got = torch.zeros_like(expect)
    device.render_device_noalloc(frame, got.data_ptr())
    torch.cuda.synchronize()
    return got, expect


def test_noalloc_matches_render_device_twice_in_a_row(api, orc):
    import torch
    host = api.HostScene(abi.PT_SCENE_CHESS32_MIXED, MODELS)
    device = api.DeviceScene.from_host(host)
    # 40 samples: three passes of 16 through the workspace; rows 5..301 of 3 shards: partial tiles, partial blocks
    big = host.frame(640, 360, sample_count=40, max_bounces=16, row_begin=5, row_end=301, shard_index=1, shard_count=3)
    got, expect = render_both(device, big)
    assert torch.equal(got.view(torch.int64), expect.view(torch.int64))
    assert bool((expect[..., 3] != 0).any())
    # a smaller frame next: the workspace still holds the first call's records
    small = host.frame(97, 61, sample_count=3, max_bounces=50)
    got, expect = render_both(device, small)
    assert torch.equal(got.view(torch.int64), expect.view(torch.int64))
    checker = orc.scene_create(host.arrays())
    want = checker.render(small, threads=8)
    assert np.array_equal(got.cpu().numpy().ravel().view(np.uint64), want.view(np.uint64))
    # adding into a buffer that already holds samples, through another scene handle on the same GPU
    other = api.DeviceScene.from_host(api.HostScene(abi.PT_SCENE_MODEL, MODELS + "/rook.triangles"))
    frame = api.HostScene(abi.PT_SCENE_MODEL, MODELS + "/rook.triangles").frame(128, 128, sample_count=2)
    got, expect = render_both(other, frame)
    more = api.HostScene(abi.PT_SCENE_MODEL, MODELS + "/rook.triangles").frame(128, 128, sample_begin=2, sample_count=2)
    other.render_device(more, expect.data_ptr())
    other.render_device_noalloc(more, got.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int64), expect.view(torch.int64))
    assert bool((got[..., 3] == 4.0).all())


def test_noalloc_rejects_frames_beyond_its_workspace(api):
    import torch
    host = api.HostScene(abi.PT_SCENE_MODEL, MODELS + "/rook.triangles")
    device = api.DeviceScene.from_host(host)
    buf = torch.zeros((1080, 1920, 4), dtype=torch.float64, device="cuda:0")
    with pytest.raises(api.PathTracerError) as e:
        device.render_device_noalloc(host.frame(1920, 1080, sample_count=1), buf.data_ptr())
    assert e.value.code == abi.PT_ERR_UNSUPPORTED and "8192" in str(e.value)
    torch.cuda.synchronize()
    assert bool((buf == 0).all())
    # the same frame in row ranges that fit: 1920 x 272 rows = 8160 tiles
    expect = torch.zeros_like(buf)
    device.render_device(host.frame(1920, 1080, sample_count=1), expect.data_ptr())
    for begin in range(0, 1080, 272):
        device.render_device_noalloc(host.frame(1920, 1080, sample_count=1, row_begin=begin, row_end=min(begin + 272, 1080)), buf.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(buf.view(torch.int64), expect.view(torch.int64))


def test_noalloc_under_graph_capture(api):
    """The entry point on a stream that is being captured into a HIP graph: nothing is allocated, no event is recorded or
    queried, so the capture stays valid; two replays into a cleared buffer give the bytes of a plain launch each."""
    import torch
    host = api.HostScene(abi.PT_SCENE_CHESS6_ROW, MODELS)
    device = api.DeviceScene.from_host(host)
    frame = host.frame(320, 180, sample_count=20, max_bounces=50)
    expect = torch.zeros((180, 320, 4), dtype=torch.float64, device="cuda:0")
    device.render_device(frame, expect.data_ptr())
    got = torch.zeros_like(expect)
    device.render_device_noalloc(frame, got.data_ptr())     # what a handle sets up once (the occupancy query) happens here, outside the capture
    torch.cuda.synchronize()
    got.zero_()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.graph(graph, stream=side):
        device.render_device_noalloc(frame, got.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert bool((got == 0).all())                          # capturing runs nothing
    for _ in range(2):
        got.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(got.view(torch.int64), expect.view(torch.int64))
