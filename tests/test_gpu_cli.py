"""The command-line driver renders through the library and writes the reference's output formats
(src/main.cpp:179-212 pixels.data; the PPM is the same pixels, RGB, rows top-down)."""
import os

import numpy as np
import pytest

from conftest import same_bits
from pathtracer_amd import ctypes_abi as abi

pytestmark = pytest.mark.gpu

MODELS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets", "models")


def test_render_command_writes_the_checkers_image(api, orc, tmp_path):
    from pathtracer_amd import render
    ppm, data, npy = tmp_path / "rook.ppm", tmp_path / "pixels.data", tmp_path / "accum.npy"
    model = os.path.join(MODELS, "rook.triangles")
    assert render.main(["--scene", "model", "--model", model, "--width", "60", "--spp", "3", "--max-bounces", "8",
                        "--out", str(ppm), "--data", str(data), "--accum", str(npy)]) == 0
    host = api.HostScene(abi.PT_SCENE_MODEL, model)
    height = int(60 / host.aspect_ratio)
    frame = host.frame(60, height, sample_count=3, max_bounces=8)
    expected = orc.scene_create(host.arrays()).render(frame, threads=4)
    accum = np.load(npy)
    assert accum.shape == (height, 60, 4) and same_bits(accum.ravel(), expected)
    image = api.tonemap_bgra8(expected, 3).reshape(height, 60, 4)
    header = b"P6\n60 %d\n255\n" % height
    raw = ppm.read_bytes()
    assert raw.startswith(header)
    rgb = np.frombuffer(raw[len(header):], np.uint8).reshape(height, 60, 3)
    assert np.array_equal(rgb, image[::-1, :, 2::-1])                    # rows top-down, B,G,R -> R,G,B
    dumped = np.frombuffer(data.read_bytes(), np.uint8).reshape(height, 60, 4)
    assert np.array_equal(dumped[..., [2, 1, 0, 3]], image)              # RGBA on disk, rows bottom-up


def test_render_command_needs_a_model_for_the_model_scene(api):
    from pathtracer_amd import render
    with pytest.raises(ValueError):
        render.render("model", 32)


def test_a_c_program_built_against_the_header_renders_the_reference_frame(golden, tmp_path):
    """tests/consumer/render_rook.c is plain C compiled against include/pathtracer_amd.h and linked with the
    library (built by `make`): it sets the reference's model scene up from the reference's constants in its
    own arrays, renders 64x48x4 through pt_render in a fresh process, and must write the buffer the
    reference's own code made."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    program = os.path.join(root, "tests", "consumer", "render_rook")
    assert os.path.exists(program), "run `make` (or __graft_entry__.build()) first"
    out = tmp_path / "rook.f64"
    done = subprocess.run([program, os.path.join(MODELS, "rook.triangles"), "64", "48", "4", str(out)],
                          capture_output=True, text=True, timeout=300)
    assert done.returncode == 0, done.stderr
    accum = np.fromfile(out, np.float64)
    assert same_bits(accum, golden["rook_accum4"].ravel())


def test_launches_through_one_handle_from_several_streams_run_one_after_another(api):
    """A scene handle owns one work counter and one colour workspace: a launch on another stream is put behind the
    end of the handle's previous launch instead of running beside it (it would share the counter), and the null
    stream's entry points (pt_primary_hits, pt_render) may follow an asynchronous launch at once."""
    import torch
    host = api.HostScene(abi.PT_SCENE_CHESS32, MODELS)
    device = api.DeviceScene.from_host(host)
    frame = host.frame(1920, 1080, sample_count=64)        # some tens of milliseconds of work
    a = torch.zeros((1080, 1920, 4), dtype=torch.float64, device="cuda:0")
    b = torch.zeros_like(a)
    c = torch.zeros_like(a)
    torch.cuda.synchronize()
    first, second = torch.cuda.Stream(), torch.cuda.Stream()
    device.render_device(frame, a.data_ptr(), first.cuda_stream)
    device.render_device(frame, b.data_ptr(), second.cuda_stream)     # while the first still runs: ordered behind it
    device.render_device(frame, c.data_ptr(), first.cuda_stream)
    small = host.frame(96, 54, sample_count=1)
    tri, _ = device.primary_hits(small)                               # null stream, right behind three asynchronous launches
    torch.cuda.synchronize()
    assert torch.equal(a.view(torch.int64), b.view(torch.int64)) and torch.equal(a.view(torch.int64), c.view(torch.int64))
    assert bool((a[..., 3] == 64.0).all()) and (tri >= 0).any()
    blocks, units, held = device.last_launch_shape()
    assert 1 <= blocks <= 8 and units > 0 and held >= 1920 * 1080 * 16 * 32


def test_workspace_limit_changes_batches_not_bytes(api):
    """pt_scene_set_workspace_limit: the cap of what a handle keeps for its per-path colour records.  A cap below one
    launch's need only means more sample batches; a lowered cap gives the memory back."""
    import torch
    host = api.HostScene(abi.PT_SCENE_CHESS6_ROW, MODELS)
    device = api.DeviceScene.from_host(host)
    frame = host.frame(640, 360, sample_count=80)
    group_bytes = 80 * 45 * 1024 * 32                      # 16 samples of the 80 x 45 tiles
    whole = torch.zeros((360, 640, 4), dtype=torch.float64, device="cuda:0")
    device.render_device(frame, whole.data_ptr())
    torch.cuda.synchronize()
    assert device.last_launch_shape()[2] == 5 * group_bytes
    device.set_workspace_limit(2 * group_bytes + 5)
    assert device.last_launch_shape()[2] == 0              # what it held beyond the new cap is released
    batched = torch.zeros_like(whole)
    device.render_device(frame, batched.data_ptr())
    torch.cuda.synchronize()
    assert device.last_launch_shape()[2] == 2 * group_bytes
    assert torch.equal(whole.view(torch.int64), batched.view(torch.int64))
    device.set_workspace_limit(0)                          # one group is the least a launch takes
    batched.zero_()
    device.render_device(frame, batched.data_ptr())
    torch.cuda.synchronize()
    assert device.last_launch_shape()[2] == group_bytes and torch.equal(whole.view(torch.int64), batched.view(torch.int64))


def test_bench_script_with_two_ranks_on_one_gpu(tmp_path):
    """bench.py as the driver launches it for N = 2 (torch.distributed.run, one process per rank), rehearsed on
    one GPU: both ranks render their row blocks on cuda:0 and the gather goes through gloo.  What is checked is
    the script's N-rank path: sharding by rank, the gather to rank 0, the barrier-bracketed timing and the one
    JSON line, whose frame must equal the CPU checker's on the verified rows."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PT_DIST_BACKEND="gloo")
    done = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                           "--spp", "16", "--width", "640", "--height", "360"],
                          capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert done.returncode == 0, done.stderr[-3000:]
    lines = [line for line in done.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["verified"] is True
    assert line["value"] > 0 and line["unit"] == "Msamples/s" and line["scaling"] == "strong"
    assert line["roofline"]["paths_per_launch"] == 640 * 360 * 16 / 2
    # every rank's own kernel and gather times are in the line (what shows imbalance once the ranks have a GPU each)
    ranks = line["per_rank"]
    assert len(ranks["kernel_ms"]) == 2 and len(ranks["gather_ms"]) == 2 and min(ranks["kernel_ms"]) > 0
    assert ranks["kernel_ms_min"] == min(ranks["kernel_ms"]) and ranks["kernel_ms_max"] == max(ranks["kernel_ms"]) and ranks["gather_ms_max"] > 0
    assert line["roofline"]["occupancy"]["waves_per_simd"] >= 1


def test_viewer_frames_accumulate_and_restart_like_the_reference_loop(api, orc):
    """pt_viewer_frame = one iteration of the reference's frame loop (src/main.cpp:890-1036): frames with an unmoved
    camera add one sample each; a frame whose input moves the camera zeroes the buffer and starts again at sample 0;
    several samples per call give the bytes of as many frames."""
    import torch
    host = api.HostScene(abi.PT_SCENE_MODEL, os.path.join(MODELS, "rook.triangles"))
    device = api.DeviceScene.from_host(host)
    checker = orc.scene_create(host.arrays())
    w, h = 120, 96
    camera = abi.PtCamera.from_buffer_copy(bytes(host.camera))
    accum = torch.full((h, w, 4), 7.0, dtype=torch.float64, device="cuda:0")      # stale contents: the first frame must not clear them ...
    accum.zero_()                                                                 # ... the caller does, as the reference allocates zeroed memory
    sample = 0
    for _ in range(3):
        sample = device.viewer_frame(camera, None, host.aspect_ratio, w, h, sample, accum.data_ptr())
    torch.cuda.synchronize()
    assert sample == 3
    assert same_bits(accum.cpu().numpy().ravel(), checker.render(host.frame(w, h, sample_count=3), threads=8))
    # the mouse is dragged and the wheel turned: new camera, accumulation restarts
    drag = abi.PtViewerInput(1, 14, -9, 2, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0.016)
    sample = device.viewer_frame(camera, drag, host.aspect_ratio, w, h, sample, accum.data_ptr())
    idle = abi.PtViewerInput(0, 5, 5, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.016)     # pointer moves without the button: nothing changes
    sample = device.viewer_frame(camera, idle, host.aspect_ratio, w, h, sample, accum.data_ptr(), samples_per_frame=4)
    torch.cuda.synchronize()
    assert sample == 5
    moved = abi.PtCamera.from_buffer_copy(bytes(host.camera))
    assert orc.camera_apply_input(moved, drag) and bytes(moved) == bytes(camera) != bytes(host.camera)
    frame = api.camera_frame(moved, host.aspect_ratio, w, h, sample_count=5)
    assert same_bits(accum.cpu().numpy().ravel(), checker.render(frame, threads=8))
    image = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda:0")
    api.tonemap_device(0, accum.data_ptr(), w * h, sample, image.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(image.cpu().numpy().ravel(), orc.tonemap_bgra8(checker.render(frame, threads=8), 5))
