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
