#!/usr/bin/env python3
"""Command-line driver: what the reference's `wWinMain` does between scene set-up and the window
(src/main.cpp:800-840 scene, :987-994 camera frame, :1001-1022 one sample per frame per row,
:1024-1034 tone-map, :179-212 pixels.data), with the frame loop replaced by one call that renders
all samples on the GPU.

    python -m pathtracer_amd.render --scene model --model assets/models/rook.triangles --width 600 --spp 64 --out rook.ppm
    python -m pathtracer_amd.render --scene chess32 --width 1920 --height 1080 --spp 256 --out chess.ppm --data pixels.data
    python -m pathtracer_amd.render --scene spheres --width 1200 --spp 128 --out spheres.ppm
"""
import argparse
import sys
import time

import numpy as np

from . import api
from . import ctypes_abi as abi

SCENES = {
    "model": abi.PT_SCENE_MODEL,                  # src/main.cpp:800-840 with --model as the .triangles file
    "spheres": abi.PT_SCENE_RANDOM_SPHERES,       # src/main.cpp:516-625
    "cornell": abi.PT_SCENE_CORNELL_BOX,          # src/main.cpp:627-798
    "chess6": abi.PT_SCENE_CHESS6_ROW,
    "chess32": abi.PT_SCENE_CHESS32,
    "mixed": abi.PT_SCENE_CHESS32_MIXED,
    "grid": abi.PT_SCENE_CHESS_GRID,
}


def render(scene_name, width, height=None, spp=16, max_bounces=50, model=None, grid=2, device=0):
    """Returns (accum f64 (height, width, 4) BGRA+count, bgra8 (height, width, 4), seconds on the GPU)."""
    if scene_name == "model":
        if model is None:
            raise ValueError("--scene model needs --model <file.triangles>")
        host = api.HostScene(SCENES[scene_name], model)
    elif scene_name in ("spheres", "cornell"):
        host = api.HostScene(SCENES[scene_name])
    else:
        host = api.HostScene(SCENES[scene_name], api.MODELS_DIR, grid if scene_name == "grid" else 0)
    if height is None:
        height = int(width / host.aspect_ratio)       # CLIENT_HEIGHT, src/main.cpp:39
    scene = api.DeviceScene.from_host(host, device=device)
    frame = host.frame(width, height, sample_count=spp, max_bounces=max_bounces)
    accum = scene.render(frame)
    seconds = scene.last_render_ms() / 1e3
    image = api.tonemap_bgra8(accum, spp)
    return accum.reshape(height, width, 4), image.reshape(height, width, 4), seconds


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--scene", default="model", choices=sorted(SCENES))
    ap.add_argument("--model", help=".triangles file of --scene model")
    ap.add_argument("--grid", type=int, default=2, help="boards per side of --scene grid")
    ap.add_argument("--width", type=int, default=600)                     # CLIENT_WIDTH, src/main.cpp:38
    ap.add_argument("--height", type=int, help="default: width / the scene's aspect ratio")
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--max-bounces", type=int, default=50)                # MAX_BOUNCE_COUNT, src/path_tracing.cpp:172
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--out", help="binary PPM to write")
    ap.add_argument("--data", help="the reference's pixels.data dump to write (raw RGBA8, rows bottom-up)")
    ap.add_argument("--accum", help=".npy of the f64 BGRA+count sums")
    args = ap.parse_args(argv)
    start = time.perf_counter()
    accum, image, gpu_seconds = render(args.scene, args.width, args.height, args.spp, args.max_bounces, args.model, args.grid, args.device)
    height, width = image.shape[:2]
    if args.out:
        api.write_ppm_file(args.out, image, width, height)
    if args.data:
        api.write_pixel_data_file(args.data, image)
    if args.accum:
        np.save(args.accum, accum)
    samples = width * height * args.spp
    print(f"{args.scene}: {width}x{height}, {args.spp} spp, {samples / 1e6:.1f} Msamples in {gpu_seconds * 1e3:.1f} ms on the GPU "
          f"({samples / gpu_seconds / 1e6:.0f} Msamples/s), {time.perf_counter() - start:.1f} s in all", file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
