#!/usr/bin/env python3
"""TEST INFRASTRUCTURE - writes tests/golden/benchmark_scenes.npz: the scenes the benchmark renders (the 32-piece chess set, the
same with every material, 4 boards of it) rendered by the reference's own code (oracle/_ref/libptref.so, the reference's
sources compiled in place).  The scene arrays come from this library's host-side scene builder (the reference defines no such
scene); the trees over them, the traversal and the bounce loop are the reference's.  Run from the repo root in a container
that has /root/reference:

    make -C oracle ref && python oracle/gen_benchmark_golden.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.bindings import RefLib  # noqa: E402
from pathtracer_amd import api, ctypes_abi as abi  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "benchmark_scenes.npz")
ROWS_1080P = (137, 540, 803)          # rows of the 1920x1080 frame rendered at 16 samples per pixel (chess32 and mixed)
# name, scene kind, parameter, samples per pixel of the 160x90 frame
SCENES = (("chess32", abi.PT_SCENE_CHESS32, 0, 4), ("mixed", abi.PT_SCENE_CHESS32_MIXED, 0, 4), ("grid2", abi.PT_SCENE_CHESS_GRID, 2, 2))


def sha(array):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(array).tobytes()).digest(), np.uint8)


def main():
    ref = RefLib()
    g = {}
    for name, kind, param, spp in SCENES:
        host = api.HostScene(kind, api.MODELS_DIR, param)
        arrays = host.arrays()
        scene = ref.scene_create(arrays)                 # the reference's own construct_*_bvh over these primitives
        g[f"{name}_triangles_sha"] = sha(arrays["triangles"])
        g[f"{name}_camera"] = np.frombuffer(bytes(host.camera), np.float64)
        frame = host.frame(160, 90, sample_count=spp, max_bounces=50)       # 50: the reference's compile-time depth
        tri, sphere = scene.primary_hits(frame)
        g[f"{name}_tri_hits"], g[f"{name}_sphere_hits"] = tri, sphere
        g[f"{name}_accum"] = scene.render(frame, threads=8)
        if name != "grid2":
            for row in ROWS_1080P:
                big = host.frame(1920, 1080, sample_count=16, max_bounces=50, row_begin=row, row_end=row + 1)
                accum = scene.render(big, threads=8).reshape(1080, 1920, 4)[row]
                g[f"{name}_1080p_row{row}_sha"] = sha(accum)
                g[f"{name}_1080p_row{row}_sums"] = accum.sum(0)
        print(name, len(arrays["triangles"]), "triangles;", int((tri >= 0).sum()), "triangle hits of", tri.size, flush=True)
    np.savez_compressed(OUT, **g)
    print(f"wrote {OUT}: {os.path.getsize(OUT) / 1024:.0f} KiB, {len(g)} arrays")


if __name__ == "__main__":
    main()
