#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — counts the reference's algorithmic work per sample for a workload.

Runs the CPU restatement with counters over a bounded sample of the workload and writes
profiles/algorithmic_work.json.  The counts are those of the REFERENCE'S traversal (every
node whose box passes is entered, both children always, src/path_tracing.cpp:141-169), not
of the GPU's pruned search, so the roofline fraction computed from them does not depend on
how the GPU walks the tree:

    bytes/sample = 64 * node visits + 72 * triangle tests + 32 * sphere tests + 32 (accumulator)
    f64 ops/sample = 24 * node visits + 74 * triangle tests + 32 * sphere tests + 150 * bounces

(node: 6 sub + 6 mul + 10 min/max + 2 compares with the three reciprocals hoisted per ray;
triangle: ~71 add/mul + 3 divides; sphere: ~30 + sqrt; ray generation / scatter: ~150.)

    python oracle/count_work.py [--scene chess32] [--width 1920 --height 1080 --spp 2]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.bindings import OracleLib  # noqa: E402
from pathtracer_amd import api, ctypes_abi as abi  # noqa: E402
from pathtracer_amd.ctypes_abi import ptr  # noqa: E402

SCENES = {"chess32": abi.PT_SCENE_CHESS32, "chess6": abi.PT_SCENE_CHESS6_ROW, "mixed": abi.PT_SCENE_CHESS32_MIXED,
          "spheres": abi.PT_SCENE_RANDOM_SPHERES, "cornell": abi.PT_SCENE_CORNELL_BOX}
NAMES = ["samples", "bounces", "triangle_nodes", "triangle_tests", "sphere_nodes", "sphere_tests", "ties", "pruned_mismatch"]


def count(scene_name, width, height, spp, max_bounces, threads, grid=0):
    orc = OracleLib()
    if scene_name == "grid":
        host = api.HostScene(abi.PT_SCENE_CHESS_GRID, api.MODELS_DIR, grid)
    else:
        host = api.HostScene(SCENES[scene_name], api.MODELS_DIR)
    scene = orc.scene_create(host.arrays())
    frame = host.frame(width, height, sample_count=spp, max_bounces=max_bounces)
    accum = np.zeros(4 * width * height)
    raw = np.zeros(72, np.uint64)
    fn = orc.dll.orc_render_counted
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    start = time.time()
    fn(scene.handle, C.byref(frame), ptr(accum), threads, 1, ptr(raw))
    seconds = time.time() - start
    k = dict(zip(NAMES, raw[:8].tolist()))
    n = k["samples"]
    nodes = k["triangle_nodes"] + k["sphere_nodes"]
    out = {
        "scene": scene_name, "width": width, "height": height, "spp_counted": spp, "max_bounces": max_bounces,
        "triangles": int(host.desc.triangle_count),
        "counters": k,
        "per_sample": {
            "traversal_pairs": k["bounces"] / n,
            "node_visits": nodes / n,
            "triangle_tests": k["triangle_tests"] / n,
            "sphere_tests": k["sphere_tests"] / n,
            "bytes": (64 * nodes + 72 * k["triangle_tests"] + 32 * k["sphere_tests"]) / n + 32,
            "f64_ops": (24 * nodes + 74 * k["triangle_tests"] + 32 * k["sphere_tests"] + 150 * k["bounces"]) / n,
        },
        "ended_at_bounce": {str(i): int(v) for i, v in enumerate(raw[8:]) if v},
        "counting_seconds": seconds,
    }
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="chess32")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=2)
    ap.add_argument("--max-bounces", type=int, default=50)
    ap.add_argument("--grid", type=int, default=0)
    ap.add_argument("--threads", type=int, default=min(16, os.cpu_count() or 1))
    args = ap.parse_args()
    result = count(args.scene, args.width, args.height, args.spp, args.max_bounces, args.threads, args.grid)
    path = os.path.join(ROOT, "profiles", "algorithmic_work.json")
    table = json.load(open(path)) if os.path.exists(path) else {}
    key = f"{args.scene}{args.grid or ''}_{args.width}x{args.height}_depth{args.max_bounces}"
    table[key] = result
    json.dump(table, open(path, "w"), indent=1, sort_keys=True)
    print(key, json.dumps(result["per_sample"]), "ties", result["counters"]["ties"], "pruned_mismatch", result["counters"]["pruned_mismatch"])


if __name__ == "__main__":
    main()
