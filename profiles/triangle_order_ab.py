#!/usr/bin/env python3
"""Does the order of the triangle records in HBM matter?  The same scene with its triangles in the caller's order, sorted
along a Morton curve of their centroids (what the order of the search hierarchy's leaves comes to) and shuffled; the search
hierarchy is built over boxes, so it is the same tree up to the numbering of its leaves, and only where a record lies changes.
usage: python profiles/triangle_order_ab.py [scene] [spp]        (scene: grid5 | chess32)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pathtracer_amd import api, ctypes_abi as abi

name = sys.argv[1] if len(sys.argv) > 1 else "grid5"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
host = api.HostScene(abi.PT_SCENE_CHESS_GRID, api.MODELS_DIR, int(name[4:])) if name.startswith("grid") else api.HostScene(abi.PT_SCENE_CHESS32, api.MODELS_DIR)
arrays = host.arrays()
frame = host.frame(1920, 1080, sample_count=spp, max_bounces=50)


def morton_order(triangles):
    centre = triangles.reshape(-1, 3, 3).mean(1)
    lo, hi = centre.min(0), centre.max(0)
    cell = np.minimum(((centre - lo) / (hi - lo) * (1 << 21)).astype(np.uint64), (1 << 21) - 1)

    def spread(v):                       # 21 bits -> every third bit of 63
        v = (v | (v << 32)) & 0x1F00000000FFFF
        v = (v | (v << 16)) & 0x1F0000FF0000FF
        v = (v | (v << 8)) & 0x100F00F00F00F00F
        v = (v | (v << 4)) & 0x10C30C30C30C30C3
        v = (v | (v << 2)) & 0x1249249249249249
        return v
    return np.argsort(spread(cell[:, 0]) | (spread(cell[:, 1]) << 1) | (spread(cell[:, 2]) << 2), kind="stable")


def device_scene(order):
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    keep = {k: np.ascontiguousarray(v) for k, v in arrays.items()}
    if order is not None:
        keep["triangles"] = np.ascontiguousarray(arrays["triangles"][order])
        keep["triangle_material_indices"] = np.ascontiguousarray(arrays["triangle_material_indices"][order])
        keep["triangle_bvh"] = api.construct_triangle_bvh(keep["triangles"])
    d = abi.PtSceneDesc()
    d.materials, d.material_count = ptr(keep["materials"]), len(keep["materials"])
    d.spheres, d.sphere_count = ptr(keep["spheres"]), len(keep["spheres"])
    d.sphere_material_indices = ptr(keep["sphere_material_indices"])
    d.sphere_bvh, d.sphere_bvh_node_count = ptr(keep["sphere_bvh"]), len(keep["sphere_bvh"])
    d.triangles, d.triangle_count = ptr(keep["triangles"]), len(keep["triangles"])
    d.triangle_material_indices = ptr(keep["triangle_material_indices"])
    d.triangle_bvh, d.triangle_bvh_node_count = ptr(keep["triangle_bvh"]), len(keep["triangle_bvh"])
    d.background_gradient_start = abi.PtColour(*arrays["background"][0])
    d.background_gradient_end = abi.PtColour(*arrays["background"][1])
    return api.DeviceScene(d, keepalive=keep)


count = len(arrays["triangles"])
buf = torch.zeros(4 * 1920 * 1080, dtype=torch.float64, device="cuda:0")
sums = {}
for label, order in (("caller's order", None), ("Morton order of centroids", morton_order(arrays["triangles"])), ("shuffled", np.random.default_rng(1).permutation(count))):
    device = device_scene(order)
    times = []
    for _ in range(3):
        buf.zero_()
        device.render_device(frame, buf.data_ptr())
        torch.cuda.synchronize()
        times.append(round(device.last_render_ms(), 2))
    sums[label] = float(buf.view(-1, 4)[:, :3].sum())
    print(f"{name} 1920x1080 {spp} spp, {count} triangles, {label}: ms {times}", flush=True)
    del device
print("image sums (the pictures agree up to the tie rule):", sums)
