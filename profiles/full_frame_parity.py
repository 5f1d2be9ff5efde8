#!/usr/bin/env python3
"""One-off: a whole benchmark frame, every pixel at the full sample count, against the CPU checker (the test suite compares whole
frames at 16 samples per pixel and three rows at 256).
usage: python profiles/full_frame_parity.py <scene> <WxH> <spp> <bounces> [threads]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pathtracer_amd import api, ctypes_abi as abi
from oracle.bindings import OracleLib

name, size, spp, bounces = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
threads = int(sys.argv[5]) if len(sys.argv) > 5 else 16
w, h = (int(v) for v in size.split("x"))
if name.startswith("grid"):
    host = api.HostScene(abi.PT_SCENE_CHESS_GRID, api.MODELS_DIR, int(name[4:]))
else:
    host = api.HostScene({"chess32": abi.PT_SCENE_CHESS32, "mixed": abi.PT_SCENE_CHESS32_MIXED, "spheres": abi.PT_SCENE_RANDOM_SPHERES,
                          "cornell": abi.PT_SCENE_CORNELL_BOX}[name], api.MODELS_DIR)
device = api.DeviceScene.from_host(host)
shard = [int(v) for v in os.environ.get("PT_SHARD", "0/1").split("/")]          # PT_SHARD=5/8: the rows one of 8 GPUs renders
frame = host.frame(w, h, sample_count=spp, max_bounces=bounces, shard_index=shard[0], shard_count=shard[1])
buf = torch.zeros(4 * w * h, dtype=torch.float64, device="cuda:0")
device.render_device(frame, buf.data_ptr())
torch.cuda.synchronize()
got = buf.cpu().numpy()
print(f"{name} {w}x{h} {spp} spp depth {bounces} shard {shard[0]} of {shard[1]}: GPU {device.last_render_ms():.1f} ms; checker on {threads} threads ...", flush=True)
checker = OracleLib().scene_create(host.arrays())
expect = np.zeros(4 * w * h)
t0 = time.time()
rows_per_piece = abi.PT_ROW_BLOCK * shard[1] * 2 if shard[1] > 1 else 40      # in pieces, so that the run keeps printing; a piece starts
for row in range(0, h, rows_per_piece):                                         # where the dealing of row blocks starts over
    checker.render(host.frame(w, h, sample_count=spp, max_bounces=bounces, row_begin=row, row_end=min(h, row + rows_per_piece),
                              shard_index=shard[0], shard_count=shard[1]), expect, threads=threads)
    print(f"  rows {row}..{min(h, row + rows_per_piece) - 1} done, {time.time() - t0:.0f} s", flush=True)
nan = np.isnan(got)
same = np.array_equal(nan, np.isnan(expect)) and np.array_equal(got[~nan].view(np.uint64), expect[~nan].view(np.uint64))
print(f"{w * h * spp / shard[1] / 1e6:.1f} M samples, {w * h} pixels x 4 doubles (rows of other shards must stay zero): {'identical' if same else 'DIFFERENT'}", flush=True)
sys.exit(0 if same else 1)
