#!/usr/bin/env python3
"""Kernel time of every row shard of the bench frame when it is split N ways (run one after another on one GPU).
usage: python profiles/shard_times.py"""
import sys, os, json
sys.path.insert(0, os.getcwd())
import torch
from pathtracer_amd import api, ctypes_abi as abi
if os.environ.get("PT_VARIANT"):
    api.LIBRARY_PATH = os.path.join(os.getcwd(), "build_variants", "libpt_%s.so" % os.environ["PT_VARIANT"])
host = api.HostScene(abi.PT_SCENE_CHESS32, api.MODELS_DIR)
dev = api.DeviceScene.from_host(host)
w, h = 1920, 1080
buf = torch.zeros(4 * w * h, dtype=torch.float64, device="cuda:0")
for world in (1, 2, 4, 8):
    out = []
    for rank in range(world):
        fr = host.frame(w, h, sample_count=256, max_bounces=50, shard_index=rank, shard_count=world)
        ms = []
        for i in range(3):
            buf.zero_()
            dev.render_device(fr, buf.data_ptr())
            torch.cuda.synchronize()
            ms.append(dev.last_render_ms())
        out.append(round(min(ms), 2))
    print(world, "GPU shards, kernel ms of each on one GPU:", out, "max", max(out), flush=True)
