#!/usr/bin/env python3
"""Kernel time of small launches of the bench frame (few samples, few bounces): what a launch costs before it scales.
usage: PT_PROBE_BOUNCES=0 python profiles/small_launches.py [scene]"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from pathtracer_amd import api, ctypes_abi as abi
name = sys.argv[1] if len(sys.argv) > 1 else "chess32"
host = api.HostScene({"chess32": abi.PT_SCENE_CHESS32, "mixed": abi.PT_SCENE_CHESS32_MIXED}[name], api.MODELS_DIR)
dev = api.DeviceScene.from_host(host)
w, h = 1920, 1080
buf = torch.zeros(4 * w * h, dtype=torch.float64, device="cuda:0")
for depth in (1, 3, 50):
    row = []
    for spp in (1, 2, 4, 16, 32):
        fr = host.frame(w, h, sample_count=spp, max_bounces=depth)
        ms = []
        for _ in range(3):
            buf.zero_(); torch.cuda.synchronize()
            dev.render_device(fr, buf.data_ptr()); torch.cuda.synchronize()
            ms.append(dev.last_render_ms())
        row.append(round(min(ms), 3))
    print("depth", depth, "spp 1/2/4/16/32 ms:", row, flush=True)
