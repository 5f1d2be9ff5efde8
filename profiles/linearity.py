#!/usr/bin/env python3
"""Frame time against samples per pixel and bounce limit: what a launch costs beyond its samples.
usage: python profiles/linearity.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from pathtracer_amd import api, ctypes_abi as abi
if os.environ.get("PT_VARIANT"):
    api.LIBRARY_PATH = os.path.join(os.getcwd(), "build_variants", "libpt_%s.so" % os.environ["PT_VARIANT"])
host = api.HostScene(abi.PT_SCENE_CHESS32, api.MODELS_DIR)
dev = api.DeviceScene.from_host(host)
w, h = 1920, 1080
buf = torch.zeros(4 * w * h, dtype=torch.float64, device="cuda:0")
for depth in (1, 2, 4, 50):
    row = []
    for spp in (16, 64, 256):
        fr = host.frame(w, h, sample_count=spp, max_bounces=depth)
        ms = []
        for _ in range(3):
            buf.zero_(); torch.cuda.synchronize()
            dev.render_device(fr, buf.data_ptr()); torch.cuda.synchronize()
            ms.append(dev.last_render_ms())
        row.append(round(min(ms), 2))
    print("depth", depth, "spp 16/64/256:", row, "per 16 spp:", [round(row[0], 2), round(row[1] / 4, 2), round(row[2] / 16, 2)], flush=True)
