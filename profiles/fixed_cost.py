#!/usr/bin/env python3
"""Where a launch's time goes when there is little to do: frame time against the rows and samples asked for.
usage: python profiles/fixed_cost.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pathtracer_amd import api, ctypes_abi as abi
if os.environ.get("PT_VARIANT"):      # an A/B build made by profiles/build_variant.sh
    api.LIBRARY_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build_variants", "libpt_%s.so" % os.environ["PT_VARIANT"])
host = api.HostScene(abi.PT_SCENE_CHESS32, api.MODELS_DIR)
dev = api.DeviceScene.from_host(host)
w, h = 1920, 1080
buf = torch.zeros(4 * w * h, dtype=torch.float64, device="cuda:0")


def run(label, **changes):
    fr = host.frame(w, h, **changes)
    ms = []
    for _ in range(4):
        buf.zero_()
        torch.cuda.synchronize()
        dev.render_device(fr, buf.data_ptr())
        torch.cuda.synchronize()
        ms.append(round(dev.last_render_ms(), 3))
    print(label, ms, flush=True)


run("full frame, 256 spp, depth 50", sample_count=256, max_bounces=50)
run("full frame, 16 spp, depth 50", sample_count=16, max_bounces=50)
run("full frame, 1 spp, depth 50", sample_count=1, max_bounces=50)
run("full frame, 16 spp, depth 1", sample_count=16, max_bounces=1)
run("full frame, 16 spp, depth 4", sample_count=16, max_bounces=4)
run("full frame, 16 spp, depth 12", sample_count=16, max_bounces=12)
run("rows 1040..1080 (top), 16 spp", sample_count=16, max_bounces=50, row_begin=1040, row_end=1080)
run("rows 0..40 (bottom), 16 spp", sample_count=16, max_bounces=50, row_begin=0, row_end=40)
run("rows 500..540 (middle), 16 spp", sample_count=16, max_bounces=50, row_begin=500, row_end=540)
run("rows 500..540 (middle), 256 spp", sample_count=256, max_bounces=50, row_begin=500, row_end=540)
