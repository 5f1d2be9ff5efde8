#!/usr/bin/env python3
"""Runs the wavefront kernel with PT_STATS=1 on a reduced-spp frame and prints phase/lane counts.
usage: PT_STATS=1 python profiles/stats_run.py [spp] [scene]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pathtracer_amd import api, ctypes_abi as abi
if os.environ.get("PT_VARIANT"):      # an A/B build made by profiles/build_variant.sh
    api.LIBRARY_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build_variants", "libpt_%s.so" % os.environ["PT_VARIANT"])
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
name = sys.argv[2] if len(sys.argv) > 2 else "chess32"
bounces = int(sys.argv[3]) if len(sys.argv) > 3 else 50
if name.startswith("grid"):
    host = api.HostScene(abi.PT_SCENE_CHESS_GRID, api.MODELS_DIR, int(name[4:]))
else:
    host = api.HostScene({"chess32": abi.PT_SCENE_CHESS32, "chess6": abi.PT_SCENE_CHESS6_ROW, "mixed": abi.PT_SCENE_CHESS32_MIXED,
                          "spheres": abi.PT_SCENE_RANDOM_SPHERES, "cornell": abi.PT_SCENE_CORNELL_BOX}[name], api.MODELS_DIR)
import time
t0 = time.time()
dev = api.DeviceScene.from_host(host)
create_seconds = time.time() - t0
w, h = (int(v) for v in os.environ.get("PT_SIZE", "1920x1080").split("x"))
buf = torch.zeros(4 * w * h, dtype=torch.float64, device="cuda:0")
fr = host.frame(w, h, sample_count=spp, max_bounces=bounces)
all_ms = []
for i in range(int(os.environ.get("ITERATIONS", "2"))):
    buf.zero_()
    dev.render_device(fr, buf.data_ptr())
    torch.cuda.synchronize()
    all_ms.append(round(dev.last_render_ms(), 2))
ms = dev.last_render_ms()
import hashlib
digest = hashlib.sha256(buf.cpu().numpy().tobytes()).hexdigest()[:16]      # variants of the library must give the same bytes
out = {"digest": digest, "all_ms": all_ms, "scene": name, "spp": spp, "max_bounces": bounces, "scene_create_s": create_seconds, "ms": ms, "msamples_per_s": w * h * spp / ms / 1e3}
try:
    st = dev.last_render_stats()
    out["stats"] = st
    for k in ("node", "leaf", "shade"):
        out[k + "_lanes_per_phase"] = st[k + "_lanes"] / max(1, st[k + "_phases"])
    out["per_path"] = {k: st[k + "_lanes"] / st["paths"] for k in ("node", "leaf", "shade")}
    if st.get("t_total"):
        out["clock_share"] = {k: st["t_" + k] / st["t_total"] for k in ("node", "leaf", "shade", "refill", "setup")}
except Exception as e:
    out["stats_error"] = str(e)
print(json.dumps(out))
