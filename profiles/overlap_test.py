#!/usr/bin/env python3
"""Does the end of one frame's launch overlap with the start of the next when they run on two streams
(two scene handles, so each launch has its own colour scratch and counter)?
usage: python profiles/overlap_test.py [spp] [shard_count]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pathtracer_amd import api, ctypes_abi as abi
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
shards = int(sys.argv[2]) if len(sys.argv) > 2 else 1
host = api.HostScene(abi.PT_SCENE_CHESS32, api.MODELS_DIR)
scenes = [api.DeviceScene.from_host(host), api.DeviceScene.from_host(host)]
w, h = 1920, 1080
bufs = [torch.zeros(4 * w * h, dtype=torch.float64, device="cuda:0") for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
fr = host.frame(w, h, sample_count=spp, max_bounces=50, shard_index=0, shard_count=shards)
K = 8


def run(two_streams):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        slot = i % 2 if two_streams else 0
        with torch.cuda.stream(streams[slot]):
            bufs[slot].zero_()
            scenes[slot].render_device(fr, bufs[slot].data_ptr(), streams[slot].cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3


for _ in range(2):
    print("one stream  %.2f ms per frame" % run(False), flush=True)
    print("two streams %.2f ms per frame" % run(True), flush=True)
ref = bufs[0].clone()
print("same bytes on both slots:", bool(torch.equal(bufs[0].view(torch.int64), bufs[1].view(torch.int64))))
