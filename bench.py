#!/usr/bin/env python3
"""Headline benchmark: chess set, 1920x1080, 256 spp, depth 50, FP64, on N MI355X of one node.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one full frame: zero the f64 BGRA+count buffer, render every pixel's 256 samples
(rows dealt in blocks of 8 over the ranks), gather the rows to rank 0.  The scene (BVH,
triangles, materials) is resident in each GPU's HBM before the timed region.  Rank 0 prints one
JSON line.  Msamples/s = width * height * spp * steps / seconds / 1e6, whole job.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from pathtracer_amd import api, ctypes_abi as abi  # noqa: E402
from pathtracer_amd.distributed import gather_rows  # noqa: E402

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E, spec
FP64_VALU_PEAK_TOPS = 39.3      # 78.6 TFLOP/s vector FP64 counts an FMA as two; the path may not fuse
SCENES = {"chess32": abi.PT_SCENE_CHESS32, "chess6": abi.PT_SCENE_CHESS6_ROW, "mixed": abi.PT_SCENE_CHESS32_MIXED,
          "grid": abi.PT_SCENE_CHESS_GRID, "spheres": abi.PT_SCENE_RANDOM_SPHERES, "cornell": abi.PT_SCENE_CORNELL_BOX}


def algorithmic_work(scene, width, height, max_bounces):
    """Per-sample bytes / f64 ops of the reference's traversal (oracle/count_work.py)."""
    table = json.load(open(os.path.join(ROOT, "profiles", "algorithmic_work.json")))
    key = f"{scene}_{width}x{height}_depth{max_bounces}"
    if key not in table:
        # per-sample work barely depends on resolution; fall back to the scene's entry at another size
        others = [k for k in sorted(table) if k.startswith(scene + "_") and k.endswith(f"_depth{max_bounces}")]
        key = others[0] if others else key
    return table.get(key, {}).get("per_sample"), key


def hbm_traffic(workload):
    """HBM bytes per launch of the dominant kernel from committed rocprofv3 PMC passes (or None)."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if not os.path.exists(path):
        return None
    entry = json.load(open(path)).get(workload)
    return None if entry is None else entry["trace_paths_kernel"]["hbm_bytes_per_launch"]


def cpu_baseline(host, width, height, spp, max_bounces):
    """The CPU renderer on this box's host cores, on a bounded sample of the same frame."""
    from oracle.bindings import OracleLib, RefLib, have_ref
    threads = int(os.environ.get("PT_CPU_THREADS", min(16, os.cpu_count() or 1)))
    use_ref = have_ref() and max_bounces == 50
    lib = RefLib() if use_ref else OracleLib()
    scene = lib.scene_create(host.arrays())       # builds its own BVH with its own builder
    frame = host.frame(width, height, sample_count=spp, max_bounces=max_bounces)
    start = time.perf_counter()
    scene.render(frame, threads=threads)
    seconds = time.perf_counter() - start
    return {
        "value": width * height * spp / seconds / 1e6, "unit": "Msamples/s", "cores": threads,
        "kind": "reference" if use_ref else "port",
        "sample": f"same scene and camera, full {width}x{height} frame, samples 0..{spp - 1} of each pixel "
                  f"({width * height * spp / 1e6:.1f} Msamples, {seconds:.1f} s); "
                  + ("the reference's own sources compiled -O2 with asserts live" if use_ref else "CPU restatement, -O2"),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="chess32", choices=sorted(SCENES))
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--max-bounces", type=int, default=50)
    ap.add_argument("--grid", type=int, default=5, help="boards per side of --scene grid (5 -> 10.56 M triangles)")
    ap.add_argument("--cpu-spp", type=int, default=8, help="samples per pixel of the CPU baseline leg (0 = skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    # ---- set-up, not timed: load models, build the BVH on the host, upload to HBM
    t0 = time.perf_counter()
    host = api.HostScene(SCENES[args.scene], api.MODELS_DIR, args.grid if args.scene == "grid" else 0)
    scene_name = f"grid{args.grid}" if args.scene == "grid" else args.scene
    t1 = time.perf_counter()
    scene = api.DeviceScene.from_host(host, device=local_rank)
    t2 = time.perf_counter()
    w, h = args.width, args.height
    frame = host.frame(w, h, sample_count=args.spp, max_bounces=args.max_bounces, shard_index=rank, shard_count=world)
    accum = torch.zeros((h, w, 4), dtype=torch.float64, device=device)
    stream = torch.cuda.current_stream(device).cuda_stream

    def step():
        accum.zero_()
        scene.render_device(frame, accum.data_ptr(), stream)
        gather_rows(accum, h, w, rank, world)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()
    fence()
    kernel_ms = []
    start = time.perf_counter()
    for _ in range(args.steps):
        step()
        kernel_ms.append(scene.last_render_ms())   # HIP events around the kernel on its launch stream
    fence()
    seconds = time.perf_counter() - start
    if world > 1:
        t = torch.tensor([seconds], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        seconds = float(t.item())

    if rank == 0:
        samples_per_step = w * h * args.spp
        value = samples_per_step * args.steps / seconds / 1e6
        work, work_key = algorithmic_work(scene_name, w, h, args.max_bounces)
        launch_ms = float(np.mean(kernel_ms))
        launch_samples = samples_per_step / world            # what one launch (this rank's) processes
        roofline, valu = None, None
        if work is not None:
            gbs = work["bytes"] * launch_samples / (launch_ms * 1e-3) / 1e9
            workload = f"{scene_name} {w}x{h} {args.spp}spp depth{args.max_bounces}"
            roofline = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                        "traffic": hbm_traffic(workload) if world == 1 else None,
                        "kernel": "trace_paths_kernel (+ sum_samples_kernel, <1%)", "kernel_ms": launch_ms,
                        "algorithmic_bytes_per_sample": work["bytes"], "work_table": work_key}
            tops = work["f64_ops"] * launch_samples / (launch_ms * 1e-3) / 1e12
            valu = {"bound": "fp64 valu (no FMA)", "achieved": tops, "peak": FP64_VALU_PEAK_TOPS, "unit": "Tops/s",
                    "frac": tops / FP64_VALU_PEAK_TOPS, "algorithmic_f64_ops_per_sample": work["f64_ops"]}
        line = {
            "metric": "Msamples/sec (whole node), chess set 1080p 256spp at 1/2/4/8 MI355X",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": seconds / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64",
            "data": "synthetic: deterministic chess-set scene built from the reference's six .triangles models; no random inputs",
            "config": {"workload": f"{scene_name} {w}x{h} {args.spp}spp depth{args.max_bounces}",
                       "triangles": int(host.desc.triangle_count), "width": w, "height": h, "spp": args.spp,
                       "max_bounces": args.max_bounces,
                       "parallelism": f"rows in blocks of 8 round-robin over {world} GPU(s), scene replicated, one gather to rank 0"},
            "roofline": roofline, "roofline_valu": valu,
            "setup_seconds": {"load_and_bvh_build": t1 - t0, "flatten_and_upload": t2 - t1},
        }
        if world == 1 and args.cpu_spp > 0:
            line["cpu_baseline"] = cpu_baseline(host, w, h, args.cpu_spp, args.max_bounces)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
