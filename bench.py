#!/usr/bin/env python3
"""Headline benchmark: chess set, 1920x1080, 256 spp, depth 50, FP64, on N MI355X of one node.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one full frame: zero the f64 BGRA+count buffer, render every pixel's 256 samples
(rows dealt in blocks of 8 over the ranks), gather the rows to rank 0.  The scene (BVH,
triangles, materials) is resident in each GPU's HBM before the timed region.  Rank 0 prints one
JSON line.  Msamples/s = width * height * spp * steps / seconds / 1e6, whole job.

What the line's `roofline` is.  The dominant kernel (trace_paths_kernel) is priced against the resources it
can run out of, each from work the kernel EXECUTES:
  valu  wave-wide vector instructions issued (committed rocprofv3 count per path, profiles/executed_work.json)
        x 64 lanes x paths of the launch / live kernel time, against SIMDs x 16 lanes x clock;
  l1    accesses of the vector L1 (committed TCP_TOTAL_CACHE_ACCESSES per path: a 16-byte load by lanes that
        each stand on their own node or triangle costs one per lane) x paths / live kernel time, against one
        access per compute unit per cycle (measured: profiles/r2/micro_l1_gather.txt, profiles/r3/micro_l1_pieces.txt);
  hbm   memory-side bytes of the launch (committed FETCH_SIZE/WRITE_SIZE passes, profiles/hbm_traffic.json)
        / live kernel time, against the 8 TB/s HBM peak.
The primary `roofline` is whichever of them is nearest its peak (the binding one); all are given, with
`occupancy` (waves per SIMD of the 8 the hardware holds, from the library's own occupancy query).  The
reference's exhaustive traversal would touch far more bytes per sample than this kernel's pruned search;
that figure is reported apart, as `algorithmic`, and is no fraction of any peak.
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
CLOCK_GHZ = 2.4                 # max engine clock
LANES_PER_SIMD = 16             # a 64-wide vector instruction occupies its SIMD for 4 cycles
SIMDS_PER_CU = 4
L1_ACCESSES_PER_CU_PER_CYCLE = 1.0      # lanes that each read 16 bytes of their own line: measured, profiles/r3/micro_l1_pieces.txt
WAVES_PER_SIMD_MAX = 8
SCENES = {"chess32": abi.PT_SCENE_CHESS32, "chess6": abi.PT_SCENE_CHESS6_ROW, "mixed": abi.PT_SCENE_CHESS32_MIXED,
          "grid": abi.PT_SCENE_CHESS_GRID, "spheres": abi.PT_SCENE_RANDOM_SPHERES, "cornell": abi.PT_SCENE_CORNELL_BOX}
# bytes the kernel asks the memory hierarchy for, per executed step (what it reads of each record)
NODE_STEP_BYTES = 96            # six 16-byte plane rows of a 128-byte node (the triangle kernels put the child references together from them)
NODE_STEP_BYTES_FETCHED_REFERENCES = 112   # the kernel over triangles and spheres together fetches the four child references as well
TRIANGLE_TEST_BYTES = 80        # a, ab, ac, n.n of a triangle record
VERTEX_BYTES = 48 + 32 + 40     # the winner's own box, its unit normal + material index, the material
PATH_STORE_BYTES = 32           # one colour record per path


def table(name, key):
    """The committed counter entry of workload `key` ("scene WxH Nspp depthD"), or, if that size was not profiled, the
    scene's entry at another size with the same bounce limit: per-path figures barely depend on resolution and sample
    count.  Returns (entry, paths of the profiled launch, key used)."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, None, None
    entries = json.load(open(path))
    scene, _, _, depth = key.split()
    used = key if key in entries else next((k for k in sorted(entries) if k.split()[0] == scene and k.split()[3] == depth), None)
    if used is None:
        return None, None, None
    _, size, spp, _ = used.split()
    width, height = (int(v) for v in size.split("x"))
    return entries[used], width * height * int(spp[:-3]), used


def algorithmic_work(scene, width, height, max_bounces):
    """Per-sample bytes / f64 ops of the REFERENCE's exhaustive traversal (oracle/count_work.py)."""
    all_work = json.load(open(os.path.join(ROOT, "profiles", "algorithmic_work.json")))
    key = f"{scene}_{width}x{height}_depth{max_bounces}"
    if key not in all_work:
        # per-sample work barely depends on resolution; fall back to the scene's entry at another size
        others = [k for k in sorted(all_work) if k.startswith(scene + "_") and k.endswith(f"_depth{max_bounces}")]
        key = others[0] if others else key
    return all_work.get(key, {}).get("per_sample"), key


def host_threads():
    """Threads for the CPU leg: the CPUs this process may run on (the GPU box gives a one-GPU job a share of
    the machine and says so in OMP_NUM_THREADS; os.cpu_count() there counts the whole machine)."""
    allowed = len(os.sched_getaffinity(0))
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    threads = int(os.environ.get("PT_CPU_THREADS", min(allowed, share) if share > 0 else allowed))
    return max(1, threads), allowed, share


def cpu_baseline(host, width, height, spp, max_bounces):
    """The CPU renderer on this box's host cores, on a bounded sample of the same frame: all the cores this job
    has, then one thread on a band of rows."""
    from oracle.bindings import OracleLib, RefLib, have_ref
    threads, allowed, share = host_threads()
    use_ref = have_ref() and max_bounces == 50
    lib = RefLib() if use_ref else OracleLib()
    scene = lib.scene_create(host.arrays())       # builds its own BVH with its own builder
    frame = host.frame(width, height, sample_count=spp, max_bounces=max_bounces)
    start = time.perf_counter()
    scene.render(frame, threads=threads)
    seconds = time.perf_counter() - start
    # one thread: every 16th row block of 8 rows, so the band sees the whole picture's mix of rows
    band_rows = 0
    start = time.perf_counter()
    for begin in range(0, height, 128):
        end = min(begin + 8, height)
        scene.render(host.frame(width, height, sample_count=spp, max_bounces=max_bounces, row_begin=begin, row_end=end), threads=1)
        band_rows += end - begin
    one_seconds = time.perf_counter() - start
    return {
        "value": width * height * spp / seconds / 1e6, "unit": "Msamples/s", "cores": threads,
        "kind": "reference" if use_ref else "port",
        "one_thread": {"value": width * band_rows * spp / one_seconds / 1e6, "unit": "Msamples/s", "cores": 1,
                       "sample": f"{band_rows} rows (8 of every 128) x {spp} samples, {one_seconds:.1f} s"},
        "nproc": os.cpu_count(), "cpus_allowed": allowed, "cpu_share_announced": share or None,
        "sample": f"same scene and camera, full {width}x{height} frame, samples 0..{spp - 1} of each pixel "
                  f"({width * height * spp / 1e6:.1f} Msamples, {seconds:.1f} s) on {threads} threads = every CPU this job may use "
                  f"(the machine has {os.cpu_count()}); "
                  + ("the reference's own sources compiled -O2 with asserts live" if use_ref else "CPU restatement, -O2"),
    }


def verify_rows(host, accum, width, height, spp, max_bounces, rows):
    """Rows of the timed region's output against the CPU checker, bit for bit."""
    from oracle.bindings import OracleLib
    checker = OracleLib().scene_create(host.arrays())
    threads = host_threads()[0]
    for row in rows:
        frame = host.frame(width, height, sample_count=spp, max_bounces=max_bounces, row_begin=row, row_end=row + 1)
        expect = checker.render(frame, threads=threads).reshape(height, width, 4)[row]
        got = accum[row].cpu().numpy()
        nan = np.isnan(got)
        if not (np.array_equal(nan, np.isnan(expect)) and np.array_equal(got[~nan].view(np.uint64), expect[~nan].view(np.uint64))):
            return False
    return True


def executed_steps(scene, host, width, height, spp, max_bounces, device):
    """What the kernel does per path, counted by its profiling build (PT_STATS=1) on a few samples of the frame."""
    buf = torch.zeros((height, width, 4), dtype=torch.float64, device=device)
    os.environ["PT_STATS"] = "1"
    try:
        scene.render_device(host.frame(width, height, sample_count=spp, max_bounces=max_bounces), buf.data_ptr(),
                            torch.cuda.current_stream(device).cuda_stream)
        torch.cuda.synchronize(device)
        stats = scene.last_render_stats()
    except api.PathTracerError:
        return None           # the scene renders with the one-lane-per-pixel kernel: no phase counters
    finally:
        os.environ.pop("PT_STATS", None)
    paths = max(1, stats["paths"])
    steps = {"node_steps": stats["node_lanes"] / paths, "triangle_tests": stats["leaf_lanes"] / paths, "vertices": stats["shade_lanes"] / paths}
    node_bytes = NODE_STEP_BYTES_FETCHED_REFERENCES if len(host.arrays()["spheres"]) > 1 else NODE_STEP_BYTES
    steps["bytes_requested"] = (node_bytes * steps["node_steps"] + TRIANGLE_TEST_BYTES * steps["triangle_tests"]
                                + VERTEX_BYTES * steps["vertices"] + PATH_STORE_BYTES)
    lanes = {k: stats[k + "_lanes"] / max(1, stats[k + "_phases"]) for k in ("node", "leaf", "shade")}
    return {"per_path": steps, "active_lanes_of_64_per_phase": lanes, "counted_on": f"{spp} samples per pixel of the same frame",
            "f32_boxes_skipped_that_f64_accepts": stats["unsafe_skips"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=3)      # (the first launches on an idle GPU run 2-3 % slower)
    ap.add_argument("--scene", default="chess32", choices=sorted(SCENES))
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--max-bounces", type=int, default=50)
    ap.add_argument("--grid", type=int, default=5, help="boards per side of --scene grid (5 -> 10.56 M triangles)")
    ap.add_argument("--cpu-spp", type=int, default=8, help="samples per pixel of the CPU baseline leg (0 = skip)")
    ap.add_argument("--verify-rows", type=int, default=3, help="rows of the timed output compared with the CPU checker afterwards (0 = skip)")
    ap.add_argument("--stats-spp", type=int, default=16, help="samples per pixel of the executed-work count after the timed region (0 = skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the path has no CPU fallback")
    # PT_DIST_BACKEND=gloo rehearses the N-rank path where fewer GPUs than ranks exist (ranks share GPUs, the gather
    # goes through host memory): it exercises this script's sharding, timing and gather, not the interconnect.
    backend = os.environ.get("PT_DIST_BACKEND", "nccl")
    local_device = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_device)
    device = torch.device("cuda", local_device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # ---- set-up, not timed: load models, build the BVH on the host, upload to HBM
    t0 = time.perf_counter()
    host = api.HostScene(SCENES[args.scene], api.MODELS_DIR, args.grid if args.scene == "grid" else 0)
    scene_name = f"grid{args.grid}" if args.scene == "grid" else args.scene
    t1 = time.perf_counter()
    scene = api.DeviceScene.from_host(host, device=local_device)
    t2 = time.perf_counter()
    w, h = args.width, args.height
    frame = host.frame(w, h, sample_count=args.spp, max_bounces=args.max_bounces, shard_index=rank, shard_count=world)
    accum = torch.zeros((h, w, 4), dtype=torch.float64, device=device)
    stream = torch.cuda.current_stream(device).cuda_stream

    # the gather's buffers are made before the timed region (gather_rows keeps them per frame size); the rehearsal
    # backend gathers through a pinned host copy
    staging = torch.empty((h, w, 4), dtype=torch.float64, pin_memory=True) if (world > 1 and backend != "nccl") else None
    gather_events = []

    def step(timed=False):
        accum.zero_()
        scene.render_device(frame, accum.data_ptr(), stream)
        if world == 1:
            return
        before, after = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        before.record()
        if backend == "nccl":
            gather_rows(accum, h, w, rank, world)
        else:
            staging.copy_(accum)
            gather_rows(staging, h, w, rank, world)
            if rank == 0:
                accum.copy_(staging)
        after.record()
        if timed:
            gather_events.append((before, after))

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
        step(timed=True)
        kernel_ms.append(scene.last_render_ms())   # HIP events around the kernels on their launch stream
    fence()
    seconds = time.perf_counter() - start
    if world > 1:
        t = torch.tensor([seconds], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        seconds = float(t.item())
    # every rank's own kernel time and gather time (means over the timed steps), for rank 0 to report: what the slowest
    # rank adds is load imbalance, what the gather adds is the exchange
    per_rank = None
    if world > 1:
        gather_ms = float(np.mean([a.elapsed_time(b) for a, b in gather_events])) if gather_events else 0.0
        mine = torch.tensor([float(np.mean(kernel_ms)), gather_ms], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        everyone = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine)
        per_rank = {"kernel_ms": [float(v[0]) for v in everyone], "gather_ms": [float(v[1]) for v in everyone]}

    if rank == 0:
        samples_per_step = w * h * args.spp
        value = samples_per_step * args.steps / seconds / 1e6
        workload = f"{scene_name} {w}x{h} {args.spp}spp depth{args.max_bounces}"
        launch_s = float(np.mean(kernel_ms)) * 1e-3
        launch_paths = samples_per_step / world            # what one launch (this rank's) processes
        compute_units = torch.cuda.get_device_properties(device).multi_processor_count
        valu_peak = compute_units * SIMDS_PER_CU * LANES_PER_SIMD * CLOCK_GHZ * 1e9 / 1e12      # T lane-operations / s

        per_path, _, work_from = table("executed_work.json", workload)      # committed rocprofv3 instruction counts of this workload
        traffic, traffic_paths, traffic_from = table("hbm_traffic.json", workload)
        hbm_bytes = None if traffic is None else traffic["trace_paths_kernel"]["hbm_bytes_per_launch"] / traffic_paths * launch_paths
        blocks_per_cu = scene.last_launch_shape()[0]
        occupancy = {"waves_per_simd": blocks_per_cu, "of": WAVES_PER_SIMD_MAX, "frac": blocks_per_cu / WAVES_PER_SIMD_MAX,
                     "basis": "persistent blocks of 256 threads per compute unit = waves per SIMD (hipOccupancyMaxActiveBlocksPerMultiprocessor of the "
                              "kernel with its LDS stacks, made by the library before the launch); registers allow 3, LDS 3"}
        valu = hbm = l1 = None
        if per_path is not None and per_path.get("l1_accesses_per_path"):
            l1_peak = compute_units * CLOCK_GHZ * L1_ACCESSES_PER_CU_PER_CYCLE              # G accesses / s
            rate = per_path["l1_accesses_per_path"] * launch_paths / launch_s / 1e9
            l1 = {"bound": "l1", "achieved": rate, "peak": l1_peak, "unit": "Gaccess/s", "frac": rate / l1_peak,
                  "basis": f"{per_path['l1_accesses_per_path']:.1f} vector-L1 accesses per path (TCP_TOTAL_CACHE_ACCESSES, {per_path['source']}) x paths "
                           "/ live kernel time; peak = CUs x 2.4 GHz x one access per cycle (measured for 16-byte loads by lanes that stand on lines of their own which come from L2 or beyond, the kernel's case: profiles/r3/micro_l1_pieces.txt, micro_flat_lds.txt)"}
        if per_path is not None:
            tops = per_path["valu_wave_instructions_per_path"] * 64 * launch_paths / launch_s / 1e12
            valu = {"bound": "valu", "achieved": tops, "peak": valu_peak, "unit": "Tlane-op/s", "frac": tops / valu_peak,
                    "basis": f"{per_path['valu_wave_instructions_per_path']:.1f} vector instructions per path (SQ_INSTS_VALU, {per_path['source']}) "
                             "x 64 lanes x paths / live kernel time; peak = CUs x 4 SIMDs x 16 lanes x 2.4 GHz, every instruction counted as one "
                             "operation per lane (no fused multiply-add credit: the f64 arithmetic may not fuse)"}
        if hbm_bytes is not None:
            gbs = hbm_bytes / launch_s / 1e9
            hbm = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                   "basis": "FETCH_SIZE (doubled, as 16-byte-per-lane reads are tallied at half) + WRITE_SIZE of the kernel's launch, "
                            f"separate rocprofv3 --pmc passes ({traffic['source']}) / live kernel time"}
        candidates = [r for r in (valu, l1, hbm) if r is not None]
        roofline = {"bound": "valu", "achieved": None, "peak": valu_peak, "unit": "Tlane-op/s", "frac": None}
        if candidates:
            roofline = dict(max(candidates, key=lambda r: r["frac"]))
        roofline.update({"traffic": hbm_bytes, "kernel": "trace_paths_kernel (+ sum_samples_kernel, 2% of the time)", "kernel_ms": launch_s * 1e3,
                         "paths_per_launch": launch_paths, "valu": valu, "l1": l1, "hbm": hbm, "occupancy": occupancy,
                         "counters_profiled_on": {"valu": work_from, "l1": work_from, "hbm": traffic_from}})

        work, work_key = algorithmic_work(scene_name, w, h, args.max_bounces)
        executed = executed_steps(scene, host, w, h, args.stats_spp, args.max_bounces, device) if (world == 1 and args.stats_spp > 0) else None
        algorithmic = None
        if work is not None:
            algorithmic = {"what": "the reference's exhaustive traversal of the same samples (every node whose box passes, both children always), "
                                   "counted by the CPU checker; NOT work this kernel does and not a fraction of any peak",
                           "bytes_per_sample": work["bytes"], "f64_ops_per_sample": work["f64_ops"], "node_visits_per_sample": work["node_visits"],
                           "triangle_tests_per_sample": work["triangle_tests"], "work_table": work_key,
                           "equivalent_GBps": work["bytes"] * launch_paths / launch_s / 1e9}
            if executed is not None:
                algorithmic["bytes_avoided_ratio"] = work["bytes"] / executed["per_path"]["bytes_requested"]
        if executed is not None:
            executed["requested_GBps_from_caches"] = executed["per_path"]["bytes_requested"] * launch_paths / launch_s / 1e9
        line = {
            "metric": "Msamples/sec (whole node), chess set 1080p 256spp at 1/2/4/8 MI355X",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": seconds / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64",
            "data": "synthetic: deterministic chess-set scene built from the reference's six .triangles models; no random inputs",
            "config": {"workload": workload, "triangles": int(host.desc.triangle_count), "width": w, "height": h, "spp": args.spp,
                       "max_bounces": args.max_bounces,
                       "parallelism": f"rows in blocks of 8 round-robin over {world} GPU(s), scene replicated, one gather to rank 0"
                                      + ("" if backend == "nccl" or world == 1 else f" (rehearsal: {backend}, ranks share GPUs)")},
            "roofline": roofline, "executed": executed, "algorithmic": algorithmic,
            "per_rank": None if per_rank is None else dict(per_rank, kernel_ms_min=min(per_rank["kernel_ms"]), kernel_ms_max=max(per_rank["kernel_ms"]),
                                                           gather_ms_max=max(per_rank["gather_ms"]),
                                                           note="means over the timed steps; kernel = HIP events around each rank's launch, gather = events around "
                                                                "its pack + gather + unpack on the rank's stream (rank 0 receives and unpacks)"),
            "setup_seconds": {"load_and_bvh_build": t1 - t0, "flatten_and_upload": t2 - t1},
        }
        # the timed region's own output against the CPU checker (rank 0 holds every rank's rows after the gather)
        rows = sorted({1, h // 2 + 3, h - 2})[: args.verify_rows]
        line["verified"] = verify_rows(host, accum, w, h, args.spp, args.max_bounces, rows) if rows else None
        line["verified_rows"] = rows
        if world == 1 and args.cpu_spp > 0:
            line["cpu_baseline"] = cpu_baseline(host, w, h, args.cpu_spp, args.max_bounces)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
