#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of profiles/profile_workload.sh (gpurun_out/$ROUND/<name>/) into the tables bench.py reads:
profiles/executed_work.json, profiles/hbm_traffic.json, and a summary per workload under profiles/$ROUND/<name>/ (ROUND defaults to r3).
usage: python profiles/make_tables.py name:scene:WxH:spp:bounces ..."""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = os.environ.get("ROUND", "r3")


def counters(directory, kernel):
    total, launches = collections.OrderedDict(), {}
    for f in sorted(glob.glob(os.path.join(directory, "*", "**", "p_counter_collection.csv"), recursive=True)):
        for row in csv.DictReader(open(f)):
            if kernel not in row["Kernel_Name"]:
                continue
            total[row["Counter_Name"]] = total.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            launches.setdefault(row["Counter_Name"], set()).add(row["Dispatch_Id"])
    return {k: v / len(launches[k]) for k, v in total.items()}      # per launch


def main():
    work = json.load(open(os.path.join(ROOT, "profiles", "executed_work.json")))
    traffic = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    for spec in sys.argv[1:]:
        name, scene, size, spp, bounces = spec.split(":")
        w, h = (int(v) for v in size.split("x"))
        src = os.path.join(ROOT, "gpurun_out", ROUND, name)
        dst = os.path.join(ROOT, "profiles", ROUND, name)
        os.makedirs(dst, exist_ok=True)
        key = f"{scene} {w}x{h} {spp}spp depth{bounces}"
        paths = w * h * int(spp)
        trace = counters(src, "trace_paths_kernel")
        sums = counters(src, "sum_samples_kernel")
        stats = glob.glob(os.path.join(src, "trace", "**", "t_kernel_stats.csv"), recursive=True)
        if stats:
            shutil.copy(stats[0], os.path.join(dst, "kernel_stats.csv"))
        kernel_ms = None
        for row in csv.DictReader(open(stats[0])) if stats else []:
            if "trace_paths_kernel" in row["Name"]:
                kernel_ms = float(row["AverageNs"]) / 1e6
        cycles = trace.get("GRBM_GUI_ACTIVE", 0.0) / 8.0           # the counter sums the 8 XCDs
        summary = {"workload": key, "paths_per_launch": paths, "trace_paths_kernel_ms": kernel_ms, "per_launch": trace, "sum_samples_per_launch": sums,
                   "per_path": {k: trace[k] / paths for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum") if k in trace}}
        if cycles:
            summary["derived"] = {
                "cycles": cycles,
                "valu_issue_busy": trace.get("SQ_ACTIVE_INST_VALU", 0.0) * 4 / 1024 / cycles,
                "l1_accesses_per_cu_per_cycle": trace.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0) / 256 / cycles,
                "td_busy": trace.get("TD_TD_BUSY_sum", 0.0) / 256 / cycles, "ta_busy": trace.get("TA_TA_BUSY_sum", 0.0) / 256 / cycles,
                "l2_hit_rate": trace.get("TCC_HIT_sum", 0.0) / max(1.0, trace.get("TCC_HIT_sum", 0.0) + trace.get("TCC_MISS_sum", 0.0))}
        hbm = 2 * 1024 * trace.get("FETCH_SIZE", 0.0) + 1024 * trace.get("WRITE_SIZE", 0.0)
        summary["hbm_bytes_per_launch"] = hbm
        if kernel_ms:
            summary["hbm_GBps"] = hbm / kernel_ms / 1e6
        json.dump(summary, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
        source = f"profiles/{ROUND}/{name}/pmc_summary.json (rocprofv3 --pmc, one pass per counter group, python3 profiles/stats_run.py {spp} {scene} {bounces}, PT_SIZE={size}, mean of two launches)"
        work[key] = {"valu_wave_instructions_per_path": trace["SQ_INSTS_VALU"] / paths, "salu_per_path": trace["SQ_INSTS_SALU"] / paths,
                     "vmem_read_per_path": trace["SQ_INSTS_VMEM_RD"] / paths, "l1_accesses_per_path": trace.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0) / paths,
                     "source": source}
        traffic[key] = {"trace_paths_kernel": {"FETCH_SIZE_KB": trace["FETCH_SIZE"], "WRITE_SIZE_KB": trace["WRITE_SIZE"], "hbm_bytes_per_launch": hbm,
                                               "note": "FETCH_SIZE doubled (gfx950 tallies 16-byte-per-lane reads at half; an upper bound for scattered reads), WRITE_SIZE as read"},
                        "sum_samples_kernel": {"FETCH_SIZE_KB": sums.get("FETCH_SIZE"), "WRITE_SIZE_KB": sums.get("WRITE_SIZE"),
                                               "hbm_bytes_per_launch": 2 * 1024 * sums.get("FETCH_SIZE", 0.0) + 1024 * sums.get("WRITE_SIZE", 0.0)},
                        "source": source}
        print(key, json.dumps(summary.get("derived")), "hbm GB/s", summary.get("hbm_GBps"), "ms", kernel_ms)
    json.dump(work, open(os.path.join(ROOT, "profiles", "executed_work.json"), "w"), indent=1)
    json.dump(traffic, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
