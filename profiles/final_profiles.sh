#!/bin/bash
# Everything the counter tables and the documentation quote, measured on the tree as it stands: the GPU test suite, the five workloads under
# rocprofv3 (kernel trace, then one counter pass per group), phase counts, shard times.
set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r3/gpu_tests.log 2>&1 || { tail -5 gpurun_out/r3/gpu_tests.log; exit 1; }
tail -1 gpurun_out/r3/gpu_tests.log
bash profiles/profile_workload.sh chess32_1080p_256 chess32 1920x1080 256 50 || exit 1
bash profiles/profile_workload.sh mixed_1080p_256_d16 mixed 1920x1080 256 16 || exit 1
bash profiles/profile_workload.sh grid5_1080p_64 grid5 1920x1080 64 50 || exit 1
bash profiles/profile_workload.sh cornell_1024_256 cornell 1024x1024 256 50 || exit 1
bash profiles/profile_workload.sh spheres_1200x800_256 spheres 1200x800 256 50 || exit 1
PT_STATS=1 PT_SIZE=1920x1080 ITERATIONS=2 timeout -k 10 200 python3 profiles/stats_run.py 64 chess32 50 > gpurun_out/r3/phase_counts_64spp.json 2>/dev/null || exit 1
timeout -k 10 300 python3 profiles/shard_times.py > gpurun_out/r3/shard_times.log 2>&1 || exit 1
cat gpurun_out/r3/shard_times.log
