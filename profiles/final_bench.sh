#!/bin/bash
# bench.py on every configuration (one GPU), then bench.py itself under rocprofv3 --kernel-trace --stats.
export TMPDIR=/tmp
ROUND=r3 bash profiles/bench_configs.sh || exit 1
mkdir -p gpurun_out/r3/bench_trace
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3/bench_trace -o t -- python3 bench.py > gpurun_out/r3/bench_under_rocprof.json 2> gpurun_out/r3/bench_under_rocprof.err || { tail -3 gpurun_out/r3/bench_under_rocprof.err; exit 1; }
find gpurun_out/r3/bench_trace -name "t_kernel_stats.csv" -exec cp {} gpurun_out/r3/bench_trace/kernel_stats.csv \;
head -4 gpurun_out/r3/bench_trace/kernel_stats.csv
