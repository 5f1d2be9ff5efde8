#!/bin/bash
# What holds the vector L1 up: its stall counters over one launch of the benchmark frame (64 spp), one rocprofv3 run per pass.
export TMPDIR=/tmp
out=gpurun_out/tcp_stalls; mkdir -p $out
timeout -s KILL 60 rocprofv3 -L > $out/counters_available.txt 2>&1
grep -o "TCP_[A-Z0-9_]*" $out/counters_available.txt | sort -u > $out/tcp_counter_names.txt
export PT_SIZE=1920x1080 ITERATIONS=2
prog="python3 profiles/stats_run.py 64 chess32 50"
run() { n=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $out/$n -o p -- $prog > $out/$n.log 2>&1 || { echo "pass $n failed"; exit 1; }; }
run t1 TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum GRBM_GUI_ACTIVE
run t2 TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
run t3 TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN2_sum
run t4 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum
echo passes done
