#!/bin/bash
# Kernel trace and counter passes of one workload per call:
#   profiles/profile_workload.sh <name> <scene> <WxH> <spp> <bounces>
# kernel trace + stats, then counter passes (each its own rocprofv3 run), all under gpurun_out/$ROUND/<name>/ (ROUND defaults to r3).
export TMPDIR=/tmp
name=$1; scene=$2; size=$3; spp=$4; bounces=$5
out=gpurun_out/${ROUND:-r3}/$name
mkdir -p $out
export PT_SIZE=$size ITERATIONS=2
prog="python3 profiles/stats_run.py $spp $scene $bounces"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- $prog > $out/trace.log 2>&1 || { echo "trace failed"; exit 1; }
run() { n=$1; shift; timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d $out/$n -o p -- $prog > $out/$n.log 2>&1 || { echo "pass $n failed"; exit 1; }; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD
run b TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE SQ_INSTS_LDS
run d FETCH_SIZE
run e WRITE_SIZE
echo "$name done"
