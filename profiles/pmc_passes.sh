#!/bin/bash
# Counter passes for the render kernels, one rocprofv3 run per pass (block slot limits).
# usage: profiles/pmc_passes.sh <outdir> -- runs `python3 profiles/stats_run.py 16`
export TMPDIR=/tmp
OUT=${1:-gpurun_out/pmc}
PROG="python3 profiles/stats_run.py ${2:-16} ${3:-chess32}"
mkdir -p $OUT
run() { n=$1; shift; timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$n -o p -- $PROG > $OUT/$n.log 2>&1 || echo "pass $n failed"; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM
run b TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
run c SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS TD_TD_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum
run d FETCH_SIZE
run e WRITE_SIZE
