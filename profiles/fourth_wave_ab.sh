#!/bin/bash
# What a fourth wave per SIMD is worth by itself: the kernel compiled for 128 registers (build_variants/libpt_w4.so,
# -DPT_WAVES_PER_SIMD=4: the compiler spills) run with three and with four blocks per compute unit, beside the product build;
# all with 24 stack slots per lane, which is what four blocks leave room for in LDS.
# usage: profiles/build_variant.sh w4 -DPT_WAVES_PER_SIMD=4; profiles/fourth_wave_ab.sh <outfile> [scene] [spp]
out=$1; scene=${2:-chess32}; spp=${3:-256}
mkdir -p "$(dirname "$out")"
run() { echo "== $1" >> $out; shift; env "$@" ITERATIONS=3 timeout -k 10 120 python3 profiles/stats_run.py $spp $scene 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps({'all_ms': d['all_ms'], 'msamples_per_s': round(d['msamples_per_s'])}))" >> $out || exit 1; }
run "product build, its own stack (36 slots), 3 blocks per CU" PT_X=0
run "product build, 24 slots, 3 blocks per CU" PT_STACK_SLOTS=24
run "128-register build, 24 slots, 3 blocks per CU" PT_VARIANT=w4 PT_STACK_SLOTS=24 PT_BLOCKS_PER_CU=3
run "128-register build, 24 slots, 4 blocks per CU" PT_VARIANT=w4 PT_STACK_SLOTS=24 PT_BLOCKS_PER_CU=4
run "product build, 24 slots, 2 blocks per CU" PT_STACK_SLOTS=24 PT_BLOCKS_PER_CU=2
