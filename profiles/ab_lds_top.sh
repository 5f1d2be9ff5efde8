#!/bin/bash
# A/B: the hierarchy's most visited nodes kept in LDS and read by flat loads (build_variants/libpt_top<K>.so), against the
# product build at the same number of blocks per CU.
out=gpurun_out/ab_lds_top.log; rm -f $out
run() { echo "== $1 ${PT_BLOCKS_PER_CU:-query} blocks/CU" >> $out; if [ "$1" = main ]; then unset PT_VARIANT; else export PT_VARIANT=$1; fi
  PT_SIZE=1920x1080 ITERATIONS=3 timeout -k 10 200 python3 profiles/stats_run.py 256 chess32 50 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps({'all_ms': d['all_ms'], 'digest': d.get('digest')}))" >> $out || exit 1; }
PT_BLOCKS_PER_CU=3 run main && run top85 && run top21 && run main
cat $out
