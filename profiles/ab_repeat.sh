#!/bin/bash
# The iteration's second node step: how many lanes must still stand on a node for it to run (build_variants/libpt_rep<N>.so).
out=gpurun_out/ab_repeat_lanes.log; rm -f $out
for v in ${VARIANTS:-rep16 rep24 rep32 rep40}; do
  export PT_VARIANT=$v
  for job in "chess32 256 50 1920x1080" "grid5 64 50 1920x1080" "cornell 256 50 1024x1024" "spheres 256 50 1200x800"; do
    set -- $job
    echo "== $v $1" >> $out
    PT_SIZE=$4 ITERATIONS=3 timeout -k 10 200 python3 profiles/stats_run.py $2 $1 $3 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps({'all_ms': d['all_ms'], 'digest': d.get('digest')}))" >> $out || exit 1
  done
done
cat $out
