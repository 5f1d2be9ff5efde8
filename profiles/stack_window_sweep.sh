#!/bin/bash
# Times the benchmark frame for several stack windows (PT_STACK_SLOTS), with the in-tree build and with build_variants/libpt_<variant>.so,
# and counts spills with the counting build at 16 spp.
# usage: profiles/stack_window_sweep.sh <outfile> <scene> <spp> <variant|main> <window> ...
out=$1; scene=$2; spp=$3; v=$4; shift 4
mkdir -p "$(dirname "$out")"
if [ "$v" = main ]; then unset PT_VARIANT; else export PT_VARIANT=$v; fi
for w in "$@"; do
  echo "== $v $scene ${spp}spp window $w" >> $out
  PT_STACK_SLOTS=$w ITERATIONS=3 timeout -k 10 120 python3 profiles/stats_run.py $spp $scene >> $out 2>&1 || exit 1
  PT_STACK_SLOTS=$w PT_STATS=1 ITERATIONS=1 timeout -k 10 120 python3 profiles/stats_run.py 16 $scene | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
s = d['stats']
print(json.dumps({'lanes': [round(d[k + '_lanes_per_phase'], 1) for k in ('node', 'leaf', 'shade')], 'per_path': d['per_path'], 'spills_per_path': s.get('spills', 0) / s['paths'], 'abandoned': s.get('abandoned'), 'unsafe_skips': s['unsafe_skips']}))" >> $out 2>&1 || exit 1
done
