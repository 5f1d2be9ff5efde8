#!/bin/bash
# Frame time against the shade / leaf thresholds (lanes that must wait before their phase runs).
# usage: profiles/threshold_sweep.sh <outfile> <scene> <spp>
out=$1; scene=$2; spp=$3
for shade in 24 28 32 36 40 48; do
  for leaf in 12 16 20 24 28 32; do
    PT_SHADE_THRESHOLD=$shade PT_LEAF_THRESHOLD=$leaf ITERATIONS=3 timeout -k 10 100 python3 profiles/stats_run.py $spp $scene 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$scene shade $shade leaf $leaf', min(d['all_ms']))" >> $out || exit 1
  done
done
