#!/bin/bash
# A/B timing of library variants on the benchmark frame: the in-tree build ("main") and build_variants/libpt_<name>.so.
# usage: profiles/ab.sh <outfile> <spp> <scene> <variant|main> ...   (each: 3 timed launches, then a PT_STATS=1 count at 16 spp)
out=$1; spp=$2; scene=$3; shift 3
for v in "$@"; do
  if [ "$v" = main ]; then unset PT_VARIANT; else export PT_VARIANT=$v; fi
  echo "== $v $scene ${spp}spp" >> $out
  ITERATIONS=3 timeout -k 10 120 python3 profiles/stats_run.py $spp $scene >> $out 2>&1 || exit 1
  PT_STATS=1 ITERATIONS=1 timeout -k 10 120 python3 profiles/stats_run.py 16 $scene | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({k: d.get(k) for k in ('node_lanes_per_phase', 'leaf_lanes_per_phase', 'shade_lanes_per_phase', 'per_path')}))" >> $out 2>&1 || exit 1
done
