#!/bin/bash
# Frame time and executed steps against the height the search hierarchy is capped to.
# usage: profiles/height_sweep.sh <outfile> <scene> <spp> <height> ...
out=$1; scene=$2; spp=$3; shift 3
for h in "$@"; do
  echo "== PT_SEARCH_HEIGHT=$h $scene ${spp}spp" >> $out
  PT_SEARCH_HEIGHT=$h PT_VERBOSE=1 ITERATIONS=3 timeout -k 10 200 python3 profiles/stats_run.py $spp $scene 2>&1 | grep "all_ms\|four-wide" | cut -c1-200 >> $out || exit 1
  PT_SEARCH_HEIGHT=$h PT_STATS=1 ITERATIONS=1 timeout -k 10 200 python3 profiles/stats_run.py 16 $scene 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({k: d.get(k) for k in ('node_lanes_per_phase', 'leaf_lanes_per_phase', 'per_path')}))" >> $out || exit 1
done
