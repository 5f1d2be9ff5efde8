#!/bin/bash
# Frame time and executed steps per path for a list of environment settings.
# usage: profiles/env_sweep.sh <outfile> <scene> <spp> "VAR=1 OTHER=2" "VAR=3" ...
out=$1; scene=$2; spp=$3; shift 3
for setting in "$@"; do
  echo "== $setting | $scene ${spp}spp" >> $out
  env $setting ITERATIONS=3 timeout -k 10 200 python3 profiles/stats_run.py $spp $scene 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['all_ms'], 'create', round(d['scene_create_s'], 2))" >> $out || exit 1
  env $setting PT_STATS=1 ITERATIONS=1 timeout -k 10 200 python3 profiles/stats_run.py 16 $scene 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: round(v, 3) for k, v in d['per_path'].items()}, 'lanes', round(d['node_lanes_per_phase'], 1), round(d['leaf_lanes_per_phase'], 1), round(d['shade_lanes_per_phase'], 1))" >> $out || exit 1
done
