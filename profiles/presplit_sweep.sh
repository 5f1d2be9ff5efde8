#!/bin/bash
# Times a workload and counts its node steps and triangle tests per path for several budgets of cut triangles (PT_PRESPLIT).
# usage: profiles/presplit_sweep.sh <outfile> <scene> <spp> <fraction> ...
out=$1; scene=$2; spp=$3; shift 3
mkdir -p "$(dirname "$out")"
for f in "$@"; do
  export PT_PRESPLIT=$f
  echo "== $scene ${spp}spp presplit $f" >> $out
  ITERATIONS=3 timeout -k 10 200 python3 profiles/stats_run.py $spp $scene 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps({'all_ms': d['all_ms'], 'msamples_per_s': round(d['msamples_per_s']), 'scene_create_s': round(d['scene_create_s'], 2)}))" >> $out || exit 1
  PT_STATS=1 ITERATIONS=1 timeout -k 10 200 python3 profiles/stats_run.py 16 $scene 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
s = d['stats']
print(json.dumps({'lanes': [round(d[k + '_lanes_per_phase'], 1) for k in ('node', 'leaf', 'shade')], 'per_path': {k: round(v, 2) for k, v in d['per_path'].items()}, 'unsafe_skips': s['unsafe_skips']}))" >> $out || exit 1
done
