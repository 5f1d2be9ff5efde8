#!/bin/bash
# Times the benchmark frame for several (shade, leaf) thresholds and reports lanes per phase.
# usage: profiles/threshold_sweep2.sh <outfile> <scene> <spp> <shade:leaf> ...
out=$1; scene=$2; spp=$3; shift 3
mkdir -p "$(dirname "$out")"
for pair in "$@"; do
  export PT_SHADE_THRESHOLD=${pair%%:*} PT_LEAF_THRESHOLD=${pair##*:}
  echo "== $scene ${spp}spp shade $PT_SHADE_THRESHOLD leaf $PT_LEAF_THRESHOLD ${PT_KERNEL:-} ${PT_VARIANT:-}" >> $out
  ITERATIONS=3 timeout -k 10 120 python3 profiles/stats_run.py $spp $scene 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps({'all_ms': d['all_ms'], 'msamples_per_s': round(d['msamples_per_s'])}))" >> $out || exit 1
  PT_STATS=1 ITERATIONS=1 timeout -k 10 120 python3 profiles/stats_run.py 16 $scene 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
s = d['stats']
print(json.dumps({'lanes': [round(d[k + '_lanes_per_phase'], 1) for k in ('node', 'leaf', 'shade')], 'per_path': {k: round(v, 2) for k, v in d['per_path'].items()}, 'loops_per_path': round(s['loops'] / s['paths'], 3), 'spills_per_path': round(s.get('spills', 0) / s['paths'], 4), 'abandoned': s.get('abandoned'), 'unsafe_skips': s['unsafe_skips']}))" >> $out || exit 1
done
