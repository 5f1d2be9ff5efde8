#!/bin/bash
# The split cost's leaf-count exponent (PT_SAH_POWER) once more on the final kernel: times and steps per path.
out=gpurun_out/sah_power_final.log; rm -f $out
for p in 0.8 0.9 1.0 1.1 1.25; do
  export PT_SAH_POWER=$p
  echo "== chess32 256spp power $p" >> $out
  PT_SIZE=1920x1080 ITERATIONS=3 timeout -k 10 200 python3 profiles/stats_run.py 256 chess32 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps({'all_ms': d['all_ms']}))" >> $out || exit 1
  PT_STATS=1 ITERATIONS=1 timeout -k 10 200 python3 profiles/stats_run.py 16 chess32 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps({'per_path': {k: round(v, 2) for k, v in d['per_path'].items()}}))" >> $out || exit 1
done
cat $out
