#!/bin/bash
# The five benchmark frames whole, at their full sample counts, against the CPU checker; then the chess frame with
# the stacks cut to 16 and 20 slots (every deep search parks entries in the backing store): same digest expected.
set -o pipefail
out=gpurun_out/full_frame_parity.log; rm -f $out
for job in "chess32 1920x1080 256 50" "mixed 1920x1080 256 16" "spheres 1200x800 256 50" "cornell 1024x1024 256 50" "grid5 1920x1080 64 50"; do
  timeout -k 10 600 python3 -u profiles/full_frame_parity.py $job 16 2>/dev/null | grep -v "^  rows " >> $out || { echo "FAILED: $job" >> $out; exit 1; }
done
for slots in 16 20 36; do
  echo "== PT_STACK_SLOTS=$slots" >> $out
  PT_STACK_SLOTS=$slots PT_SIZE=1920x1080 ITERATIONS=1 timeout -k 10 200 python3 profiles/stats_run.py 64 chess32 50 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps({'all_ms': d['all_ms'], 'digest': d.get('digest')}))" >> $out || exit 1
done
cat $out
