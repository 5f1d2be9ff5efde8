#!/bin/bash
# Launch time with and without the probing launch that orders the blocks: whole frame at 16/64/256 spp and the slowest eighth shard.
# usage: profiles/tail_ab.sh <outfile> [scene]
out=$1; scene=${2:-chess32}
for probe in 0 2 3 6; do
  export PT_PROBE_BOUNCES=$probe
  echo "== PT_PROBE_BOUNCES=$probe $scene" >> $out
  for spp in 16 64 256; do
    ITERATIONS=3 timeout -k 10 120 python3 profiles/stats_run.py $spp $scene 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['spp'], 'spp', d['all_ms'])" >> $out || exit 1
  done
  timeout -k 10 200 python3 profiles/shard_times.py 2>/dev/null >> $out || exit 1
done
