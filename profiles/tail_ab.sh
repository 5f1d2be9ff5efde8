#!/bin/bash
# Launch time of library variants where the end of a launch weighs most: whole frame at 16/64/256 spp and the slowest eighth shard.
# usage: profiles/tail_ab.sh <outfile> <scene> <variant|main> ...
out=$1; scene=$2; shift 2
for v in "$@"; do
  if [ "$v" = main ]; then unset PT_VARIANT; else export PT_VARIANT=$v; fi
  echo "== $v $scene" >> $out
  for spp in 16 64 256; do
    ITERATIONS=3 timeout -k 10 120 python3 profiles/stats_run.py $spp $scene 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['spp'], 'spp', d['all_ms'])" >> $out || exit 1
  done
  timeout -k 10 200 python3 profiles/shard_times.py 2>/dev/null >> $out || exit 1
done
