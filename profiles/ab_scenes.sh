#!/bin/bash
# A/B over the benchmark workloads: for each library variant ("main" = in-tree build, else build_variants/libpt_<name>.so)
# three timed launches of each workload.
# usage: profiles/ab_scenes.sh <outfile> <variant> ...      (extra environment, e.g. PT_KERNEL or PT_STACK_SLOTS, is passed through)
out=$1; shift
mkdir -p "$(dirname "$out")"
for v in "$@"; do
  if [ "$v" = main ]; then unset PT_VARIANT; else export PT_VARIANT=$v; fi
  for job in "chess32 256 50 1920x1080" "mixed 256 16 1920x1080" "grid5 64 50 1920x1080" "cornell 256 50 1024x1024" "spheres 256 50 1200x800"; do
    set -- $job
    echo "== $v $1 $4 ${2}spp depth $3 ${PT_KERNEL:-} window ${PT_STACK_SLOTS:-default}" >> $out
    PT_SIZE=$4 ITERATIONS=3 timeout -k 10 200 python3 profiles/stats_run.py $2 $1 $3 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps({'all_ms': d['all_ms'], 'msamples_per_s': round(d['msamples_per_s']), 'digest': d.get('digest')}))" >> $out || exit 1
  done
done
