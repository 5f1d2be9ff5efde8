#!/bin/bash
# Quick A/B on the headline workload only: three timed launches per library variant ("main" = in-tree build) and, for the
# in-tree build, per pair of shade / leaf thresholds given in THRESHOLDS ("40/12 36/12 ...").
# usage: THRESHOLDS="..." profiles/ab_chess_only.sh <outfile> <variant> ...
out=$1; shift
mkdir -p "$(dirname "$out")"
one() {
  echo "== $1 shade ${PT_SHADE_THRESHOLD:-default} leaf ${PT_LEAF_THRESHOLD:-default}" >> $out
  PT_SIZE=1920x1080 ITERATIONS=3 timeout -k 10 200 python3 profiles/stats_run.py 256 chess32 50 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps({'all_ms': d['all_ms'], 'digest': d.get('digest')}))" >> $out || exit 1
}
for v in "$@"; do
  if [ "$v" = main ]; then unset PT_VARIANT; else export PT_VARIANT=$v; fi
  one $v
done
unset PT_VARIANT
for t in $THRESHOLDS; do
  export PT_SHADE_THRESHOLD=${t%/*} PT_LEAF_THRESHOLD=${t#*/}
  one main
done
