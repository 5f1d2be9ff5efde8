#!/bin/bash
# Blocks of tiles dealt costliest first (by what they cost the previous launch) against dealt as they lie: the benchmark
# frame, four launches each (the first of a handle has no costs yet), then the shards of 2, 4 and 8 GPUs one after another.
out=gpurun_out/ab_block_order.log; rm -f $out
for order in 0 1; do
  echo "== PT_BLOCK_ORDER=$order chess32 1920x1080 256 spp" >> $out
  PT_BLOCK_ORDER=$order PT_SIZE=1920x1080 ITERATIONS=5 timeout -k 10 200 python3 profiles/stats_run.py 256 chess32 50 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps({'all_ms': d['all_ms'], 'digest': d.get('digest')}))" >> $out || exit 1
  echo "== PT_BLOCK_ORDER=$order chess32 1920x1080 64 spp" >> $out
  PT_BLOCK_ORDER=$order PT_SIZE=1920x1080 ITERATIONS=5 timeout -k 10 200 python3 profiles/stats_run.py 64 chess32 50 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps({'all_ms': d['all_ms'], 'digest': d.get('digest')}))" >> $out || exit 1
  echo "== PT_BLOCK_ORDER=$order shards" >> $out
  PT_BLOCK_ORDER=$order timeout -k 10 300 python3 profiles/shard_times.py 2>/dev/null >> $out || exit 1
done
cat $out
