out=gpurun_out/presplit_floor_final.log; rm -f $out
for fl in 0.1 0.2 0.35 0.5; do
  export PT_PRESPLIT_FLOOR=$fl
  echo "== chess32 256spp floor $fl" >> $out
  PT_SIZE=1920x1080 ITERATIONS=3 timeout -k 10 200 python3 profiles/stats_run.py 256 chess32 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps({'all_ms': d['all_ms']}))" >> $out || exit 1
done
unset PT_PRESPLIT_FLOOR
for f in 1.0 1.5 2.5; do
  export PT_PRESPLIT=$f
  echo "== chess32 256spp budget $f" >> $out
  PT_SIZE=1920x1080 ITERATIONS=3 timeout -k 10 200 python3 profiles/stats_run.py 256 chess32 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps({'all_ms': d['all_ms']}))" >> $out || exit 1
done
cat $out
