#!/bin/bash
# bench.py on the headline workload and the other configurations (one GPU); lines under gpurun_out/${ROUND:-r3}/configs/.
out=gpurun_out/${ROUND:-r3}/configs; mkdir -p $out
run() { name=$1; shift; timeout -k 10 400 python3 bench.py "$@" > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; exit 1; }; python3 -c "
import json, sys
d = json.loads(open('$out/$name.json').read().strip().splitlines()[-1])
print('$name', round(d['value'], 1), 'Msamples/s', round(d['ms_per_step'], 2), 'ms', d['roofline']['bound'], d['roofline']['frac'] and round(d['roofline']['frac'], 3), 'verified', d['verified'], 'setup', {k: round(v, 2) for k, v in d['setup_seconds'].items()})"; }
run bench_default
run bench_config3_4k_1024spp_1gpu --width 3840 --height 2160 --spp 1024 --steps 2 --cpu-spp 2
run bench_config4_mixed_depth16 --scene mixed --max-bounces 16
run bench_config5_grid5_64spp --scene grid --spp 64 --verify-rows 1
run bench_cornell_1024_256spp --scene cornell --width 1024 --height 1024
run bench_spheres_1200x800_256spp --scene spheres --width 1200 --height 800
