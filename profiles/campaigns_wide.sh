#!/bin/bash
# The random parity campaigns at a larger width than profiles/campaigns_final.sh (the same generators, more seeds), each
# against the CPU checker, bit for bit.
set -o pipefail
out=gpurun_out/campaigns_wide.log; rm -f $out
# (each command writes straight into a file under gpurun_out/ as it goes, so that a long run is seen to be alive)
run() { echo "== $*" >> $out; env "$@" > gpurun_out/campaigns_wide_step.log 2>&1 || { tail -5 gpurun_out/campaigns_wide_step.log >> $out; echo FAILED >> $out; exit 1; }; tail -2 gpurun_out/campaigns_wide_step.log >> $out; }
run PT_RANDOM_SCENES=8000 timeout -k 10 900 python -u -m pytest tests/test_gpu_edges.py -m gpu -q -x -k random_scenes_against
run PT_RANDOM_FRAMES=20000 timeout -k 10 600 python -u -m pytest tests/test_gpu_edges.py -m gpu -q -x -k random_frames
run PT_RANDOM_CUT_SCENES=3000 timeout -k 10 600 python -u -m pytest tests/test_gpu_edges.py -m gpu -q -x -k every_triangle_cut
run PT_X=1 timeout -k 10 600 python3 -u profiles/random_big_scenes.py 12
cat $out
