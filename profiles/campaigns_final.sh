#!/bin/bash
# The random parity campaigns on the final build (four waves per SIMD, stacks of 24 slots with a backing store), each
# against the CPU checker, bit for bit; then the large soups once more with stacks of 16 slots, so that deep searches park.
set -o pipefail
out=gpurun_out/campaigns_final.log; rm -f $out
run() { echo "== $*" >> $out; env "$@" 2>&1 | tail -2 >> $out || { echo FAILED >> $out; exit 1; }; }
run PT_RANDOM_SCENES=3000 timeout -k 10 600 python -u -m pytest tests/test_gpu_edges.py -m gpu -q -x -k random_scenes_against
run PT_RANDOM_FRAMES=6000 timeout -k 10 600 python -u -m pytest tests/test_gpu_edges.py -m gpu -q -x -k random_frames
run PT_RANDOM_CUT_SCENES=1000 timeout -k 10 600 python -u -m pytest tests/test_gpu_edges.py -m gpu -q -x -k every_triangle_cut
run PT_X=1 timeout -k 10 600 python3 -u profiles/random_big_scenes.py 6
run PT_STACK_SLOTS=16 timeout -k 10 600 python3 -u profiles/random_big_scenes.py 6
cat $out
