#!/bin/bash
# the mode fuzzer with a point light and an infinite plane in every scene (--lights), far cameras / scenes away from the origin
O=gpurun_out/${1:-r03fuzz10}; mkdir -p $O; rc=0
f() { name=$1; shift; timeout -k 10 ${T:-300} python3 tools/fuzz_modes.py --out $O "$@" > $O/$name.txt 2>&1 || rc=1; tail -1 $O/$name.txt; }
f lights_far_81 --seed 81 --scenes 3000 --lights --far --reference-walk
f lights_offset_far_82 --seed 82 --scenes 3000 --lights --offset --far --reference-walk
f lights_83 --seed 83 --scenes 1500 --lights --reference-walk
f lights_far_wide_84 --seed 84 --scenes 1500 --lights --far --triangles 0.5 --qnodes 2 --reference-walk
exit $rc
