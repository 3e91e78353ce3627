#!/bin/bash
# the mode fuzzer with exact copies among the spheres (--dups: ties in the hit distance; the smaller sorted index must win)
O=gpurun_out/${1:-r03fuzz11}; mkdir -p $O; rc=0
f() { name=$1; shift; timeout -k 10 ${T:-300} python3 tools/fuzz_modes.py --out $O "$@" > $O/$name.txt 2>&1 || rc=1; tail -1 $O/$name.txt; }
f dups_91 --seed 91 --scenes 2000 --dups --reference-walk
f dups_far_92 --seed 92 --scenes 2000 --dups --far --reference-walk
f dups_wide_93 --seed 93 --scenes 1500 --dups --triangles 0.5 --qnodes 2 --reference-walk
f dups_lights_94 --seed 94 --scenes 1500 --dups --lights --reference-walk
exit $rc
