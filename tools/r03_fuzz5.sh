#!/bin/bash
# the mode fuzzer on scenes moved away from the world origin (--offset), with and without far cameras
O=gpurun_out/${1:-r03fuzz7}; mkdir -p $O; rc=0
f() { name=$1; shift; timeout -k 10 ${T:-300} python3 tools/fuzz_modes.py --out $O "$@" > $O/$name.txt 2>&1 || rc=1; tail -1 $O/$name.txt; }
f offset_spheres_71 --seed 71 --scenes 3000 --offset --reference-walk
f offset_far_spheres_72 --seed 72 --scenes 3000 --offset --far --reference-walk
f offset_wide_73 --seed 73 --scenes 2000 --offset --triangles 1.0 --qnodes 2 --reference-walk
f offset_far_wide_74 --seed 74 --scenes 2000 --offset --far --triangles 0.3 --qnodes 2 --reference-walk
f offset_exact_75 --seed 75 --scenes 1000 --offset --triangles 0.3 --reference-walk
exit $rc
