#!/bin/bash
# the mode fuzzer in the regime of its one finding (--far: cameras 10^3 .. 10^5 scene sizes away, large overlapping spheres)
O=gpurun_out/${1:-r03fuzz4}; mkdir -p $O; rc=0
f() { name=$1; shift; timeout -k 10 ${T:-300} python3 tools/fuzz_modes.py --out $O "$@" > $O/$name.txt 2>&1 || rc=1; tail -1 $O/$name.txt; }
f far_spheres_61 --seed 61 --scenes 3000 --far --reference-walk
f far_wide_62 --seed 62 --scenes 2000 --far --triangles 1.0 --qnodes 2 --reference-walk
f far_exact_63 --seed 63 --scenes 2000 --far --triangles 1.0 --qnodes 0 --reference-walk
exit $rc
