#!/bin/bash
# mode fuzzer batches (tools/fuzz_modes.py): every default-mode frame against the reference-walk mode of the same library
O=gpurun_out/${1:-r03fuzz2}; mkdir -p $O; rc=0
f() { name=$1; shift; timeout -k 10 ${T:-400} python3 tools/fuzz_modes.py --out $O "$@" > $O/$name.txt 2>&1 || rc=1; tail -1 $O/$name.txt; }
f mixed_wide_47 --seed 47 --scenes 1000 --triangles 3.0 --qnodes 2 --reference-walk
f spheres_45 --seed 45 --scenes 1500 --reference-walk
f spheres_46 --seed 46 --scenes 1500 --reference-walk
f mixed_exact_48 --seed 48 --scenes 1000 --triangles 1.0 --qnodes 1 --reference-walk
exit $rc
