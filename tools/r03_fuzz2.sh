#!/bin/bash
# more batches of the mode fuzzer on the final code (see r03_fuzz.sh)
O=gpurun_out/${1:-r03fuzz3}; mkdir -p $O; rc=0
f() { name=$1; shift; timeout -k 10 ${T:-300} python3 tools/fuzz_modes.py --out $O "$@" > $O/$name.txt 2>&1 || rc=1; tail -1 $O/$name.txt; }
f mixed_wide_51 --seed 51 --scenes 1500 --triangles 2.0 --qnodes 2 --reference-walk
f mixed_wide_52 --seed 52 --scenes 1500 --triangles 0.5 --qnodes 2 --reference-walk
f spheres_53 --seed 53 --scenes 2000 --reference-walk
f spheres_54 --seed 54 --scenes 2000 --reference-walk
f mixed_exact_55 --seed 55 --scenes 1500 --triangles 1.0 --qnodes 1 --reference-walk
f mixed_exact_56 --seed 56 --scenes 1500 --triangles 3.0 --qnodes 0 --reference-walk
exit $rc
