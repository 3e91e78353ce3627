#!/bin/bash
# a last round of every regime of the mode fuzzer, fresh seeds
O=gpurun_out/${1:-r03fuzz12}; mkdir -p $O; rc=0
f() { name=$1; shift; timeout -k 10 ${T:-200} python3 tools/fuzz_modes.py --out $O "$@" > $O/$name.txt 2>&1 || rc=1; tail -1 $O/$name.txt; }
f all_101 --seed 101 --scenes 3000 --far --offset --lights --dups --reference-walk
f all_wide_102 --seed 102 --scenes 2000 --far --offset --lights --dups --triangles 0.5 --qnodes 2 --reference-walk
f far_103 --seed 103 --scenes 4000 --far --reference-walk
f offset_104 --seed 104 --scenes 2000 --offset --reference-walk
f general_105 --seed 105 --scenes 1500 --reference-walk
f far_few_tris_106 --seed 106 --scenes 3000 --far --lights --triangles 0.1 --reference-walk
exit $rc
