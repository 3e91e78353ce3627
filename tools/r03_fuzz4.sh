#!/bin/bash
# far-camera batches that walk the EXACT records (qnodes 0 on sphere scenes; few triangles among many spheres, default options)
O=gpurun_out/${1:-r03fuzz6}; mkdir -p $O; rc=0
f() { name=$1; shift; timeout -k 10 ${T:-300} python3 tools/fuzz_modes.py --out $O "$@" > $O/$name.txt 2>&1 || rc=1; tail -1 $O/$name.txt; }
f far_spheres_exact_64 --seed 64 --scenes 4000 --far --qnodes 0 --reference-walk
f far_few_triangles_65 --seed 65 --scenes 3000 --far --triangles 0.1 --reference-walk
f far_spheres_66 --seed 66 --scenes 3000 --far --reference-walk
exit $rc
