#!/bin/bash
# leaf tests in every step of a pass (runtime flag of the leafrep2 variant) against the default build, same box
O=gpurun_out/${1:-r03w}; mkdir -p $O; rm -f $O/ab.txt
SPECS="tenthousand:1920:1080:16 spiral:1920:1080:16 redchair:3840:2160:64 synth:3840:2160:8"
run() { echo "== $1 MIRT_LEAF_REP=${MIRT_LEAF_REP:-}" >> $O/ab.txt; PERF_COUNT=0 PERF_FRAMES=4 timeout -k 10 300 python3 tools/perf4.py $SPECS >> $O/ab.txt 2>&1 || { cat $O/ab.txt; exit 1; }; }
for rep in 1 2; do
  unset MIRT_LIB MIRT_LEAF_REP; run default
  export MIRT_LIB=$PWD/cuda_ray_tracer_amd/_build/ab/leafrep2/libmirt.so; run leafrep2-off
  export MIRT_LEAF_REP=1; run leafrep2-on
done
grep -v amdgpu.ids $O/ab.txt
