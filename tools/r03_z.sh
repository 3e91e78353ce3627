#!/bin/bash
# same-box A/B of tools/ab.py variants: r03_z.sh TAG "SPEC ..." lib ...   ("-" = the default build)
O=gpurun_out/$1; mkdir -p $O; rm -f $O/ab.txt; SPECS=$2; shift 2
for rep in 1 2; do for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset MIRT_LIB; else export MIRT_LIB=$PWD/cuda_ray_tracer_amd/_build/ab/$lib/libmirt.so; fi
  echo "== $lib" >> $O/ab.txt
  PERF_COUNT=0 PERF_FRAMES=4 timeout -k 10 300 python3 tools/perf4.py $SPECS >> $O/ab.txt 2>&1 || { cat $O/ab.txt; exit 1; }
done; done
grep -v amdgpu.ids $O/ab.txt
