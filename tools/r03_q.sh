#!/bin/bash
O=gpurun_out/${1:-r03q}; mkdir -p $O; rm -f $O/dbl.txt
for v in "" dblinit; do
  if [ -z "$v" ]; then L=""; else L=cuda_ray_tracer_amd/_build/ab/$v/libmirt.so; fi
  MIRT_LIB=$L PERF_COUNT=0 PERF_FRAMES=4 timeout -k 10 300 python3 tools/perf4.py tenthousand:1920:1080:16 redchair:1920:1080:16 spiral:1920:1080:16 >> $O/dbl.txt 2>&1
done
grep -v amdgpu.ids $O/dbl.txt
