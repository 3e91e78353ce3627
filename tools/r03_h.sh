#!/bin/bash
O=gpurun_out/${1:-r03h}; mkdir -p $O
S="redchair:1920:1080:16 redchair:1920:1080:16:qnodes=2 redchair:3840:2160:64 redchair:3840:2160:64:qnodes=2"
for v in "" w3; do
  if [ -z "$v" ]; then L=""; else L=cuda_ray_tracer_amd/_build/ab/$v/libmirt.so; fi
  MIRT_LIB=$L PERF_COUNT=0 timeout -k 10 300 python3 tools/perf4.py $S >> $O/w3.txt 2>&1 || { cat $O/w3.txt; exit 1; }
done
grep -v amdgpu.ids $O/w3.txt
