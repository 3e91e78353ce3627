#!/bin/bash
# sphere-only kernels after the vetting went in: refill / init thresholds again
O=gpurun_out/${1:-r03r}; mkdir -p $O; rm -f $O/sweep.txt
for rk in 28 32 36 40 44; do for ik in 8 10 14; do
  PERF_COUNT=0 PERF_FRAMES=3 timeout -k 10 200 python3 tools/perf4.py tenthousand:1920:1080:16:refill_k=$rk,init_k=$ik spiral:1920:1080:16:refill_k=$rk,init_k=$ik >> $O/sweep.txt 2>&1 || { cat $O/sweep.txt; exit 1; }
done; done
grep -v amdgpu.ids $O/sweep.txt
