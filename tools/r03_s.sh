#!/bin/bash
# spiral.txt: steps per header pass x primitive-test threshold
O=gpurun_out/${1:-r03s}; mkdir -p $O; rm -f $O/sweep.txt
for r in 2 3 4 5; do for k in 4 8 12; do
  PERF_COUNT=0 PERF_FRAMES=3 timeout -k 10 200 python3 tools/perf4.py spiral:1920:1080:16:reps=$r,leaf_k=$k >> $O/sweep.txt 2>&1 || { cat $O/sweep.txt; exit 1; }
done; done
grep -v amdgpu.ids $O/sweep.txt
