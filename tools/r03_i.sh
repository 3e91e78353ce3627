#!/bin/bash
O=gpurun_out/${1:-r03i}; mkdir -p $O; rm -f $O/initk3.txt
run() { PERF_COUNT=0 PERF_FRAMES=${F:-4} timeout -k 10 400 python3 tools/perf4.py "$@" >> $O/initk3.txt 2>&1 || { cat $O/initk3.txt; exit 1; }; }
R=redchair:1920:1080:16
run $R:init_k=44,refill_k=48 $R:init_k=44,refill_k=52 $R:init_k=48,refill_k=48 $R:init_k=52,refill_k=52 $R:init_k=52,refill_k=56 $R:init_k=48,refill_k=52,batch_k=12 $R:init_k=48,refill_k=52,batch_k=4 $R:init_k=48,refill_k=52,leaf_k=12 $R:init_k=48,refill_k=52,leaf_k=4
T=tenthousand:1920:1080:16
run $T:init_k=10,refill_k=28 $T:init_k=10,refill_k=36 $T:init_k=10,refill_k=40 $T:init_k=16,refill_k=40 $T:init_k=10,batch_k=12 $T:init_k=10,leaf_k=12
S=spiral:1920:1080:16
run $S:init_k=10,refill_k=28 $S:init_k=10,refill_k=36 $S:init_k=10,refill_k=40
grep -v amdgpu.ids $O/initk3.txt
