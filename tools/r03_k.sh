#!/bin/bash
O=gpurun_out/${1:-r03k}; mkdir -p $O; rm -f $O/rq.txt
run() { PERF_COUNT=0 PERF_FRAMES=${F:-4} timeout -k 10 500 python3 tools/perf4.py "$@" >> $O/rq.txt 2>&1 || { cat $O/rq.txt; exit 1; }; }
R=redchair:1920:1080:16
run $R $R:qnodes=2,refill_k=52,init_k=48 $R:qnodes=2,refill_k=44,init_k=40 $R:qnodes=2,refill_k=52,init_k=48,reps=4 $R:qnodes=2,refill_k=36,init_k=32
F=2 run redchair:3840:2160:64:qnodes=2,refill_k=52,init_k=48,reps=4
grep -v amdgpu.ids $O/rq.txt
