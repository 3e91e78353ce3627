#!/bin/bash
O=gpurun_out/${1:-r03j}; mkdir -p $O; rm -f $O/synth.txt
run() { PERF_COUNT=0 PERF_FRAMES=${F:-2} timeout -k 10 500 python3 tools/perf4.py "$@" >> $O/synth.txt 2>&1 || { cat $O/synth.txt; exit 1; }; }
S=synth:3840:2160:8
run $S $S:batch_k=4 $S:batch_k=6 $S:refill_k=24,init_k=8 $S:refill_k=26,init_k=8,batch_k=4 $S:leaf_k=6 $S:init_k=4 $S:init_k=12 $S:drain_lanes=8 $S:reps=5 $S:reps=3
F=1 run synth:3840:2160:64 synth:3840:2160:64:slab_log2=28
grep -v amdgpu.ids $O/synth.txt
