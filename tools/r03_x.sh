#!/bin/bash
O=gpurun_out/${1:-r03x}; mkdir -p $O; rm -f $O/sweep3.txt
run() { PERF_COUNT=0 PERF_FRAMES=${F:-3} timeout -k 10 500 python3 tools/perf4.py "$@" >> $O/sweep3.txt 2>&1 || { cat $O/sweep3.txt; exit 1; }; }
F=2 run redchair:3840:2160:64:refill_k=64,init_k=64 tri:1920:1080:16 tri:1920:1080:16:refill_k=64,init_k=64
F=2 run synth:3840:2160:8 synth:3840:2160:8:refill_k=40,init_k=32 synth:3840:2160:8:refill_k=64,init_k=64 synth:3840:2160:8:refill_k=32,init_k=24 synth:3840:2160:8:refill_k=24,init_k=20
run tenthousand:1920:1080:16:refill_k=64,init_k=64 tenthousand:1920:1080:16:refill_k=36,init_k=10,batch_k=16 spiral:1920:1080:16:refill_k=28,init_k=10
grep -v amdgpu.ids $O/sweep3.txt
