#!/bin/bash
# round-3 run B: what moves the 2 M-primitive scene (latency-bound): wavefront path, five waves per SIMD, sibling prefetch
O=gpurun_out/${1:-r03b}; mkdir -p $O
S="synth:3840:2160:8"
timeout -k 10 300 python3 tools/perf4.py $S $S:qnodes=0 $S:wavefront=1 > $O/head.txt 2>&1 || { cat $O/head.txt; exit 1; }
cat $O/head.txt
MIRT_LIB=cuda_ray_tracer_amd/_build/ab/pf/libmirt.so timeout -k 10 300 python3 tools/perf4.py $S $S:qnodes=0 redchair:1920:1080:16 redchair:1920:1080:16:qnodes=0 tenthousand:1920:1080:16 > $O/pf.txt 2>&1 || { cat $O/pf.txt; exit 1; }
cat $O/pf.txt
MIRT_LIB=cuda_ray_tracer_amd/_build/ab/w5/libmirt.so timeout -k 10 300 python3 tools/perf4.py $S $S:qnodes=0 > $O/w5.txt 2>&1 || { cat $O/w5.txt; exit 1; }
cat $O/w5.txt
