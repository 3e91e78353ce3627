#!/bin/bash
# timing of the four bench scenes (default options) + the fuzz case that found the far-camera tie
O=gpurun_out/${1:-r03y}; mkdir -p $O; rm -f $O/perf.txt
run() { PERF_COUNT=${C:-0} PERF_FRAMES=${F:-4} timeout -k 10 500 python3 tools/perf4.py "$@" >> $O/perf.txt 2>&1 || { cat $O/perf.txt; exit 1; }; }
run tenthousand:1920:1080:16 spiral:1920:1080:16
F=3 run redchair:3840:2160:64 tri:1920:1080:16
F=2 run synth:3840:2160:8
grep -v amdgpu.ids $O/perf.txt
