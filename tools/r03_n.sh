#!/bin/bash
O=gpurun_out/${1:-r03n}; mkdir -p $O
PERF_COUNT=0 PERF_FRAMES=5 timeout -k 10 500 python3 tools/perf4.py tenthousand:1920:1080:16 spiral:1920:1080:16 redchair:1920:1080:16 > $O/perf.txt 2>&1; grep -v amdgpu.ids $O/perf.txt
PERF_COUNT=0 PERF_FRAMES=2 timeout -k 10 500 python3 tools/perf4.py redchair:3840:2160:64 synth:3840:2160:8 synth:3840:2160:64 > $O/perf2.txt 2>&1; grep -v amdgpu.ids $O/perf2.txt
bash tools/share_fif.sh "8 4 2" "1 2" "24" > $O/share.txt 2>&1; cat $O/share.txt
