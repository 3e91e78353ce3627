#!/bin/bash
O=gpurun_out/${1:-r03v}; mkdir -p $O
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
PERF_COUNT=0 PERF_FRAMES=4 timeout -k 10 500 python3 tools/perf4.py tenthousand:1920:1080:16 spiral:1920:1080:16 redchair:1920:1080:16 redchair:3840:2160:64 redchair:3840:2160:64:sched=0 synth:3840:2160:8 synth:3840:2160:64 synth:3840:2160:64:sched=0 > $O/perf.txt 2>&1; grep -v amdgpu.ids $O/perf.txt
bash tools/share_fif.sh "8 4 2" "1 2" "24" > $O/share.txt 2>&1; cat $O/share.txt
