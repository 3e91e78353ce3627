#!/bin/bash
O=gpurun_out/${1:-r03m2}; mkdir -p $O
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
PERF_COUNT=0 PERF_FRAMES=5 timeout -k 10 500 python3 tools/perf4.py tenthousand:1920:1080:16 spiral:1920:1080:16 redchair:1920:1080:16 > $O/perf.txt 2>&1; grep -v amdgpu.ids $O/perf.txt
bash tools/share_fif.sh "8" "1 2" "24" > $O/share.txt 2>&1; cat $O/share.txt
timeout -k 10 300 python bench.py --headline-only --cpu-step 0 > $O/bench.json 2> $O/bench.err; cut -c1-900 $O/bench.json
