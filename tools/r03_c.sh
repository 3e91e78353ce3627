#!/bin/bash
# round-3 run C: GPU tests, then the wide quantised walk against the binary one and the exact records (scenes with triangles)
O=gpurun_out/${1:-r03c}; mkdir -p $O
timeout -k 10 480 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; rc=$?
echo "pytest rc=$rc" | tee -a $O/tests.log; tail -5 $O/tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
S="synth:3840:2160:8"; R="redchair:1920:1080:16"; R4="redchair:3840:2160:64"
timeout -k 10 400 python3 tools/perf4.py $S $S:wide=0 $S:qnodes=0 $R $R:wide=0 $R:qnodes=0 $R4 $R4:qnodes=0 tri:1920:1080:16 tri:1920:1080:16:qnodes=0 > $O/perf.txt 2>&1 || { cat $O/perf.txt; exit 1; }
cat $O/perf.txt
