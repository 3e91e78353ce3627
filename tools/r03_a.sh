#!/bin/bash
# round-3 run A: GPU tests, then same-box A/B of the round-2 library against HEAD (whole frames, serial)
O=gpurun_out/${1:-r03a}; mkdir -p $O
timeout -k 10 420 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; rc=$?
echo "pytest rc=$rc" | tee -a $O/tests.log; tail -5 $O/tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
SPECS="tenthousand:1920:1080:16 spiral:1920:1080:16 redchair:1920:1080:16 redchair:3840:2160:64 synth:3840:2160:8"
MIRT_LIB=cuda_ray_tracer_amd/_build/ab/r02/libmirt.so timeout -k 10 300 python3 tools/perf4.py $SPECS > $O/perf_r02.txt 2>&1 || exit 1
cat $O/perf_r02.txt
timeout -k 10 300 python3 tools/perf4.py $SPECS redchair:1920:1080:16:qnodes=0 redchair:3840:2160:64:qnodes=0 synth:3840:2160:8:qnodes=0 tenthousand:1920:1080:16 > $O/perf_head.txt 2>&1 || exit 1
cat $O/perf_head.txt
