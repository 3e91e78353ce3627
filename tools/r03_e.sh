#!/bin/bash
# round-3 run E: GPU tests, then knob sweep of the wide kernel on the 2 M-primitive scene
O=gpurun_out/${1:-r03e}; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; rc=$?
echo "pytest rc=$rc" | tee -a $O/tests.log; tail -5 $O/tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
S="synth:3840:2160:8"
PERF_COUNT=0 timeout -k 10 500 python3 tools/perf4.py $S $S:refill_k=16 $S:refill_k=24 $S:refill_k=28 $S:refill_k=36 $S:refill_k=24,batch_k=4 $S:refill_k=28,batch_k=4 $S:refill_k=32,batch_k=4 $S:refill_k=32,batch_k=2 $S:refill_k=28,batch_k=4,drain_lanes=32 $S:refill_k=28,batch_k=4,leaf_k=6 > $O/knobs.txt 2>&1 || { cat $O/knobs.txt; exit 1; }
cat $O/knobs.txt
