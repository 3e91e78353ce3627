#!/bin/bash
# round-3 run D: where the 2 M-primitive scene's wide kernel spends its time (counter passes), and a few knobs
O=gpurun_out/${1:-r03d}; mkdir -p $O
S="synth:3840:2160:8"
PERF_COUNT=0 timeout -k 10 500 python3 tools/perf4.py $S $S:refill_k=32 $S:refill_k=56 $S:leaf_k=4 $S:leaf_k=16 $S:reps=2 $S:reps=6 $S:batch_k=4 $S:batch_k=16 $S:chunk_shift=6 > $O/knobs.txt 2>&1 || { cat $O/knobs.txt; exit 1; }
cat $O/knobs.txt
timeout -k 10 900 python3 tools/pmc_profile.py $O/pmc_config5_wide.json --tag r03d --program tools/config5_probe.py 8 3 --label "synthetic 1M spheres + 1M triangles 3840x2160 8spp (BASELINE config 5 scene, one slab)" > $O/pmc.log 2>&1
tail -12 $O/pmc.log
