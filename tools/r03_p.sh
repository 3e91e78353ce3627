#!/bin/bash
O=gpurun_out/${1:-r03p}; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 600 python3 tools/pmc_profile.py $O/pmc_share8.json --tag r03p --groups ta,sq --program tools/share_probe.py 8 serial --label "tenthousand 1080p16, part 0 of 8 (serial)" 2>&1 | tail -3
rm -rf gpurun_out/pmc_r03p
