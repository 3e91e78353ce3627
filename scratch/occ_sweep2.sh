#!/bin/bash
for cfg in "4 24" "5 24" "6 16" "8 16"; do
  set -- $cfg
  MIRT_WAVES_PER_SIMD=$1 MIRT_STACK_LDS=$2 python -m cuda_ray_tracer_amd.build --force > /dev/null 2>&1
  MIRT_REFILL_K=32 timeout -k 10 120 python scratch/perf2.py tenthousand 2>&1 | grep -v amdgpu.ids | sed "s/^/W=$1 LDS=$2 /"
  MIRT_PROF=1 MIRT_REFILL_K=32 timeout -k 10 120 python scratch/perf2.py tenthousand 2>&1 | grep -E "mirt prof" | tail -1 | sed "s/^/W=$1 /"
done
