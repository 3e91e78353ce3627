#!/bin/bash
# rebuilds render.hip with different waves/SIMD caps on the GPU box and times the headline frame
for w in 2 3 4; do
  MIRT_WAVES_PER_SIMD=$w python -m cuda_ray_tracer_amd.build --force > /dev/null 2>&1
  for k in 24 32; do
    MIRT_REFILL_K=$k timeout -k 10 120 python scratch/perf2.py tenthousand 2>&1 | grep -v amdgpu.ids | sed "s/^/W=$w /"
  done
  MIRT_PROF=1 MIRT_REFILL_K=32 timeout -k 10 120 python scratch/perf2.py tenthousand 2>&1 | grep -E "mirt prof" | tail -1 | sed "s/^/W=$w /"
done
