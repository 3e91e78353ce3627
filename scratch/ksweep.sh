#!/bin/bash
sed -i 's/for parts in (8,):/for parts in (1, 8):/; s/    for nfl in (2, 3, 4):/    for nfl in (1,):/' scratch/overlap_probe.py
for cfg in "MIRT_REFILL_K=32" "MIRT_REFILL_K=24" "MIRT_REFILL_K=40" "MIRT_REFILL_K=48" "MIRT_DRAIN_LANES=8" "MIRT_DRAIN_LANES=32" "MIRT_DRAIN_LANES=48" "MIRT_BATCH_K=4" "MIRT_BATCH_K=12"; do
  env $cfg timeout -k 10 200 python scratch/overlap_probe.py 2>&1 | grep -v amdgpu.ids | sed "s/^/$cfg /"
done
