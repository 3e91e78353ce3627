#!/bin/bash
for sw in 2 4; do
  MIRT_WF_SHADE_WAVES=$sw python -m cuda_ray_tracer_amd.build --force > /dev/null 2>&1
  if [ $sw = 2 ]; then MIRT_WAVEFRONT=1 timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2; fi
  for pool in 4194304 8388608; do
  MIRT_WAVEFRONT=1 MIRT_WF_POOL=$pool timeout -k 10 120 python scratch/perf3.py tenthousand 2>&1 | grep -v amdgpu.ids | sed "s/^/SW=$sw /"
  done
done
