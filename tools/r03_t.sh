#!/bin/bash
O=gpurun_out/${1:-r03t}; mkdir -p $O
export MIRT_LIB=cuda_ray_tracer_amd/_build/ab/stamps/libmirt.so MIRT_STAMPS=1
timeout -k 10 200 python bench.py --share-of 8 --serial --cpu-step 0 --steps 3 --warmup 2 2> $O/stamps_share8.txt > /dev/null; grep "mirt stamps" $O/stamps_share8.txt | tail -3
timeout -k 10 200 python bench.py --serial --cpu-step 0 --steps 3 --warmup 2 --headline-only 2> $O/stamps_whole.txt > /dev/null; grep "mirt stamps" $O/stamps_whole.txt | tail -2
