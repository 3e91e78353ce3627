#!/bin/bash
O=gpurun_out/${1:-r03w}; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R; rm -f $O/fb.txt
for v in "" fb0 fb1 fb3; do
  if [ -z "$v" ]; then L=""; else L=cuda_ray_tracer_amd/_build/ab/$v/libmirt.so; fi
  export MIRT_LIB=$L
  PERF_COUNT=0 PERF_FRAMES=4 timeout -k 10 300 python3 tools/perf4.py tenthousand:1920:1080:16 spiral:1920:1080:16 redchair:1920:1080:16 redchair:3840:2160:64 >> $O/fb.txt 2>&1
  timeout -k 10 120 python bench.py --share-of 8 --frames-in-flight 2 --cpu-step 0 --steps 24 --warmup 4 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('share 1/8 [$v] fif 2: ms/frame', round(d['ms_per_step'],3))" >> $O/fb.txt
done
grep -v amdgpu.ids $O/fb.txt
