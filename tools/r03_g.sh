#!/bin/bash
# round-3 run G: non-temporal hints on the pending list / the sample buffer (A/B), then the stripe-share probe
O=gpurun_out/${1:-r03g}; mkdir -p $O
S="redchair:3840:2160:64 redchair:1920:1080:16 tenthousand:1920:1080:16"
for v in "" ntp nts ntb; do
  if [ -z "$v" ]; then L=""; else L=cuda_ray_tracer_amd/_build/ab/$v/libmirt.so; fi
  MIRT_LIB=$L PERF_COUNT=0 timeout -k 10 300 python3 tools/perf4.py $S >> $O/nt.txt 2>&1 || { cat $O/nt.txt; exit 1; }
done
grep -v amdgpu.ids $O/nt.txt
bash tools/share_fif.sh "8" "1 2 3" "24" > $O/share.txt 2>&1; cat $O/share.txt
