#!/bin/bash
# round-3 lock-in: GPU tests, the counter passes of the four workloads the bench line quotes (headline, spiral, redchair 4K x 64,
# the 2 M-primitive scene) -- installed into profiles/ on the box so that the bench line that follows finds them --, the default
# bench line, the serial one, and rocprofv3 kernel stats of the serial command -> gpurun_out/<TAG>/.
# Copy into profiles/ afterwards:  <TAG>_bench_*.json, <TAG>_*kernel_stats.csv, <TAG>_pmc_trace_kernel*.json
R=$GRAFT_REPO_ROOT; T=${1:-r03}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
if [ "$2" != "notests" ]; then
  timeout -k 10 800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -4 $O/pytest.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
fi
if [ "$3" != "nopmc" ]; then
  timeout -k 10 900 python3 tools/pmc_profile.py $O/pmc_trace_kernel.json --tag $T 2>&1 | tail -1 || exit 1
  timeout -k 10 900 python3 tools/pmc_profile.py $O/pmc_trace_kernel_spiral_1080p16.json --tag $T --scene spiral 2>&1 | tail -1 || exit 1
  timeout -k 10 900 python3 tools/pmc_profile.py $O/pmc_trace_kernel_redchair_4k64.json --tag $T --scene redchair --width 3840 --height 2160 --spp 64 2>&1 | tail -1 || exit 1
  timeout -k 10 1200 python3 tools/pmc_profile.py $O/pmc_trace_kernel_config5_scene_4k_8spp.json --tag $T --program tools/config5_probe.py 8 3 --label "synthetic 1M spheres + 1M triangles 3840x2160 8spp (BASELINE config 5 scene, one slab)" 2>&1 | tail -1
  rm -rf $R/gpurun_out/pmc_$T
  for f in $O/pmc_trace_kernel*.json; do cp $f profiles/${T}_$(basename $f); done
fi
timeout -k 10 600 python bench.py > $O/bench_tenthousand_1080p16.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cut -c1-1200 $O/bench_tenthousand_1080p16.json
timeout -k 10 200 python bench.py --serial --cpu-step 0 --headline-only > $O/bench_tenthousand_1080p16_serial.json 2>/dev/null || exit 1
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 6 --warmup 2 --cpu-step 0 --headline-only --serial > $O/bench_tenthousand_1080p16_under_rocprof.json 2> $O/bench_under_rocprof.err || { tail -5 $O/bench_under_rocprof.err; exit 1; }
cd $R
cp $O/prof/*/*kernel_stats.csv $O/bench_tenthousand_1080p16_kernel_stats.csv 2>/dev/null; head -4 $O/bench_tenthousand_1080p16_kernel_stats.csv
rm -rf $O/prof
