#!/bin/bash
# tests, bench (default / serial), kernel-trace stats and counter passes of the current build -> gpurun_out/r02c
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r02c}; mkdir -p $O; cd $R
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout -k 10 120 python tools/perf3.py tenthousand 2>&1 | grep -v Warn | grep -v amdgpu.ids || { echo "perf3 failed/hung"; exit 1; }
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tee $O/pytest.log | tail -6
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-1500 $O/bench.json
timeout -k 10 200 python bench.py --serial --cpu-step 0 --headline-only > $O/bench_serial.json 2>/dev/null; cut -c1-400 $O/bench_serial.json
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 6 --warmup 2 --cpu-step 0 --headline-only > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cd $R
cp $O/prof/*/*kernel_stats.csv $O/kernel_stats.csv 2>/dev/null; head -5 $O/kernel_stats.csv
python3 tools/pmc_profile.py $O/pmc_trace_kernel.json --tag ${1:-r02c} 2>&1 | tail -3
