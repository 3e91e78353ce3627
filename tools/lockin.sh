#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python bench.py > gpurun_out/bench3.json 2> gpurun_out/bench3.err; cat gpurun_out/bench3.json
python bench.py --serial --cpu-step 0 > gpurun_out/bench3_serial.json 2>/dev/null; cat gpurun_out/bench3_serial.json
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01c -- python3 $R/bench.py --steps 6 --warmup 2 --cpu-step 0 > $R/gpurun_out/bench_prof3.json 2> $R/gpurun_out/bench_prof3.err
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch3 -- python3 $R/bench.py --steps 3 --warmup 2 --cpu-step 0 --serial > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write3 -- python3 $R/bench.py --steps 3 --warmup 2 --cpu-step 0 --serial > /dev/null 2>&1
cd $R
cat gpurun_out/bench_prof3.json
for s in spiral redchair; do python bench.py --scene $s --cpu-step 0 --steps 6 2>/dev/null; done
