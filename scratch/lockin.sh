#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/t5.log 2>&1; tail -3 gpurun_out/t5.log
python bench.py > gpurun_out/bench2.json 2> gpurun_out/bench2.err; cat gpurun_out/bench2.json
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01b -- python3 $R/bench.py --steps 5 --warmup 1 --cpu-step 0 > $R/gpurun_out/bench_prof2.json 2> $R/gpurun_out/bench_prof2.err
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 0 --cpu-step 0 > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --steps 2 --warmup 0 --cpu-step 0 > /dev/null 2>&1
cd $R
cat gpurun_out/bench_prof2.json
for s in spiral redchair; do python bench.py --scene $s --cpu-step 0 --steps 5 2>/dev/null; done
