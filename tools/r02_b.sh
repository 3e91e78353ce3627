#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02b; mkdir -p $O; cd $R
export PYTHONUNBUFFERED=1
timeout -k 10 120 python tools/perf3.py tenthousand 2>&1 | grep -v Warn | grep -v amdgpu.ids || { echo "perf3 failed/hung"; exit 1; }
timeout -k 10 500 python -m pytest tests -m gpu -x -q 2>&1 | tee $O/pytest.log | tail -8
bash tools/ab_run.sh "tenthousand spiral redchair" -
for t in 1 2; do MIRT_TRAVERSAL=$t timeout -k 10 200 python tools/synth.py 2>&1 | grep -v Warn | grep -v amdgpu.ids; done
