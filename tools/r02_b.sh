#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02b; mkdir -p $O; cd $R
export PYTHONUNBUFFERED=1
timeout -k 10 120 python tools/perf3.py tenthousand 2>&1 | grep -v Warn | grep -v amdgpu.ids || { echo "perf3 failed/hung"; exit 1; }
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tee $O/pytest.log | tail -8
bash tools/knobs.sh "tenthousand spiral redchair" QNODES "1 0" QNODES "1 0"
