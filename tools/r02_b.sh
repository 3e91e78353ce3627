#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02b; mkdir -p $O; cd $R
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
for t in 1 0 2; do echo "== traversal $t"; MIRT_TRAVERSAL=$t python tools/perf3.py tenthousand spiral redchair 2>&1 | grep -v Warn | grep -v amdgpu.ids; done
