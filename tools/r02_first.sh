#!/bin/bash
# first GPU run of round 2: the full GPU test suite, then counter passes on the unchanged round-1 kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02a; mkdir -p $O; cd $R
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
python3 tools/pmc_profile.py $O/pmc_trace_kernel.json --tag r02a 2>&1 | tee $O/pmc.log
python3 tools/pmc_profile.py $O/pmc_nosched.json --tag r02a_nosched --env MIRT_NO_SCHED=1 --groups write,fetch 2>&1 | tee -a $O/pmc.log
python3 tools/pmc_profile.py $O/pmc_spiral.json --tag r02a_spiral --scene spiral --groups ta,l1,write 2>&1 | tee -a $O/pmc.log
