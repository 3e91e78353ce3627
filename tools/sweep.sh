#!/bin/bash
# usage: sweep.sh VAR "v1 v2 ..." [bench args]   -- serial bench per value of an env knob, prints kernel_ms
R=$GRAFT_REPO_ROOT; cd $R
VAR=$1; VALS=$2; shift 2
for v in $VALS; do
  env $VAR=$v python bench.py --serial --cpu-step 0 --steps 8 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', 'ms/step', round(d['ms_per_step'],2), 'kernel_ms', round(d['roofline']['kernel_ms'],2), 'frac', round(d['roofline']['frac'],3))"
done
