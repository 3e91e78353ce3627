#!/bin/bash
# usage: knobs.sh "scenes" VAR "v1 v2 ..." [VAR2 "..."]  -- perf3.py per value of an option (MIRT_<VAR> in the environment), one option at a time
R=$GRAFT_REPO_ROOT; cd $R
SCENES=$1; shift
while [ $# -ge 2 ]; do
  VAR=$1; VALS=$2; shift 2
  for v in $VALS; do
    env MIRT_$VAR=$v python tools/perf3.py $SCENES 2>&1 | python3 -c "
import sys, re
out = []
for l in sys.stdin:
    m = re.match(r'(\w+) .*trace-kernels ([0-9.]+) ms', l)
    if m: out.append(m.group(1) + ' ' + m.group(2))
print('$VAR=$v:', ' | '.join(out))"
  done
done
