#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for reps in 3 4; do for rk in 36 40; do for lk in 4 6 8; do
  env MIRT_REPS=$reps MIRT_REFILL_K=$rk MIRT_LEAF_K=$lk python tools/perf3.py tenthousand spiral redchair 2>&1 | python3 -c "
import sys, re
out = []
for l in sys.stdin:
    m = re.match(r'(\w+) .*trace-kernels ([0-9.]+) ms', l)
    if m: out.append(m.group(1) + ' ' + m.group(2))
print('reps=$reps refill=$rk leaf=$lk:', ' | '.join(out))"
done; done; done
