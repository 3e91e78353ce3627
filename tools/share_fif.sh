#!/bin/bash
# bench.py --share-of N for several frames-in-flight / step counts (what one rank of N does on its own GPU, no gather)
R=$GRAFT_REPO_ROOT; cd $R
for n in ${1:-8}; do for f in ${2:-1 2 3 4}; do for st in ${3:-10 24}; do
  python bench.py --share-of $n --frames-in-flight $f --cpu-step 0 --steps $st --warmup 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('share 1/$n frames_in_flight $f steps $st: ms/frame', round(d['ms_per_step'],2), 'kernel_ms', round(d['roofline']['kernel_ms'],2))"
done; done; done
