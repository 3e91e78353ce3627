#!/bin/bash
# usage: share_knobs.sh N FIF VAR "v1 v2 ..." ...  -- bench.py --share-of N --frames-in-flight FIF per value of an option
R=$GRAFT_REPO_ROOT; cd $R
N=$1; F=$2; shift 2
while [ $# -ge 2 ]; do
  VAR=$1; VALS=$2; shift 2
  for v in $VALS; do
    env MIRT_$VAR=$v python bench.py --share-of $N --frames-in-flight $F --cpu-step 0 --steps 24 --warmup 4 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('share 1/$N fif $F $VAR=$v: ms/frame', round(d['ms_per_step'],2), 'kernel_ms', round(d['roofline']['kernel_ms'],2))"
  done
done
