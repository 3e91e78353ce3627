#!/bin/bash
# frames in flight x hand-out order, whole frame and a 1/8 share
O=gpurun_out/${1:-r03fif}; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
probe() { env MIRT_SCHED=$2 timeout -k 10 120 python bench.py --share-of $1 --frames-in-flight $3 --cpu-step 0 --steps 24 --warmup 4 --headline-only 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('share 1/$1 sched=$2 fif $3: ms/frame', round(d['ms_per_step'],3), 'kernel_ms', round(d['roofline']['kernel_ms'],2))"; }
for n in 1 8; do for s in 1 2; do for f in 1 2 3; do probe $n $s $f; done; done; done 2>&1 | tee $O/fif.txt
