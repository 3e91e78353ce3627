#!/bin/bash
# knobs for one rank's 1/8 share of the headline frame (two frames in flight), hand-out by sample
O=gpurun_out/${1:-r03o}; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
probe() { env MIRT_$1=$2 timeout -k 10 120 python bench.py --share-of 8 --frames-in-flight $3 --cpu-step 0 --steps 24 --warmup 4 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('share 1/8 $1=$2 fif $3: ms/frame', round(d['ms_per_step'],3), 'kernel_ms', round(d['roofline']['kernel_ms'],2))"; }
for f in 2 3; do
  probe SCHED 2 $f
  for v in 2048 3072; do probe TRACE_WAVES $v $f; done
  for v in 5 7; do probe CHUNK_SHIFT $v $f; done
  for v in 4 20; do probe INIT_K $v $f; done
  for v in 24 40; do probe REFILL_K $v $f; done
done 2>&1 | tee $O/share_knobs_sched2.txt
