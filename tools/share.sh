#!/bin/bash
# what one rank of N does on its own GPU (no gather): ms per frame of part 0 of an N-way stripe partition, by frames in flight
R=$GRAFT_REPO_ROOT; cd $R
for n in ${1:-8}; do for f in ${2:-1 2 4}; do
  python bench.py --share-of $n --frames-in-flight $f --cpu-step 0 --steps 24 --warmup 4 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('share 1/$n frames_in_flight $f: ms/frame', round(d['ms_per_step'],2), 'kernel_ms', round(d['roofline']['kernel_ms'],2))"
done; done
