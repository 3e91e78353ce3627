#!/bin/bash
O=gpurun_out/${1:-r03o}; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R
probe() { # N VAR VAL FIF
  env MIRT_$2=$3 python bench.py --share-of $1 --frames-in-flight $4 --cpu-step 0 --steps 24 --warmup 4 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('share 1/$1 $2=$3 fif $4: ms/frame', round(d['ms_per_step'],3), 'kernel_ms', round(d['roofline']['kernel_ms'],2))"
}
for f in 1 2; do
  for v in 4 5 6; do probe 8 CHUNK_SHIFT $v $f; done
  for v in 5 6 7; do probe 4 CHUNK_SHIFT $v $f; done
  for v in 5 6 7 8; do probe 1 CHUNK_SHIFT $v $f; done
done 2>&1 | tee $O/share_chunk.txt
