#!/bin/bash
O=gpurun_out/${1:-r03u}; mkdir -p $O; R=$GRAFT_REPO_ROOT; cd $R; rm -f $O/sched2.txt
PERF_COUNT=0 PERF_FRAMES=5 timeout -k 10 300 python3 tools/perf4.py tenthousand:1920:1080:16 tenthousand:1920:1080:16:sched=2 spiral:1920:1080:16 spiral:1920:1080:16:sched=2 redchair:1920:1080:16 redchair:1920:1080:16:sched=2 synth:3840:2160:8 synth:3840:2160:8:sched=2 >> $O/sched2.txt 2>&1
grep -v amdgpu.ids $O/sched2.txt
probe() { env MIRT_SCHED=$2 timeout -k 10 120 python bench.py --share-of $1 --frames-in-flight $3 --cpu-step 0 --steps 24 --warmup 4 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('share 1/$1 sched=$2 fif $3: ms/frame', round(d['ms_per_step'],3), 'kernel_ms', round(d['roofline']['kernel_ms'],2))"; }
for n in 8 4; do for f in 1 2; do probe $n 1 $f; probe $n 2 $f; done; done 2>&1 | tee $O/share_sched2.txt
