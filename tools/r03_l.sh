#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03l}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tl -- python3 $R/bench.py --steps 4 --warmup 2 --cpu-step 0 --headline-only --serial > $O/bench.json 2> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
cd $R
python3 - $O <<'PY'
import csv, glob, sys
o = sys.argv[1]
ev = []
for f in glob.glob(f"{o}/tl/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
for f in glob.glob(f"{o}/tl/*/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
ev.sort()
t0 = ev[-60][0] if len(ev) > 60 else ev[0][0]
for s, e, n in ev[-60:]:
    print(f"{(s - t0) / 1e6:9.3f} {(e - t0) / 1e6:9.3f} {(e - s) / 1e3:9.1f} us  {n}")
PY
