#!/bin/bash
# usage: pmc.sh OUTNAME "C1 C2 ..." "C3 C4 ..."   -- one rocprofv3 --pmc pass per group over a short bench run
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 $R/bench.py --steps 2 --warmup 0 --cpu-step 0 > $OUT/g$i.json 2> $OUT/g$i.err || echo "group $i ($grp) failed"; echo "group $i done" >> $OUT/progress.txt
done
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
for f in sorted(glob.glob(out + "/g*/*/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if ("trace_kernel" in r["Kernel_Name"] or "shade_kernel" in r["Kernel_Name"]) and "<true" not in r["Kernel_Name"]:
            kn = "shade" if "shade" in r["Kernel_Name"] else "trace"
            agg[kn + ":" + r["Counter_Name"]].append(float(r["Counter_Value"]))
            continue
        if False:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(f"{k:52s} n={len(v)} sum={sum(v):.5g} mean={sum(v)/len(v):.5g}")
PY
