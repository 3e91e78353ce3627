#!/bin/bash
# usage: pmc_ab.sh "C1 C2 ..." lib1 lib2 ...  -- one rocprofv3 --pmc pass per library variant (serial bench, 3 frames), trace-kernel sums
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CTRS=$1; shift
i=0
for lib in "$@"; do
  i=$((i+1))
  OUT=$R/gpurun_out/pmc_ab/v$i
  rm -rf $OUT; mkdir -p $OUT
  cd /tmp
  MIRT_LIB=$R/$lib timeout -k 10 150 rocprofv3 --pmc $CTRS --output-format csv -d $OUT -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-step 0 --serial > $OUT/bench.json 2> $OUT/err.txt || echo "variant $lib failed"
  cd $R
  echo "== $lib"
  python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] + "/*/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "trace_kernel" in r["Kernel_Name"] and "<true" not in r["Kernel_Name"] and "Lb1ELb0" not in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print(f"  {k:40s} launches={len(v)} mean={sum(v)/len(v):.5g}")
PY
done
