#!/bin/bash
# usage: lockin.sh TAG -- the measurement set behind profiles/<TAG>_*: default and serial bench, rocprofv3 kernel stats of the
# same command, FETCH_SIZE / WRITE_SIZE passes, the other bundled scenes, the multi-GPU share probe
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-lock}
O=$R/gpurun_out/$T; mkdir -p $O
cd $R
python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json
python bench.py --serial --cpu-step 0 > $O/bench_serial.json 2>/dev/null; cat $O/bench_serial.json
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 6 --warmup 2 --cpu-step 0 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 2 --cpu-step 0 --serial > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 3 --warmup 2 --cpu-step 0 --serial > /dev/null 2>&1
cd $R
cat $O/bench_under_rocprof.json
cp $O/prof/*/*kernel_stats.csv $O/kernel_stats.csv
python3 - $O <<'PY'
import csv, glob, sys, json
o = sys.argv[1]
res = {}
for name in ("fetch", "write"):
    vals = []
    for f in glob.glob(f"{o}/pmc_{name}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "trace_kernel" in r["Kernel_Name"] and "Lb1ELb0" not in r["Kernel_Name"] and "<true" not in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    res[name.upper() + "_SIZE_KB_per_launch"] = vals
print(json.dumps(res))
json.dump(res, open(f"{o}/pmc_raw.json", "w"))
PY
for s in spiral redchair; do python bench.py --scene $s --cpu-step 0 --steps 6 2>/dev/null | tee $O/bench_$s.json; done
tools/share.sh '2 4 8' '1 2 4' | tee $O/share.txt
