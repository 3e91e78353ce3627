#!/usr/bin/env python3
"""Print the start/end of every trace kernel (ms, relative) from a rocprofv3 --kernel-trace CSV: shows which frames overlapped."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
rows = [r for r in csv.DictReader(open(f)) if "trace_kernel" in r["Kernel_Name"] or "order_kernel" in r["Kernel_Name"] or "resolve" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
last = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for r in rows[-last:]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    name = "trace" if "trace_kernel" in r["Kernel_Name"] else ("order" if "order" in r["Kernel_Name"] else "resolve")
    print(f"{name:8s} queue {r.get('Queue_Id','?'):>3s} start {s:9.3f} end {e:9.3f} dur {e-s:7.3f}  grid {r.get('Grid_Size','?')}")
