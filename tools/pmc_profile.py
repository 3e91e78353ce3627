#!/usr/bin/env python3
"""Counter evidence for the trace kernel: one `rocprofv3 --pmc` pass per counter group over a short serial bench run
(each pass its own child process, the program directly after `--`), summarised per launch into one JSON.

    python3 tools/pmc_profile.py OUT_JSON [--scene tenthousand] [--tag r02] [--env KEY=VAL ...] [--groups a,b,...]

Run on the GPU box (gpurun); copy the JSON into profiles/.  Derived figures:
  ta_busy            TA_TA_BUSY_sum / (256 address units x launch cycles); launch cycles = GRBM_GUI_ACTIVE / 8 (summed over the XCDs)
  l1_requests        TCP_TOTAL_CACHE_ACCESSES_sum per launch (one per active lane per vector-memory instruction)
  ta_floor_ms        l1_requests / (256 CUs x 2.4 GHz): the address units retire about one lane request per cycle per CU
  l1_hit_rate        1 - TCP_TCC_READ_REQ_sum / TCP_TOTAL_CACHE_ACCESSES_sum
  l2_hit_rate        TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
  valu_issue_frac    SQ_INSTS_VALU x 4 issue cycles / (1024 SIMDs x launch cycles)
  active_lanes       SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64)
  hbm_bytes          2 x FETCH_SIZE + WRITE_SIZE (KB -> B; gfx950 tallies 128-B read requests as 64 B, MI355X_MICROARCH.md) -- an UPPER bound for
                     scattered record fetches: the guide validates the doubling for wide streaming reads only
  rdreq (group)      TCC_EA0_RDREQ_sum / _32B_sum: the L2's read requests to the fabric and how many of them are 32-byte ones; the others are
                     64-byte tallies that stand for 64 or 128 bytes -> hbm_read_bytes_low / _high bracket the truth; bench.py prices the LOW one
The JSON records the git HEAD and a hash of csrc/ + this file: bench.py marks a counter block stale when the hash is not the tree's.
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = {
    "ta": "GRBM_GUI_ACTIVE TA_TA_BUSY_sum",
    "l1": "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum",
    "l2": "TCC_HIT_sum TCC_MISS_sum",
    "sq": "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES",
    "fetch": "FETCH_SIZE",
    "write": "WRITE_SIZE",
    # (not in the default set) instruction cache and instruction mix
    "icache": "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH",
    "mix": "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU",
    "wrreq": "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum",      # write requests L2 -> memory, and how many of them are 64-byte ones
    "rdreq": "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum",   # read requests L2 -> fabric by size; all requests the L2 received
}
DEFAULT_GROUPS = "ta,l1,l2,sq,fetch,write,rdreq"
CLOCK_HZ = 2.4e9
CUS = 256
XCDS = 8          # GRBM_GUI_ACTIVE is summed over the eight XCDs: cycles of the launch = sum / 8


def source_hash():
    sys.path.insert(0, ROOT)
    from cuda_ray_tracer_amd import build as B
    return B.source_hash()


def git_head():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--scene", default="tenthousand")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--tag", default="r02")
    ap.add_argument("--env", action="append", default=[])
    ap.add_argument("--groups", default=DEFAULT_GROUPS)
    ap.add_argument("--kernel", default="trace_kernel")
    ap.add_argument("--program", nargs="+", default=None, help="profile `python3 PROGRAM ARGS...` instead of bench.py (it prints one JSON line with roofline.kernel_ms)")
    ap.add_argument("--label", default=None, help="workload label of the output (with --program)")
    args = ap.parse_args()

    env = dict(os.environ)
    env["TMPDIR"] = "/tmp"
    for kv in args.env:
        k, v = kv.split("=", 1)
        env[k] = v
    scratch = os.path.join(ROOT, "gpurun_out", "pmc_" + args.tag)
    shutil.rmtree(scratch, ignore_errors=True)
    os.makedirs(scratch, exist_ok=True)
    bench = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--cpu-step", "0", "--serial", "--headline-only",
             "--scene", args.scene, "--width", str(args.width), "--height", str(args.height), "--spp", str(args.spp)]
    workload = f"{args.scene}.txt {args.width}x{args.height} {args.spp}spp"
    if args.program:
        bench = [sys.executable, os.path.join(ROOT, args.program[0])] + args.program[1:]
        workload = args.label or " ".join(args.program)
    raw = {}
    launches = {}
    kernel_ms = None
    for g in args.groups.split(","):
        d = os.path.join(scratch, g)
        cmd = ["rocprofv3", "--pmc"] + GROUPS[g].split() + ["--output-format", "csv", "-d", d, "--"] + bench
        try:
            p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
        except subprocess.TimeoutExpired:
            print(f"group {g}: timeout", flush=True)
            continue
        open(os.path.join(scratch, g + ".err"), "wb").write(p.stderr[-20000:])
        for line in p.stdout.decode().splitlines():
            if line.startswith("{"):
                try:
                    kernel_ms = json.loads(line)["roofline"]["kernel_ms"]
                except Exception:
                    pass
        agg = collections.defaultdict(list)
        for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                kn = r["Kernel_Name"]
                # the plain instantiation only (not the counters variant <true, ...>)
                if args.kernel in kn and "<true" not in kn:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            raw[k] = sum(v) / len(v)
            launches[k] = len(v)
        print(f"group {g}: rc={p.returncode} " + " ".join(f"{k}={raw[k]:.5g}" for k in agg), flush=True)

    out = {"workload": workload, "kernel": args.kernel, "tag": args.tag, "csrc_sha16": source_hash(), "git_head": git_head(),
           "command": "rocprofv3 --pmc <group> --output-format csv -- python3 " + " ".join([os.path.relpath(bench[1], ROOT)] + bench[2:]) + " (one pass per group)",
           "env": args.env, "per_launch": raw, "launches_averaged": launches, "kernel_ms_under_pmc": kernel_ms, "derived": {}}
    dv = out["derived"]
    g = raw.get("GRBM_GUI_ACTIVE")
    if g:
        g = g / XCDS
    if g and "TA_TA_BUSY_sum" in raw:
        dv["ta_busy"] = raw["TA_TA_BUSY_sum"] / (CUS * g)
        dv["gui_active_ms"] = g / CLOCK_HZ * 1e3
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in raw:
        dv["l1_requests"] = raw["TCP_TOTAL_CACHE_ACCESSES_sum"]
        dv["ta_floor_ms"] = raw["TCP_TOTAL_CACHE_ACCESSES_sum"] / (CUS * CLOCK_HZ) * 1e3
        if "TCP_TCC_READ_REQ_sum" in raw:
            dv["l1_hit_rate"] = 1.0 - raw["TCP_TCC_READ_REQ_sum"] / raw["TCP_TOTAL_CACHE_ACCESSES_sum"]
    if "TCC_HIT_sum" in raw and "TCC_MISS_sum" in raw:
        dv["l2_hit_rate"] = raw["TCC_HIT_sum"] / max(raw["TCC_HIT_sum"] + raw["TCC_MISS_sum"], 1.0)
    if "SQ_ACTIVE_INST_VALU" in raw:
        if "SQ_BUSY_CYCLES" in raw and g:
            dv["valu_issue_frac"] = raw["SQ_INSTS_VALU"] * 4.0 / (4 * CUS * g)          # 4 issue cycles per wave instruction
            dv["valu_active_frac"] = raw["SQ_ACTIVE_INST_VALU"] * 4.0 / (4 * CUS * g)    # quad-cycles with a VALU instruction in flight
        if "SQ_THREAD_CYCLES_VALU" in raw:
            dv["active_lanes"] = raw["SQ_THREAD_CYCLES_VALU"] / (raw["SQ_ACTIVE_INST_VALU"] * 64.0) * 64.0
        if "SQ_WAVE_CYCLES" in raw and "SQ_WAIT_ANY" in raw:
            dv["wave_wait_frac"] = raw["SQ_WAIT_ANY"] / raw["SQ_WAVE_CYCLES"]
    if "FETCH_SIZE" in raw or "WRITE_SIZE" in raw:
        dv["hbm_read_bytes"] = 2.0 * raw.get("FETCH_SIZE", 0.0) * 1024.0
        dv["hbm_write_bytes"] = raw.get("WRITE_SIZE", 0.0) * 1024.0
        dv["hbm_bytes_per_launch"] = dv["hbm_read_bytes"] + dv["hbm_write_bytes"]
        out["hbm_bytes_per_launch"] = dv["hbm_bytes_per_launch"]
    if "TCC_EA0_RDREQ_sum" in raw:
        n, n32 = raw["TCC_EA0_RDREQ_sum"], raw.get("TCC_EA0_RDREQ_32B_sum", 0.0)
        dv["rdreq"] = n
        dv["rdreq_32B"] = n32
        dv["hbm_read_bytes_low"] = 32.0 * n32 + 64.0 * (n - n32)        # every other request counted at the 64 bytes it is tallied as
        dv["hbm_read_bytes_high"] = 32.0 * n32 + 128.0 * (n - n32)      # ... at 128 bytes (what the guide measured for wide streaming reads)
        if "TCC_REQ_sum" in raw:
            dv["l2_requests"] = raw["TCC_REQ_sum"]
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps(dv))


if __name__ == "__main__":
    main()
