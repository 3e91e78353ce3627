#!/usr/bin/env python3
"""Static look at the trace kernel's traversal loop: compiles render.hip to gfx950 assembly with the given extra defines
and counts, per instruction class, what sits inside the depth>=2 loop of trace_kernel<false,false> (spill traffic there
is what to watch: the kernel runs at the 128-VGPR edge and small source changes move spills in and out of the loop).

    python tools/hotloop.py [-DMACRO=V ...]
"""
import os, re, subprocess, sys, tempfile, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cuda_ray_tracer_amd import build as B
out = os.path.join(tempfile.gettempdir(), "mirt_hotloop.s")
cmd = [B._hipcc()] + [c for c in B.COMMON if c != "-fPIC"] + sys.argv[1:] + ["-x", "hip", "-S", "--cuda-device-only", os.path.join(B.CSRC, "render.hip"), "-o", out, "-Rpass-analysis=kernel-resource-usage"]
r = subprocess.run(cmd, capture_output=True, text=True)
want = False
for line in r.stderr.splitlines():
    if "Function Name" in line:
        want = "trace_kernelILb0ELb0" in line
    elif want and re.search(r"VGPRs:|SGPRs:|Spill|ScratchSize|Occupancy", line):
        print(line.split("remark:")[1].split("[")[0].strip())
txt = open(out).read().splitlines()
# the traversal loop is the last depth-2 loop of the kernel; its blocks carry "Header=<label> Depth=2" (inner loops: Depth=3)
inside = False
body = []
for l in txt:
    if re.match(r"^_ZN4mirt.*trace_kernelILb0ELb0.*:", l):
        inside = True; continue
    if inside and "s_endpgm" in l:
        break
    if inside:
        body.append(l)
hdrs = re.findall(r"Header=(BB\d+_\d+) Depth=2", "\n".join(body))
hdr = hdrs[-1]
cnt = collections.Counter()
cur = None
pos = 0; fetch_at = None; spills = []
for l in body:
    if re.match(r"^\.LBB\d+_\d+:", l) or re.match(r"^; %bb", l):
        if ("Header=" + hdr + " Depth=2") in l or l.startswith(".L" + hdr + ":"): cur = "loop"
        elif "Depth=3" in l or "Depth 3" in l: cur = cur if cur == "loop3" or cur == "loop" and False else ("loop3" if cur in ("loop", "loop3") else None)
        else: cur = None
        continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    if cur != "loop":
        continue
    op = t.split()[0]
    pos += 1
    if op.startswith("global_load_dwordx4") and fetch_at is None and ", v9" in t and "offset" not in t: fetch_at = pos
    if op.startswith("scratch_") or op.startswith("v_readlane") or op.startswith("v_writelane"): spills.append((pos, t.split(";")[0].strip()))
    if op.startswith("scratch_"): k = "scratch"
    elif op.startswith("v_readlane") or op.startswith("v_writelane"): k = "sgpr-spill"
    elif op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_"): k = "vmem"
    elif op.startswith("ds_"): k = "lds"
    elif op.startswith("v_"): k = "valu"
    elif op.startswith("s_waitcnt"): k = "waitcnt"
    elif op.startswith("s_cbranch") or op.startswith("s_branch"): k = "branch"
    elif op.startswith("s_"): k = "salu"
    else: k = "other"
    cnt[k] += 1
print("static instruction counts inside the traversal loop:", dict(cnt))
print("record fetch at loop instruction", fetch_at, "of", pos, "; spill instructions in the loop (position: instruction):")
for q, t in spills: print("  %5d%s %s" % (q, " *" if fetch_at and q > fetch_at else "  ", t))
