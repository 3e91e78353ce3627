#!/usr/bin/env python3
"""Static look at the trace kernel's traversal loop: compiles render.hip to gfx950 assembly with the given extra defines
and counts, per instruction class, what sits inside the depth-2 loop of trace_kernel<false, ...> (spill traffic there
is what to watch: the kernel runs at the 128-VGPR edge and small source changes move spills in and out of the loop --
a scratch reload of the lane's LDS stack address on every push and pop has cost 10-50 % more than once).

    python tools/hotloop.py [-DMACRO=V ...]

tests/test_codegen.py uses analyze() to keep spills out of the loop.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cuda_ray_tracer_amd import build as B   # noqa: E402


_COMPILED = {}


def _compile(defines):
    """render.hip -> (resource-usage remarks, gfx950 assembly); one compilation per set of defines and process."""
    if defines not in _COMPILED:
        out = os.path.join(tempfile.gettempdir(), "mirt_hotloop.s")
        cmd = [B._hipcc()] + [c for c in B.COMMON if c != "-fPIC"] + list(defines) + ["-x", "hip", "-S", "--cuda-device-only", os.path.join(B.CSRC, "render.hip"),
                                                                                   "-o", out, "-Rpass-analysis=kernel-resource-usage"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-2000:])
        _COMPILED[defines] = (r.stderr, open(out).read())
    return _COMPILED[defines]


def analyze(defines=(), kernel="trace_kernelILb0ELi8ELb1ELi7EE"):
    """Returns (resources, counts, spills): the kernel's resource usage lines, instruction counts by class inside the
    traversal loop, and [(position, instruction)] of every spill instruction in it.  `kernel`: the mangled instantiation,
    trace_kernel<COUNT, TABLES, QN, SPECX>: ...Li8ELb1ELi6EE = byte-indexed RNG tables, quantised nodes, no point lights /
    transparency / gi (the headline scene's kernel); ...Li8ELb0ELi2EE = 64-byte node records (scenes with triangles), no point
    lights (redchair.txt); ...Li0EE: the general kernels."""
    remarks, asm = _compile(tuple(defines))
    res, want = [], False
    for line in remarks.splitlines():
        if "Function Name" in line:
            want = kernel in line
        elif want and re.search(r"VGPRs:|SGPRs:|Spill|ScratchSize|Occupancy", line):
            res.append(line.split("remark:")[1].rsplit("[-Rpass", 1)[0].strip())
    body, inside = [], False
    for l in asm.splitlines():
        if re.match(r"^_ZN4mirt.*" + kernel + r".*:", l):
            inside = True
            continue
        if inside and "s_endpgm" in l:
            break
        if inside:
            body.append(l)
    # the traversal loop is the last depth-2 loop of the kernel: its blocks carry "in Loop: Header=<label> Depth=2", the
    # blocks of its inner loops (plane loops, step loop) "Header=<inner> Depth=3", where <inner> is a loop header whose
    # comment names <label> as its depth-2 parent
    text = "\n".join(body)

    def members(hdr):
        inner = set()
        for m in re.finditer(r"^\.L(BB\d+_\d+):((?:[^\n]*\n\s+;[^\n]*)*)", text, re.M):
            if ("Parent Loop " + hdr + " Depth=2") in m.group(0) and "Inner Loop Header: Depth=3" in m.group(0):
                inner.add(m.group(1))

        def in_loop(l):
            if ("Header=" + hdr + " Depth=2") in l or l.startswith(".L" + hdr + ":"):
                return True
            m = re.search(r"Header=(BB\d+_\d+) Depth=3", l)
            if m and m.group(1) in inner:
                return True
            m = re.match(r"^\.L(BB\d+_\d+):", l)
            return bool(m and m.group(1) in inner)
        return in_loop

    # the traversal loop: the depth-2 loop (with its inner loops) that pushes onto the LDS stack
    in_loop = None
    for hdr in dict.fromkeys(re.findall(r"Header=(BB\d+_\d+) Depth=2", text)):
        f, cur, hit = members(hdr), None, False
        for l in body:
            if re.match(r"^\.LBB\d+_\d+:", l) or re.match(r"^; %bb", l):
                cur = f(l)
            elif cur and "ds_write_b32" in l:
                hit = True
                break
        if hit:
            in_loop = f
    assert in_loop is not None, "traversal loop not found"
    cnt, spills, cur, pos = collections.Counter(), [], None, 0
    for l in body:
        if re.match(r"^\.LBB\d+_\d+:", l) or re.match(r"^; %bb", l):
            cur = "loop" if in_loop(l) else None
            continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith(".") or cur != "loop":
            continue
        op = t.split()[0]
        pos += 1
        if op.startswith("scratch_"): k = "scratch"
        elif op.startswith("v_readlane") or op.startswith("v_writelane"): k = "sgpr-spill"
        elif op.startswith(("global_", "flat_", "buffer_")): k = "vmem"
        elif op.startswith("ds_"): k = "lds"
        elif op.startswith("v_"): k = "valu"
        elif op.startswith("s_waitcnt"): k = "waitcnt"
        elif op.startswith(("s_cbranch", "s_branch")): k = "branch"
        elif op.startswith("s_"): k = "salu"
        else: k = "other"
        cnt[k] += 1
        if k in ("scratch", "sgpr-spill"):
            spills.append((pos, t.split(";")[0].strip()))
    return res, dict(cnt), spills


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a.startswith("-")]
    kern = [a for a in sys.argv[1:] if not a.startswith("-")]
    res, cnt, spills = analyze(args, *kern)
    print("\n".join(res))
    print("static instruction counts inside the traversal loop:", cnt)
    print("spill instructions in the loop (position: instruction):")
    for q, t in spills:
        print("  %5d  %s" % (q, t))
