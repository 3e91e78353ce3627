"""The trace kernel runs at the 128-VGPR edge of 4 waves per SIMD, and what the compiler decides to spill inside the
traversal loop has moved its speed by 10-50 % more than once (a scratch reload of the lane's LDS stack address on every
push and pop).  This test compiles render.hip to assembly (no GPU needed) and keeps register spills to scratch out of
the loop, apart from the pointer reloads on the global stack-spill path that the bundled scenes never take and at most
one other reload."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


import pytest


@pytest.mark.parametrize("kernel", ["trace_kernelILb0ELi8ELb1ELi7EE", "trace_kernelILb0ELi8ELb1ELi1EE", "trace_kernelILb0ELi8ELb1ELi6EE", "trace_kernelILb0ELi8ELb0ELi6EE",
                                    "trace_kernelILb0ELi8ELb0ELi2EE", "trace_kernelILb0ELi8ELb0ELi0EE"])
def test_no_scratch_reloads_on_the_traversal_loop_hot_path(kernel):
    """Both node formats of the 16-spp kernel -- quantised records (sphere-only scenes: the headline) and 64-byte records -- in
    each of their specialisations (render.hip, SPECX)."""
    import hotloop
    res, cnt, spills = hotloop.analyze(kernel=kernel)
    occ = [r for r in res if r.startswith("Occupancy")]
    assert occ and occ[0].rsplit(":", 1)[1].strip() == "4", res      # 4 waves per SIMD
    scratch = [t for _, t in spills if t.startswith("scratch_")]
    assert not [t for t in scratch if t.startswith("scratch_store")], scratch
    # the 64-bit spill-area pointer on the global stack-spill path (never taken by the bundled scenes), once per step-loop copy
    assert len([t for t in scratch if t.startswith("scratch_load_dwordx2")]) <= 4, (cnt, scratch)
    # at most one single-register reload (today: an LDS address of the shade phase restored after the triangle test used
    # its register); the expensive cases were two of them, on the push and on the pop
    assert len([t for t in scratch if not t.startswith("scratch_load_dwordx2")]) <= 1, (cnt, scratch)


def _kernel_instructions(src, kernel):
    """Compiles csrc/<src> to gfx950 assembly (no GPU needed) and returns the instructions of `kernel`."""
    import re
    import subprocess
    import tempfile
    from cuda_ray_tracer_amd import build as B
    out = os.path.join(tempfile.gettempdir(), "mirt_codegen_" + src + ".s")
    cmd = [B._hipcc()] + [c for c in B.COMMON if c != "-fPIC"] + ["-x", "hip", "-S", "--cuda-device-only", os.path.join(B.CSRC, src), "-o", out]
    subprocess.run(cmd, check=True, capture_output=True)
    body, inside = [], False
    for l in open(out).read().splitlines():
        if re.match(r"^_ZN4mirt.*" + kernel + r".*:", l):
            inside = True
            continue
        if inside and "s_endpgm" in l:
            break
        if inside:
            t = l.split(";")[0].strip()
            if t and not t.startswith("."):
                body.append(t)
    assert body, kernel
    return body


def test_refit_hand_off_keeps_its_write_through_stores_and_bypassing_loads():
    """refit_pack_kernel hands boxes from the first thread at a node to the second without cache-wide fences: every box word
    is stored write-through (sc1), the stores are drained (s_waitcnt vmcnt(0)) before the arrival counter is bumped, and the
    sibling's box is read with L1-bypassing (sc1) loads.  The reference has a plain race there (lbvh_builder.cu:343-359,381).
    The source expresses this with relaxed agent-scope atomics + an inline s_waitcnt; this test pins what the compiler makes
    of it, so that a toolchain update cannot silently bring the race back."""
    ins = _kernel_instructions("lbvh_build.hip", "refit_pack_kernel")
    dword_stores = [t for t in ins if t.startswith("global_store_dword ")]
    assert len(dword_stores) >= 12 and all(t.endswith(" sc1") or " sc1" in t for t in dword_stores), dword_stores   # leaf box + merged box
    sc1_loads = [t for t in ins if t.startswith("global_load_dword ") and " sc1" in t]
    assert len(sc1_loads) >= 12, sc1_loads                                                                          # both children's boxes
    atomics = [i for i, t in enumerate(ins) if t.startswith("global_atomic_add")]
    assert atomics
    for i in atomics:
        # walking back from the counter add, the first memory-related instruction must be the drain
        drained = False
        for t in reversed(ins[:i]):
            if t.startswith("s_waitcnt") and "vmcnt(0)" in t:
                drained = True
                break
            if t.startswith(("global_", "flat_", "buffer_", "scratch_")):
                break
        assert drained, ins[max(0, i - 8):i + 1]
    # no wider box stores sneaked in without the bit (the dwordx4 stores are the node record, which is only read by later kernels)
    assert not [t for t in ins if t.startswith(("global_store_dwordx2", "global_store_dwordx3"))]


def test_the_diagnostic_patch_still_applies():
    """tools/stamps.patch (per-wave time stamps: a diagnostic build kept OUT of the shipping sources) must keep applying to them."""
    import shutil
    import subprocess
    if not os.path.isdir(os.path.join(ROOT, ".git")) or not shutil.which("git"):
        pytest.skip("not a git checkout")
    r = subprocess.run(["git", "-C", ROOT, "apply", "--check", os.path.join("tools", "stamps.patch")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
