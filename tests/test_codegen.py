"""The trace kernel runs at the 128-VGPR edge of 4 waves per SIMD, and what the compiler decides to spill inside the
traversal loop has moved its speed by 10-50 % more than once (a scratch reload of the lane's LDS stack address on every
push and pop).  This test compiles render.hip to assembly (no GPU needed) and keeps register spills to scratch out of
the loop, apart from the pointer reloads on the global stack-spill path that the bundled scenes never take and at most
one other reload."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_no_scratch_reloads_on_the_traversal_loop_hot_path():
    import hotloop
    res, cnt, spills = hotloop.analyze()
    occ = [r for r in res if r.startswith("Occupancy")]
    assert occ and occ[0].rsplit(":", 1)[1].strip() == "4", res      # 4 waves per SIMD
    scratch = [t for _, t in spills if t.startswith("scratch_")]
    assert not [t for t in scratch if t.startswith("scratch_store")], scratch
    # the 64-bit spill-area pointer on the global stack-spill path (never taken by the bundled scenes), once per step-loop copy
    assert len([t for t in scratch if t.startswith("scratch_load_dwordx2")]) <= 4, (cnt, scratch)
    # at most one single-register reload (today: an LDS address of the shade phase restored after the triangle test used
    # its register); the expensive cases were two of them, on the push and on the pop
    assert len([t for t in scratch if not t.startswith("scratch_load_dwordx2")]) <= 1, (cnt, scratch)
