#!/usr/bin/env python3
"""bench.py --share-of N [--serial] as a program without dashes in its arguments (for tools/pmc_profile.py --program):
    python3 tools/share_probe.py N [serial|fif2] [steps]"""
import os
import runpy
import sys

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
n = sys.argv[1] if len(sys.argv) > 1 else "8"
mode = sys.argv[2] if len(sys.argv) > 2 else "serial"
steps = sys.argv[3] if len(sys.argv) > 3 else "6"
sys.argv = ["bench.py", "--share-of", n, "--cpu-step", "0", "--steps", steps, "--warmup", "2"] + (["--serial"] if mode == "serial" else ["--frames-in-flight", "2"])
runpy.run_path(os.path.join(root, "bench.py"), run_name="__main__")
