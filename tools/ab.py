#!/usr/bin/env python3
"""Build A/B variants of libmirt.so for same-box comparisons (different gpurun boxes differ by a few per cent).

    python tools/ab.py NAME [-DMACRO=V ...]      -> cuda_ray_tracer_amd/_build/ab/NAME/libmirt.so
    MIRT_LIB=cuda_ray_tracer_amd/_build/ab/NAME/libmirt.so python bench.py --serial --cpu-step 0

Only the HIP sources are recompiled with the extra defines; host objects come from the default build.
"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cuda_ray_tracer_amd import build as B

name, defs = sys.argv[1], sys.argv[2:]
B.build()
out = os.path.join(B.OUT, "ab", name)
os.makedirs(out, exist_ok=True)
objs = []
for src in B.LIB_SOURCES:
    base = os.path.splitext(src)[0] + ".o"
    if (src.endswith(".hip") and src != "api.hip") or src == "xorwow_tables.cpp":
        obj = os.path.join(out, base)
        subprocess.check_call([B._hipcc()] + B.COMMON + defs + ["-x", "hip", "-c", os.path.join(B.CSRC, src), "-o", obj])
    else:
        obj = os.path.join(B.OUT, base)
    objs.append(obj)
lib = os.path.join(out, "libmirt.so")
subprocess.check_call([B._hipcc(), "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-o", lib] + objs + ["-lz", "-ldl"])
print(lib)
