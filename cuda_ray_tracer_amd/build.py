"""Builds libmirt.so (host C++ + HIP kernels for gfx950) and the `raytracer` CLI, in-tree, with hipcc.

    python -m cuda_ray_tracer_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU.  Outputs go to cuda_ray_tracer_amd/_build/ (git-ignored, but
shipped to the GPU box by gpurun).  -ffp-contract=off: the kernels and the host parser compute exactly what
the reference source says, one rounding per operation ("Arithmetic").
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libmirt.so")
CLI = os.path.join(OUT, "raytracer")
ARCH = "gfx950"

LIB_SOURCES = ["host_scene.cpp", "png_writer.cpp", "xorwow_tables.cpp", "multi.cpp", "lbvh_build.hip", "render.hip", "wavefront.hip", "api.hip"]
COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
          f"--offload-arch={ARCH}", "-fno-gpu-flush-denormals-to-zero"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _all_deps():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(HERE, "..", "include", "mirt.h"))
    deps.append(os.path.abspath(__file__))
    return deps


def source_hash():
    """sha256[:16] over csrc/: names the build a counter summary (tools/pmc_profile.py) was taken on; bench.py compares."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        h.update(f.encode())
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def build(force=False, verbose=False):
    os.makedirs(OUT, exist_ok=True)
    hipcc = _hipcc()
    deps = _all_deps()
    extra = []
    for var in ("MIRT_WAVES_PER_SIMD", "MIRT_STACK_LDS", "MIRT_WF_WAVES_PER_SIMD", "MIRT_WF_SHADE_WAVES", "MIRT_TRACE_BLOCK"):      # tuning experiments only; defaults live in render.hip
        if os.environ.get(var):
            extra.append(f"-D{var}=" + os.environ[var])
    objs = []
    for src in LIB_SOURCES:
        obj = os.path.join(OUT, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, deps):
            cmd = [hipcc] + COMMON + extra + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs + ["-lz", "-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    cli_src = os.path.join(CSRC, "raytracer_main.cpp")
    if os.path.exists(cli_src) and (force or _stale(CLI, [cli_src, LIB])):
        cmd = [hipcc, "-O2", "-std=c++17", "-x", "hip", f"--offload-arch={ARCH}", cli_src, "-o", CLI, "-L" + OUT, "-lmirt", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
