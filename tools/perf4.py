#!/usr/bin/env python3
"""Wall / trace-kernel time and visit counters of whole frames, per scene option set, on one box.

    python3 tools/perf4.py SPEC [SPEC ...]      SPEC = scene:W:H:SPP[:opt=v,opt=v...]      scene = a bundled name or `synth`

Prints one line per SPEC: best-of-3 serial frame time, trace-kernel time, rays, node visits and leaf tests per ray and a byte sum
of the image (equal sums between option sets that must not change pixels).  Honours MIRT_LIB (tools/ab.py variants)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import cuda_ray_tracer_amd as m      # noqa: E402
from cuda_ray_tracer_amd import api   # noqa: E402

synth = None
for spec in sys.argv[1:]:
    f = spec.split(":")
    name, w, h, spp = f[0], int(f[1]), int(f[2]), int(f[3])
    opts = dict(kv.split("=") for kv in f[4].split(",")) if len(f) > 4 and f[4] else {}
    if name == "synth":
        synth = synth or m.syntheticScene(1_000_000, 1_000_000, seed=1234)
        stl = synth
    else:
        stl = m.parseInput(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scenes", name + ".txt"))
    raw = m.initRawConfigFromStl(stl, 0)
    for k, v in opts.items():
        raw.set_option(k, int(v))
    m.build_lbvh_karas(raw)
    frames = int(os.environ.get("PERF_FRAMES", "3"))
    img = torch.empty(w * h * 4, dtype=torch.uint8, device="cuda")
    st = None
    if os.environ.get("PERF_COUNT", "1") != "0":
        m.render(img, w, h, spp, raw, params=api.render_params(w, h, spp, counters=True)); torch.cuda.synchronize()
        st = raw.stats()
    p2 = api.render_params(w, h, spp)
    best = (1e9, 0.0)
    for i in range(frames):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.render(img, w, h, spp, raw, params=p2); torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        s2 = raw.stats()
        if wall < best[0]:
            best = (wall, s2["trace_kernel_ms"])
    extra = ""
    if st:
        extra = f" | rays {st['rays'] / 1e6:.1f} M, I/ray {st['internal_visits'] / st['rays']:.2f}, leaf/ray {(st['sphere_tests'] + st['tri_tests']) / st['rays']:.2f}, Mrays/s {st['rays'] / best[0] / 1e3:.0f}"
    print(f"{spec} [{os.path.basename(os.path.dirname(api.LIB_PATH))}] wall {best[0]:.2f} ms  trace {best[1]:.2f} ms{extra}  sum={int(img.sum(dtype=torch.int64).item())}", flush=True)
    raw.close()
