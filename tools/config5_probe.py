"""BASELINE config 5's scene (1 M spheres + 1 M triangles, 3840x2160) at a reduced sample count -- one slab, i.e. one trace
launch per frame -- for tools/pmc_profile.py --program: the per-launch counters of the kernel that renders config 5.

    python3 tools/config5_probe.py [spp=8] [frames=3]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import cuda_ray_tracer_amd as m      # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3
stl = m.syntheticScene(1_000_000, 1_000_000, seed=1234)
raw = m.initRawConfigFromStl(stl, 0)
m.build_lbvh_karas(raw)
w, h = 3840, 2160
img = torch.empty(w * h * 4, dtype=torch.uint8, device="cuda")
ms = []
for i in range(frames):
    m.render(img, w, h, spp, raw)
    torch.cuda.synchronize()
    ms.append(raw.stats()["trace_kernel_ms"])
print(json.dumps({"workload": f"synthetic 1M+1M 3840x2160 {spp}spp", "roofline": {"kernel_ms": sum(ms[1:]) / max(len(ms) - 1, 1)}}), flush=True)
raw.close()
