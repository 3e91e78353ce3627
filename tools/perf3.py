import sys, os, time, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import cuda_ray_tracer_amd as m
from cuda_ray_tracer_amd import api
scenes = sys.argv[1:] or ["tenthousand"]
for name in scenes:
    w, h, spp = 1920, 1080, int(os.environ.get("PERF_SPP", "16"))
    stl = m.parseInput(f"scenes/{name}.txt")
    raw = m.initRawConfigFromStl(stl, 0)
    m.build_lbvh_karas(raw)
    p = api.render_params(w, h, spp, counters=True)
    n = api.num_pixels(p)
    img = torch.empty(n * 4, dtype=torch.uint8, device="cuda")
    m.render(img, w, h, spp, raw, params=p); torch.cuda.synchronize()
    st = raw.stats()
    p2 = api.render_params(w, h, spp)
    best = (1e9, 0, 0)
    for i in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.render(img, w, h, spp, raw, params=p2); torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        s2 = raw.stats()
        if wall < best[0]: best = (wall, s2["trace_kernel_ms"], s2["render_ms"])
    ab = st["internal_visits"] * 64 + st["sphere_tests"] * 16 + st["tri_tests"] * 48 + st["mat_fetches"] * 44
    tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("MIRT_"))
    print(f"{name} [{tag}] wall {best[0]:.1f} ms  render_ms {best[2]:.1f}  trace-kernels {best[1]:.1f} ms | Mrays/s(wall) {st['rays']/best[0]/1e3:.0f}  frac(trace) {ab/best[1]/1e6/8000:.3f}  sum={int(img.sum().item())}", flush=True)
    raw.close()
