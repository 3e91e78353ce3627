import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuda_ray_tracer_amd as m
from cuda_ray_tracer_amd import api
w, h, spp = 1920, 1080, 16
stl = m.parseInput("scenes/tenthousand.txt"); raw = m.initRawConfigFromStl(stl, 0); m.build_lbvh_karas(raw)
for parts in (1, 8, 27, 135, 540):
    p = api.render_params(w, h, spp, 2, parts, 0)
    img = torch.empty(api.num_pixels(p) * 4, dtype=torch.uint8, device="cuda")
    best = 1e9
    for i in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.render(img, w, h, spp, raw, params=p); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    print(f"parts={parts}: pixels {api.num_pixels(p)} wall {best:.2f} ms trace {raw.stats()['trace_kernel_ms']:.2f}", flush=True)
