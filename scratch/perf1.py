import sys, time, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import cuda_ray_tracer_amd as m
from cuda_ray_tracer_amd import api
for name, w, h, spp in [("tenthousand", 1920, 1080, 16), ("spiral", 1920, 1080, 16), ("redchair", 1920, 1080, 32)]:
    stl = m.parseInput(f"scenes/{name}.txt")
    raw = m.initRawConfigFromStl(stl, 0)
    ms = m.build_lbvh_karas(raw)
    p = api.render_params(w, h, spp, counters=True)
    n = api.num_pixels(p)
    img = torch.empty(n * 4, dtype=torch.uint8, device="cuda")
    m.render(img, w, h, spp, raw, params=p); torch.cuda.synchronize()
    st = raw.stats()
    p2 = api.render_params(w, h, spp)
    for i in range(3):
        t = time.time(); m.render(img, w, h, spp, raw, params=p2); torch.cuda.synchronize(); dt = time.time() - t
        s2 = raw.stats()
        print(name, "build %.3f ms" % ms, "frame %.1f ms (trace kernel %.1f ms)" % (dt * 1e3, s2["trace_kernel_ms"]), "Mrays/s %.0f" % (st["rays"] / dt / 1e6),
              "rays/sample %.2f I/ray %.1f" % (st["rays"] / st["samples"], st["internal_visits"] / st["rays"]),
              "alg GB/s %.0f" % ((st["internal_visits"] * 64 + st["sphere_tests"] * 16 + st["tri_tests"] * 48 + st["mat_fetches"] * 44) / (s2["trace_kernel_ms"] * 1e-3) / 1e9), flush=True)
    print("counted-variant trace ms", st["trace_kernel_ms"], st)
    raw.close()
