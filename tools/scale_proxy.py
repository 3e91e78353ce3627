import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuda_ray_tracer_amd as m
from cuda_ray_tracer_amd import api
w, h, spp = 1920, 1080, 16
stl = m.parseInput("scenes/tenthousand.txt"); raw = m.initRawConfigFromStl(stl, 0); m.build_lbvh_karas(raw)
def run(rows, parts, part):
    p = api.render_params(w, h, spp, rows, parts, part)
    img = torch.empty(api.num_pixels(p) * 4, dtype=torch.uint8, device="cuda")
    best = 1e9
    for i in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.render(img, w, h, spp, raw, params=p); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    return best, raw.stats()["trace_kernel_ms"]
full = run(h, 1, 0)
print("full frame: wall %.2f ms trace %.2f" % full)
for parts in (4, 8):
    for rows in (1, 4):
        ts = [run(rows, parts, k) for k in range(parts)]
        worst = max(t[0] for t in ts)
        print(f"parts={parts} stripe_rows={rows}: slowest part wall {worst:.2f} ms (trace {max(t[1] for t in ts):.2f}) -> ideal-scaling efficiency {full[0]/parts/worst:.3f} (speedup {full[0]/worst:.2f}x)", flush=True)
