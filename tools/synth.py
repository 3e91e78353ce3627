import sys, time, torch
sys.path.insert(0, ".")
import cuda_ray_tracer_amd as m
from cuda_ray_tracer_amd import api
t = time.time(); stl = m.syntheticScene(1_000_000, 1_000_000, seed=1234); print("generate %.2f s" % (time.time() - t), flush=True)
t = time.time(); raw = m.initRawConfigFromStl(stl, 0); print("upload %.2f s" % (time.time() - t), flush=True)
for i in range(3):
    ms = m.build_lbvh_karas(raw); print("LBVH build (N=2,000,000): %.3f ms" % ms, flush=True)
for (w, h, spp) in [(1920, 1080, 4), (3840, 2160, 4)]:
    p = api.render_params(w, h, spp, counters=True)
    img = torch.empty(api.num_pixels(p) * 4, dtype=torch.uint8, device="cuda")
    m.render(img, w, h, spp, raw, params=p); torch.cuda.synchronize(); st = raw.stats()
    p2 = api.render_params(w, h, spp)
    m.render(img, w, h, spp, raw, params=p2); torch.cuda.synchronize(); s2 = raw.stats()
    ab = st["internal_visits"] * 64 + st["sphere_tests"] * 16 + st["tri_tests"] * 48 + st["mat_fetches"] * 44
    print(f"synthetic 1M+1M {w}x{h} {spp}spp: trace {s2['trace_kernel_ms']:.1f} ms, rays {st['rays']/1e6:.1f} M, {st['rays']/s2['trace_kernel_ms']/1e3:.0f} Mrays/s, I/ray {st['internal_visits']/st['rays']:.1f}, leaf/ray {(st['sphere_tests']+st['tri_tests'])/st['rays']:.2f}, max_stack {st['max_stack']}, alg GB/s {ab/s2['trace_kernel_ms']/1e6:.0f} (frac {ab/s2['trace_kernel_ms']/1e6/8000:.3f})", flush=True)
m.write_png("gpurun_out/synthetic_1080p.png", img[:0].cpu().numpy() if False else torch.zeros(4, dtype=torch.uint8).numpy(), 1, 1)
raw.close()
