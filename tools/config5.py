"""BASELINE config 5 on one GPU: 1 M spheres + 1 M triangles, 3840x2160 at 256 spp (2.1 G samples, 34 GB of sample buffer)."""
import sys, time, torch
sys.path.insert(0, ".")
import cuda_ray_tracer_amd as m
from cuda_ray_tracer_amd import api
stl = m.syntheticScene(1_000_000, 1_000_000, seed=1234)
raw = m.initRawConfigFromStl(stl, 0)
print("LBVH build %.3f ms" % m.build_lbvh_karas(raw), flush=True)
w, h, spp = 3840, 2160, 256
img = torch.empty(w * h * 4, dtype=torch.uint8, device="cuda")
for i in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.render(img, w, h, spp, raw); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = raw.stats()
    print(f"frame {i}: {dt*1e3:.0f} ms wall, trace kernel {st['trace_kernel_ms']:.0f} ms", flush=True)
a = img.cpu().numpy().reshape(h, w, 4)
print("alpha>0 pixels:", int((a[..., 3] > 0).sum()), "of", w * h, "mean rgb", a[..., :3].mean(axis=(0, 1)).round(2).tolist())
m.write_png("gpurun_out/config5_4k_256spp.png", a.reshape(-1), w, h)
raw.close()
