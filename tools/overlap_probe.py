import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuda_ray_tracer_amd as m
from cuda_ray_tracer_amd import api
w, h, spp = 1920, 1080, 16
stl = m.parseInput("scenes/tenthousand.txt"); raw = m.initRawConfigFromStl(stl, 0); m.build_lbvh_karas(raw)
for parts in (8,):
    p = api.render_params(w, h, spp, 4 if parts > 1 else h, parts, 0)
    n = api.num_pixels(p)
    for nfl in (2, 3, 4):
        streams = [torch.cuda.Stream() for _ in range(nfl)]
        bufs = [torch.empty(n * 4, dtype=torch.uint8, device="cuda") for _ in range(nfl)]
        K = 12
        for rep in range(2):
            raw.stats(); torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(K):
                with torch.cuda.stream(streams[i % nfl]):
                    m.render(bufs[i % nfl], w, h, spp, raw, params=p)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K * 1e3
        st = raw.stats()
        print(f"parts={parts} frames_in_flight={nfl}: {dt:.2f} ms/frame  (mean trace kernel {st['trace_kernel_ms_mean']:.2f} ms over {st['frames_timed']} frames)", flush=True)
