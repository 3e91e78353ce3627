"""Which pixels differ between two option sets?  usage: diffmodes.py scene W H SPP "k=v,k=v" "k=v,..." """
import sys, os, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import cuda_ray_tracer_amd as m
from cuda_ray_tracer_amd import api
name, w, h, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
stl = m.syntheticScene(1_000_000, 1_000_000, seed=1234) if name == "synthetic" else m.parseInput(f"scenes/{name}.txt")
out = []
for optstr in sys.argv[5:7]:
    raw = m.initRawConfigFromStl(stl, 0)
    for kv in optstr.split(","):
        if kv:
            k, v = kv.split("="); raw.set_option(k, int(v))
    m.build_lbvh_karas(raw)
    img = torch.empty(w * h * 4, dtype=torch.uint8, device="cuda"); flt = torch.empty(w * h * 4, dtype=torch.float32, device="cuda")
    m.render(img, w, h, spp, raw, d_float=flt); torch.cuda.synchronize()
    out.append((img.cpu().numpy().reshape(h, w, 4), flt.cpu().numpy().reshape(h, w, 4)))
    raw.close()
d = np.argwhere(np.any(out[0][1].view(np.uint32) != out[1][1].view(np.uint32), axis=-1))
print(name, sys.argv[5], "vs", sys.argv[6], ":", len(d), "pixels differ")
for y, x in d[:20]:
    print("  (x,y)=", (int(x), int(y)), out[0][1][y, x].tolist(), out[1][1][y, x].tolist(), out[0][0][y, x].tolist(), out[1][0][y, x].tolist())
