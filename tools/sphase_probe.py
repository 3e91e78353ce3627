"""How much of the shade phase is the rough-normal sampling (Box-Muller in double)?  Renders tenthousand.txt as is and
with roughness / depth of field removed, under MIRT_PROF=1 (prints the stamp summary of the trace kernel)."""
import os, sys, re, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuda_ray_tracer_amd as m
from cuda_ray_tracer_amd import api
text = open("scenes/tenthousand.txt").read()
variants = {"as is": text,
            "roughness 0": re.sub(r"^roughness .*$", "roughness 0", text, flags=re.M),
            "roughness 0, no dof": re.sub(r"^dof .*$", "", re.sub(r"^roughness .*$", "roughness 0", text, flags=re.M), flags=re.M)}
w, h, spp = 1920, 1080, 16
for name, t in variants.items():
    stl = m.parseText(t); raw = m.initRawConfigFromStl(stl, 0); m.build_lbvh_karas(raw)
    img = torch.empty(w * h * 4, dtype=torch.uint8, device="cuda")
    print("==", name, flush=True)
    for i in range(2):
        m.render(img, w, h, spp, raw); torch.cuda.synchronize()
    raw.close()
