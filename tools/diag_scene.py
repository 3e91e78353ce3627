#!/usr/bin/env python3
"""One scene text file through libmirt (default options and the reference-walk mode) and through the oracle (the mirror of each):
which of the four frames differ, where, and every counter.    python3 tools/diag_scene.py FILE SPP [qnodes] [W H]"""
import os
import sys

import numpy as np
import torch

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import cuda_ray_tracer_amd as m          # noqa: E402
from cuda_ray_tracer_amd import api      # noqa: E402
import oracle_lib as ol                  # noqa: E402
import pyscene                           # noqa: E402

path, spp = sys.argv[1], int(sys.argv[2])
qn = int(sys.argv[3]) if len(sys.argv) > 3 else 1
w, h = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (192, 108)
text = open(path).read()
stl = m.parseText(text)
raw = m.initRawConfigFromStl(stl, 0)
m.build_lbvh_karas(raw)
raw.set_option("qnodes", qn)


def gpu():
    p = api.render_params(w, h, spp, counters=True)
    img = torch.empty(w * h * 4, dtype=torch.uint8, device="cuda"); flt = torch.empty(w * h * 4, dtype=torch.float32, device="cuda")
    m.render(img, w, h, spp, raw, d_float=flt, params=p); torch.cuda.synchronize()
    return flt.cpu().numpy().reshape(h, w, 4), raw.stats()


gd, sd = gpu()
for k in ("traversal", "shadow_anyhit", "skip_unlit", "qnodes"):
    raw.set_option(k, 0)
gr, sr = gpu()
raw.close()
o = ol.OracleScene(pyscene.parse_lines(text.split("\n")), bounds_mode=0)
nfo = o.grid_ok()
fl = ol.product_flags(stl.num_triangles > 0, qnodes=qn, nprims=stl.num_prims, grid_ok=nfo)
od = o.render(w, h, spp, flags=fl, nthreads=16); orf = o.render(w, h, spp, flags=0, nthreads=16)
K = ("rays", "shadow_rays", "internal_visits", "sphere_tests", "tri_tests", "mat_fetches")


def diff(a, b):
    d = (a.view(np.uint32) != b.view(np.uint32)) & ~(np.isnan(a) & np.isnan(b))
    return np.argwhere(d.any(axis=-1))


print("grid_ok", nfo, "mirror flags", fl, "prims", stl.num_prims, "retraces (oracle mirror)", od["stats"]["qn_retraces"])
for name, a, b in (("gpu default vs gpu reference-walk", gd, gr), ("gpu default vs oracle mirror", gd, od["f32"]), ("gpu reference-walk vs oracle plain", gr, orf["f32"]),
                   ("oracle mirror vs oracle plain", od["f32"], orf["f32"])):
    dd = diff(a, b)
    print(f"{name}: {len(dd)} pixels", [(int(x), int(y), a[y, x].tolist(), b[y, x].tolist()) for y, x in dd[:3]])
print("gpu default  ", {k: sd[k] for k in K}); print("oracle mirror", {k: od["stats"][k] for k in K})
print("gpu ref-walk ", {k: sr[k] for k in K}); print("oracle plain ", {k: orf["stats"][k] for k in K})
