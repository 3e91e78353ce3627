import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuda_ray_tracer_amd as m
stl = m.syntheticScene(1_000_000, 1_000_000, seed=1234)
raw = m.initRawConfigFromStl(stl, 0)
for i in range(3):
    print("build ms", m.build_lbvh_karas(raw), flush=True)
raw.close()
