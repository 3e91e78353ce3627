#!/usr/bin/env python3
"""Random sphere scenes through libmirt in its default mode (near child first, quantised node records with the permute box
test, kernels specialised by scene) and in the reference's mode (`traversal = 0`: left-first walk over the 64-byte float
records): the frames must be identical byte for byte, float image included, with the same number of rays.  Needs a GPU.

    python tools/fuzz_modes.py [--scenes 200] [--seed 1]

`--triangles F` adds F triangles per sphere (mixed scenes: only sphere-only subtrees may be reordered).  Scales from 1e-3 to 1e6, cameras inside, outside and far away from the scene, pinhole / fisheye / panorama, overlapping and
nested spheres, point lights, glass and gi in a part of the scenes (those run the general kernels).  Prints one line per
mismatch and a summary; exit status 1 on any mismatch."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cuda_ray_tracer_amd as m          # noqa: E402
from cuda_ray_tracer_amd import api      # noqa: E402
from fuzz_scenes import scene_text       # noqa: E402


def render(raw, w, h, spp):
    p = api.render_params(w, h, spp, counters=True)
    img = torch.empty(w * h * 4, dtype=torch.uint8, device="cuda")
    flt = torch.empty(w * h * 4, dtype=torch.float32, device="cuda")
    m.render(img, w, h, spp, raw, d_float=flt, params=p)
    torch.cuda.synchronize()
    return img.cpu().numpy(), flt.cpu().numpy().view(np.uint32), raw.stats()


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default="gpurun_out", help="where the scene text of a mismatch is written")
    ap.add_argument("--triangles", type=float, default=0.0, help="triangles per sphere added to every scene (mixed scenes: float node records)")
    ap.add_argument("--qnodes", type=int, default=1, help="scene option qnodes of the first render (2: quantised records on every scene, i.e. the wide walk on the mixed ones)")
    ap.add_argument("--far", action="store_true", help="every camera 10^3 .. 10^5 scene sizes away, many large overlapping spheres (scene_text)")
    ap.add_argument("--offset", action="store_true", help="every scene moved 3 .. 3000 scene sizes away from the world origin (scene_text)")
    ap.add_argument("--lights", action="store_true", help="every scene has a point light and an infinite plane (and the general kernels: glass / gi now and then)")
    ap.add_argument("--dups", action="store_true", help="a third of the spheres are exact copies of earlier ones: ties in the hit distance everywhere")
    ap.add_argument("--reference-walk", action="store_true", help="compare with {traversal 0, shadow_anyhit 0, skip_unlit 0, qnodes 0}: the reference's walk ray for ray")
    args = ap.parse_args(argv)
    rng = np.random.default_rng(args.seed)
    bad = 0
    visits = [0, 0]
    for i in range(args.scenes):
        text = scene_text(rng, args.triangles, args.far, args.offset, args.lights, args.dups)
        w, h, spp = 192, 108, int(rng.choice([0, 1, 2, 4]))
        stl = m.parseText(text)
        raw = m.initRawConfigFromStl(stl, 0)
        m.build_lbvh_karas(raw)
        raw.set_option("qnodes", args.qnodes)
        a8, af, sa = render(raw, w, h, spp)
        a8b, afb, _ = render(raw, w, h, spp)          # the same options again: now in the measured hand-out order
        raw.set_option("traversal", 0)
        if args.reference_walk:
            for k in ("shadow_anyhit", "skip_unlit", "qnodes"):
                raw.set_option(k, 0)
        b8, bf, sb = render(raw, w, h, spp)
        raw.close()
        visits[0] += sa["internal_visits"]; visits[1] += sb["internal_visits"]
        ok = np.array_equal(a8, b8) and np.array_equal(af, bf) and sa["rays"] == sb["rays"] and np.array_equal(a8, a8b) and np.array_equal(af, afb)
        if not ok:
            bad += 1
            nd = int((a8.reshape(-1, 4) != b8.reshape(-1, 4)).any(axis=1).sum())
            print(f"MISMATCH scene {i} (seed {args.seed}): {nd} pixels differ, rays {sa['rays']} vs {sb['rays']}, spp {spp}, {stl.num_prims} primitives", flush=True)
            os.makedirs(args.out, exist_ok=True)
            with open(os.path.join(args.out, f"fuzz_mismatch_{args.seed}_{i}.txt"), "w") as f:
                f.write(text)
        if i % 25 == 24:
            print(f"{i + 1} scenes, {bad} mismatches", flush=True)
    print(f"done: {args.scenes} scenes, {bad} mismatches; node visits default / reference order: {visits[0]} / {visits[1]}", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
