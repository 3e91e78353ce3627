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
import cuda_ray_tracer_amd as m          # noqa: E402
from cuda_ray_tracer_amd import api      # noqa: E402


def scene_text(rng, tri_fraction=0.0):
    scale = 10.0 ** rng.uniform(-3, 6)
    n = int(10 ** rng.uniform(0.3, 3.7))
    lines = ["png 64 64 fuzz.png", "bounces %d" % rng.integers(1, 8)]
    mode = rng.integers(0, 4)
    if mode == 1:
        lines.append("fisheye")
    elif mode == 2:
        lines.append("panorama")
    where = rng.integers(0, 3)          # 0 inside the cloud, 1 outside, 2 far away
    dist = [0.2, 3.0, 10.0 ** rng.uniform(2, 4.5)][where]
    eye = rng.normal(size=3)
    eye = eye / np.linalg.norm(eye) * dist * scale
    if rng.random() < 0.3:              # axis-aligned view: exact zeros in ray directions
        eye = np.array([0.0, 0.0, dist * scale])
    lines.append("eye %.9g %.9g %.9g" % tuple(eye))
    fwd = -eye if np.linalg.norm(eye) > 0 else np.array([0.0, 0.0, -1.0])
    if where == 2:
        fwd = fwd * 30.0               # long lens
    elif where == 0:
        fwd = rng.normal(size=3)
    lines.append("forward %.9g %.9g %.9g" % tuple(fwd))
    if rng.random() < 0.3:
        lines.append("dof %.6g %.6g" % (dist * scale, 0.01 * scale))
    general = rng.random() < 0.25
    if general and rng.random() < 0.5:
        lines.append("gi %d" % rng.integers(1, 3))
    for _ in range(rng.integers(1, 4)):
        lines.append("color %.3f %.3f %.3f" % tuple(rng.uniform(0.3, 1.2, 3)))
        lines.append("sun %.4f %.4f %.4f" % tuple(rng.normal(size=3)))
    if general and rng.random() < 0.6:
        lines.append("color 1 0.9 0.8")
        lines.append("bulb %.6g %.6g %.6g" % tuple(rng.normal(size=3) * 2 * scale))
    if rng.random() < 0.7:
        lines.append("color 0.5 0.5 0.5")
        lines.append("plane 0 1 0 %.6g" % (1.5 * scale))
    for _ in range(n):
        if rng.random() < 0.3:
            lines.append("color %.3f %.3f %.3f" % tuple(rng.uniform(0.1, 1.0, 3)))
        if rng.random() < 0.1:
            lines.append("shininess %.3f" % rng.choice([0.0, 0.3, 0.8]))
        if rng.random() < 0.1:
            lines.append("roughness %.3f" % rng.choice([0.0, 0.05, 0.3]))
        if general and rng.random() < 0.05:
            lines.append("transparency %.2f" % rng.choice([0.0, 0.7]))
        c = rng.normal(size=3) * scale
        r = scale * 10.0 ** rng.uniform(-2.5, 0.3)
        if rng.random() < 0.05:
            r = scale * 3.0             # a big sphere that contains many others
        lines.append("sphere %.9g %.9g %.9g %.9g" % (c[0], c[1], c[2], r))
    if tri_fraction > 0:
        # triangles among the spheres: the tree then has subtrees that must keep the reference's order
        for _ in range(int(n * tri_fraction) + 1):
            if rng.random() < 0.3:
                lines.append("color %.3f %.3f %.3f" % tuple(rng.uniform(0.1, 1.0, 3)))
            c = rng.normal(size=3) * scale
            size = scale * 10.0 ** rng.uniform(-2.0, 0.0)
            for _k in range(3):
                v = c + rng.normal(size=3) * size
                if rng.random() < 0.2:
                    v[rng.integers(0, 3)] = c[0]          # axis-aligned edges and flat boxes now and then
                lines.append("xyz %.9g %.9g %.9g" % tuple(v))
            lines.append("tri -3 -2 -1")
    return "\n".join(lines) + "\n"


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
    ap.add_argument("--reference-walk", action="store_true", help="compare with {traversal 0, shadow_anyhit 0, skip_unlit 0, qnodes 0}: the reference's walk ray for ray")
    args = ap.parse_args(argv)
    rng = np.random.default_rng(args.seed)
    bad = 0
    visits = [0, 0]
    for i in range(args.scenes):
        text = scene_text(rng, args.triangles)
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
