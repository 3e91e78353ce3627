"""Random scene texts for the mode fuzzer (tools/fuzz_modes.py) and for tests/test_oracle_units.py: numpy only."""
import numpy as np


def scene_text(rng, tri_fraction=0.0, far=False, offset=False, lights=False, dups=False):
    """far: the regime of tests/golden/far_camera_tie.txt -- the camera 10^3 .. 10^5 scene sizes away (ulp(t) of the size of the
    spheres' features), many large overlapping spheres: sphere hits that round below the entry of their own boxes are common."""
    scale = 10.0 ** rng.uniform(-3, 6)
    # offset: the whole scene (camera, primitives, point lights) moved away from the world origin by 3 .. 3000 scene sizes --
    # coordinates whose ulp approaches the quantised grid's step (grid_ok, lbvh_build.hip) and the spheres' features
    off = np.zeros(3)
    if offset:
        v = rng.normal(size=3)
        off = v / np.linalg.norm(v) * scale * 10.0 ** rng.uniform(0.5, 3.5)
    n = int(10 ** rng.uniform(0.3, 2.7 if far else 3.7))
    lines = ["png 64 64 fuzz.png", "bounces %d" % rng.integers(1, 8)]
    mode = rng.integers(0, 4)
    if mode == 1:
        lines.append("fisheye")
    elif mode == 2:
        lines.append("panorama")
    where = 2 if far else rng.integers(0, 3)          # 0 inside the cloud, 1 outside, 2 far away
    dist = [0.2, 3.0, 10.0 ** (rng.uniform(3, 5) if far else rng.uniform(2, 4.5))][where]
    eye = rng.normal(size=3)
    eye = eye / np.linalg.norm(eye) * dist * scale
    if rng.random() < 0.3:              # axis-aligned view: exact zeros in ray directions
        eye = np.array([0.0, 0.0, dist * scale])
    lines.append("eye %.9g %.9g %.9g" % tuple(eye + off))
    fwd = -eye if np.linalg.norm(eye) > 0 else np.array([0.0, 0.0, -1.0])
    if where == 2:
        fwd = fwd * 30.0               # long lens
    elif where == 0:
        fwd = rng.normal(size=3)
    lines.append("forward %.9g %.9g %.9g" % tuple(fwd))
    if rng.random() < 0.3:
        lines.append("dof %.6g %.6g" % (dist * scale, 0.01 * scale))
    general = lights or rng.random() < 0.25      # (lights: every scene has a point light, an infinite plane, glass / gi now and then)
    if general and rng.random() < 0.5:
        lines.append("gi %d" % rng.integers(1, 3))
    for _ in range(rng.integers(1, 4)):
        lines.append("color %.3f %.3f %.3f" % tuple(rng.uniform(0.3, 1.2, 3)))
        lines.append("sun %.4f %.4f %.4f" % tuple(rng.normal(size=3)))
    if general and (lights or rng.random() < 0.6):
        lines.append("color 1 0.9 0.8")
        lines.append("bulb %.9g %.9g %.9g" % tuple(rng.normal(size=3) * 2 * scale + off))
    if lights or rng.random() < 0.7:
        lines.append("color 0.5 0.5 0.5")
        lines.append("plane 0 1 0 %.6g" % (1.5 * scale))
    placed = []
    for _ in range(n):
        if rng.random() < 0.3:
            lines.append("color %.3f %.3f %.3f" % tuple(rng.uniform(0.1, 1.0, 3)))
        if rng.random() < 0.1:
            lines.append("shininess %.3f" % rng.choice([0.0, 0.3, 0.8]))
        if rng.random() < 0.1:
            lines.append("roughness %.3f" % rng.choice([0.0, 0.05, 0.3]))
        if general and rng.random() < 0.05:
            lines.append("transparency %.2f" % rng.choice([0.0, 0.7]))
        c = rng.normal(size=3) * scale + off
        r = scale * 10.0 ** rng.uniform(-2.5, 0.3)
        if rng.random() < (0.3 if far else 0.05):
            r = scale * 3.0             # a big sphere that contains many others
        if dups and placed and rng.random() < 0.3:      # an exact copy of an earlier sphere (another material): every hit a tie
            c, r = placed[rng.integers(0, len(placed))]
        placed.append((c, r))
        lines.append("sphere %.9g %.9g %.9g %.9g" % (c[0], c[1], c[2], r))
    if tri_fraction > 0:
        # triangles among the spheres: the tree then has subtrees that must keep the reference's order
        for _ in range(int(n * tri_fraction) + 1):
            if rng.random() < 0.3:
                lines.append("color %.3f %.3f %.3f" % tuple(rng.uniform(0.1, 1.0, 3)))
            c = rng.normal(size=3) * scale + off
            size = scale * 10.0 ** rng.uniform(-2.0, 0.0)
            for _k in range(3):
                v = c + rng.normal(size=3) * size
                if rng.random() < 0.2:
                    v[rng.integers(0, 3)] = c[0]          # axis-aligned edges and flat boxes now and then
                lines.append("xyz %.9g %.9g %.9g" % tuple(v))
            lines.append("tri -3 -2 -1")
    return "\n".join(lines) + "\n"
