"""Small synthetic scenes for the edge cases the bundled files do not reach (test data, generated not stored)."""
import numpy as np

HEADER = "png 64 48 edge.png\n"


def bulbs_and_planes():
    return HEADER + """bounces 3
color 1 0.9 0.8
bulb 0 3 1
color 0.2 0.4 1
bulb -2 1 0.5
color 1 1 1
sun 0.3 1 0.2
color 0.7 0.7 0.7
plane 0 1 0 1
color 0.5 0.6 0.7
shininess 0.3
plane 0 0 1 6
color 1 0.3 0.2
shininess 0.5
roughness 0.1
sphere 0 0 -3 1
color 0.2 1 0.3
shininess 0
sphere 1.5 -0.5 -2.5 0.5
xyz -2 -1 -4
xyz 2 -1 -4
xyz 0 2 -5
color 0.9 0.9 0.2
tri 1 2 3
"""


def fisheye():
    return HEADER + "fisheye\nforward 0 0 -0.6\n" + "color 1 1 1\nsun 1 1 1\nsphere 0 0 -2 0.7\nsphere 1 0.5 -1.5 0.3\nplane 0 1 0 1\n"


def panorama():
    return HEADER + "panorama\neye 0 0.2 0\n" + "color 1 1 1\nsun 1 1 1\ncolor 1 0 0\nsphere 0 0 -2 0.7\ncolor 0 1 0\nsphere 2 0 0 0.7\ncolor 0 0 1\nsphere 0 0 2 0.7\nplane 0 1 0 1\n"


def empty():
    return HEADER + "color 1 1 1\nsun 0 1 0\n"


def plane_only():
    return HEADER + "color 1 1 1\nsun 0 1 0.2\ncolor 0.4 0.5 0.6\nshininess 0.5\nplane 0 1 0 1\n"


def single_sphere():
    return HEADER + "color 1 1 1\nsun 1 1 1\nshininess 0.4\nsphere 0 0 -2 0.8\n"


def single_triangle():
    return HEADER + "color 1 1 1\nsun 0 0 1\nxyz -1 -1 -2\nxyz 1 -1 -2\nxyz 0 1 -2\ntri 1 2 3\n"


def zero_bounces():
    return HEADER + "bounces 0\ncolor 1 1 1\nsun 1 1 1\nsphere 0 0 -2 0.8\n"


def one_bounce_glass_gi():
    return HEADER + """bounces 5
gi 2
expose 1.5
dof 2.0 0.05
color 1 1 1
sun 1 1 0.5
color 0.3 0.3 0.3
plane 0 1 0 1
color 1 1 1
transparency 0.9
shininess 0.1
ior 1.3
sphere 0 0 -2 0.7
transparency 0 0.5 0
shininess 0.6 0.2 0.1
roughness 0.2
color 0.9 0.4 0.1
sphere 1.2 0 -2.5 0.5
transparency 0
shininess 0
color 0.2 0.3 0.9
xyz -3 -1 -5
xyz 3 -1 -5
xyz 0 3 -5
tri -3 -2 -1
"""


def glass_spheres_bulb():
    """Spheres only, but with a point light, glass and gi: quantised nodes with the general (unspecialised) kernel."""
    return HEADER + """bounces 4
gi 1
color 1 1 1
sun 1 1 0.5
color 1 0.8 0.6
bulb 0.5 2 0
color 0.4 0.4 0.4
plane 0 1 0 1
color 1 1 1
transparency 0.8
shininess 0.1
ior 1.4
sphere 0 0 -2 0.7
transparency 0
shininess 0.5
roughness 0.1
color 0.9 0.4 0.1
sphere 1.2 0 -2.5 0.5
color 0.2 0.9 0.3
shininess 0
sphere -1.1 -0.3 -1.8 0.4
"""


def deep_stack(n=3000):
    """n concentric, slightly shifted spheres: every box overlaps every other, so rays push at every level of a tree that
    the duplicate-code tie-break makes deep; exercises stack depths beyond the LDS part of the traversal stack."""
    rng = np.random.default_rng(7)
    lines = [HEADER, "color 1 1 1\nsun 1 1 1\nbounces 2\nshininess 0.2\n"]
    for i in range(n):
        r = 0.5 + 0.4 * rng.random()
        lines.append(f"sphere {1e-5 * rng.standard_normal():.8f} {1e-5 * rng.standard_normal():.8f} {-3 + 1e-5 * rng.standard_normal():.8f} {r:.6f}\n")
    return "".join(lines)


def far_camera(n=400):
    """A unit-sized cluster of reflective spheres over a plane, seen through a long lens from 30 000 units away: every ray
    parameter near the cluster is ~3e4, whose ulp (0.002) is 60 grid steps of the quantised node records (2 / 65535) -- the
    regime in which their box test has to round outwards (shade_common.h, box_pair_q)."""
    rng = np.random.default_rng(11)
    out = [HEADER, "bounces 3\n", "eye 0.1 0.3 30000\n", "forward 0 0 -30000\n", "color 1 1 1\n", "sun 1 1 1\n", "sun -1 2 0.5\n",
           "color 0.6 0.6 0.6\n", "shininess 0.3\n", "plane 0 1 0 1\n", "shininess 0.5\n"]
    for _ in range(n):
        c = rng.uniform(-1, 1, 3)
        col = rng.uniform(0.2, 1, 3)
        out.append("color %.3f %.3f %.3f\n" % tuple(col))
        out.append("sphere %.4f %.4f %.4f %.4f\n" % (c[0], c[1], c[2], rng.uniform(0.03, 0.12)))
    return "".join(out)


def far_from_origin(n=300, offset=1000.0):
    """A unit-sized cluster of spheres 1 000 units from the world origin: ulp(coordinate) = 6e-5 is two steps of the quantised
    grid (2 / 65535), so the outward rounding of the quantised boxes no longer covers the rounding of a sphere's own box planes
    -- the build clears `grid_ok` and the scene is walked over the exact records, in the reference's order (lbvh_build.hip)."""
    rng = np.random.default_rng(12)
    out = [HEADER, "bounces 3\n", "eye %.1f 0.3 4\n" % offset, "forward 0 0 -1\n", "color 1 1 1\n", "sun 1 1 1\n", "sun -1 2 0.5\n",
           "color 0.6 0.6 0.6\n", "shininess 0.3\n", "plane 0 1 0 1\n", "shininess 0.5\n"]
    for _ in range(n):
        c = rng.uniform(-1, 1, 3)
        out.append("color %.3f %.3f %.3f\n" % tuple(rng.uniform(0.2, 1, 3)))
        out.append("sphere %.4f %.4f %.4f %.4f\n" % (offset + c[0], c[1], c[2], rng.uniform(0.03, 0.12)))
    return "".join(out)


def axis_parallel_rays(n=6):
    """An n x n x n lattice of spheres around the camera, which sits inside the scene's bounds and looks straight down -z: at
    spp 0 and an even frame size the central column and row of pixels have direction components that are exactly 0 (reciprocal
    +inf, grid parameters inf / NaN in the quantised box test, shade_common.h quantised_axis), and many rays run exactly along
    lattice planes, i.e. along faces of node boxes."""
    out = [HEADER, "bounces 3\n", "eye 0.5 0.5 0.5\n", "color 1 1 1\n", "sun 1 2 3\n", "sun -2 1 0\n", "shininess 0.4\n"]
    for i in range(-n // 2, n // 2 + 1):
        for j in range(-n // 2, n // 2 + 1):
            for k in range(-n // 2, n // 2 + 1):
                if (i, j, k) == (0, 0, 0):
                    continue
                out.append("color %.2f %.2f %.2f\n" % (0.3 + 0.1 * (i % 5), 0.3 + 0.1 * (j % 5), 0.3 + 0.1 * (k % 5)))
                out.append("sphere %d %d %d 0.25\n" % (i, j, k))
    return "".join(out)


ALL = {"bulbs_and_planes": bulbs_and_planes, "fisheye": fisheye, "panorama": panorama, "empty": empty, "plane_only": plane_only,
       "single_sphere": single_sphere, "single_triangle": single_triangle, "zero_bounces": zero_bounces,
       "glass_gi_dof": one_bounce_glass_gi, "glass_spheres_bulb": glass_spheres_bulb, "axis_parallel_rays": axis_parallel_rays, "deep_stack": deep_stack}
