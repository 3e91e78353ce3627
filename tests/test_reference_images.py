"""Pins the oracle (CPU test) and the HIP path (GPU test) to the only outputs the reference holds: its three 800x800
renders, docs/{tenthousand,spiral,redchair}.png (tests/golden/ref_docs/, data).

What this pins, statistically: geometry, camera, depth of field, rough reflections, shadows, GI, refraction, the sRGB
transfer and the sample mean -- a render that drops or alters any of them fails (negative controls below).
What it cannot pin: cuRAND bit-compatibility, CUDA libm rounding, `pow(x,2)` promotion, argument evaluation order.

docs/redchair.png was rendered WITHOUT exposure although redchair.txt says `expose 2` and the source applies it
(helper.cu:40-45 via config_utils.cu:31): with the line the render is 24 % brighter than the shipped image, without
it the means agree to 0.5/255.  The source is the specification (the product keeps exposure); the image is compared
with the `expose` line stripped.  The shipped images were also quantised by truncation (their means sit 0.5 x coverage
below a rounded render, and within 0.05/255 of a truncated one), as render_kernel does (draw.cu:129-132); the current
render_kernel_warp_aa rounds (draw.cu:9-11,202-205) and so does the product.
"""
import numpy as np
import pytest

import oracle_lib as ol
import pyscene
import refimg
from conftest import scene_path

SCENES = ["tenthousand", "spiral", "redchair"]


def oracle_render(name, size, spp, **override):
    sc = pyscene.parse_file(scene_path(name))
    if name == "redchair" and "expose" not in override:
        sc.expose = np.float32(np.inf)
    for k, v in override.items():
        setattr(sc, k, v)
    o = ol.OracleScene(sc, bounds_mode=0)
    r = o.render(size, size, spp, nthreads=8)
    o.close()
    return r["u8"]


@pytest.mark.parametrize("name", SCENES)
def test_oracle_agrees_with_the_reference_held_render(name):
    """Oracle at 400x400, 8 spp, against the 2x2 box-filtered reference image."""
    c = refimg.compare(name, oracle_render(name, 400, 8))
    assert np.all(np.abs(c["dmean"]) <= refimg.MEAN_TOL), c
    lo_full, lo_100 = refimg.PSNR_FLOOR[name]
    assert c["psnr_full"] >= lo_full and c["psnr_100"] >= lo_100, c


@pytest.mark.parametrize("override", [dict(expose=np.float32(2.0)), dict(gi=0), dict(bounces=1)])
def test_the_comparison_rejects_a_structurally_wrong_render(override):
    """Negative controls on redchair.txt (GI, a glass sphere, reflections): exposure as in the scene file, no GI, one
    bounce -- each must fail the criterion the faithful render passes."""
    c = refimg.compare("redchair", oracle_render("redchair", 400, 8, **override))
    assert not refimg.passes("redchair", c), c


@pytest.mark.gpu
@pytest.mark.parametrize("name", SCENES)
def test_hip_path_agrees_with_the_reference_held_render(name):
    """The product at the reference's native size with each scene's own `aa` (40 / 32 / 32: this is also the `aa`
    generalisation at the sizes the reference was run at), through the C ABI, against the shipped image."""
    import torch
    import cuda_ray_tracer_amd as m
    text = open(scene_path(name)).read()
    if name == "redchair":
        text = refimg.strip_expose(text)
    stl = m.parseText(text)
    assert (stl.width, stl.height) == (800, 800) and stl.aa > 1
    raw = m.initRawConfigFromStl(stl, 0)
    m.build_lbvh_karas(raw)
    img = torch.empty(800 * 800 * 4, dtype=torch.uint8, device="cuda")
    flt = torch.empty(800 * 800 * 4, dtype=torch.float32, device="cuda")
    m.render(img, 800, 800, stl.aa, raw, d_float=flt)
    torch.cuda.synchronize()
    u8 = img.cpu().numpy().reshape(800, 800, 4)
    f32 = flt.cpu().numpy().reshape(800, 800, 4)
    raw.close()
    c = refimg.compare(name, u8)
    assert np.all(np.abs(c["dmean"]) <= refimg.MEAN_TOL), c
    lo_full, lo_100 = refimg.PSNR_FLOOR_NATIVE[name]
    assert c["psnr_full"] >= lo_full and c["psnr_100"] >= lo_100, c
    # the shipped images were quantised by truncation: with the same quantiser the means agree to a few hundredths of a level
    ct = refimg.compare(name, refimg.srgb_truncated(f32))
    assert np.all(np.abs(ct["dmean"]) <= refimg.MEAN_TOL_TRUNCATED_NATIVE), ct
