"""GPU parity on synthetic edge-case scenes: bulbs, several planes, fisheye / panorama cameras, empty scene, N = 1,
bounces 0, glass + GI + exposure + depth of field, and a tree deep enough to use the stack spill path."""
import numpy as np
import pytest
import torch

import cuda_ray_tracer_amd as m
from cuda_ray_tracer_amd import api
import oracle_lib as ol
import pyscene
import edge_scenes

pytestmark = pytest.mark.gpu


def run_case(text, w, h, spp, **options):
    stl = m.parseText(text)
    raw = m.initRawConfigFromStl(stl, 0)
    for k, v in options.items():
        raw.set_option(k, v)
    m.build_lbvh_karas(raw)
    p = api.render_params(w, h, spp, counters=True)
    img = torch.empty(w * h * 4, dtype=torch.uint8, device="cuda")
    flt = torch.empty(w * h * 4, dtype=torch.float32, device="cuda")
    m.render(img, w, h, spp, raw, d_float=flt, params=p)
    torch.cuda.synchronize()
    st = raw.stats()
    tree = raw.tree() if stl.num_prims > 0 else None
    raw.close()
    o = ol.OracleScene(pyscene.parse_lines(text.split("\n")), bounds_mode=0)
    ref = o.render(w, h, spp, flags=ol.product_flags(stl.num_triangles > 0, traversal=options.get("traversal", 1), qnodes=options.get("qnodes", 1),
                                                      nprims=stl.num_prims, grid_ok=o.grid_ok()), nthreads=8)
    if tree is not None:
        on = o.nodes()
        for f in ("left", "right"):
            assert np.array_equal(tree[0][f], on[f]), f
    o.close()
    gf, gu = flt.cpu().numpy().reshape(h, w, 4), img.cpu().numpy().reshape(h, w, 4)
    both_nan = np.isnan(gf) & np.isnan(ref["f32"])
    d = np.where(both_nan, 0.0, np.abs(gf.astype(np.float64) - ref["f32"].astype(np.float64)))
    assert np.array_equal(np.isnan(gf), np.isnan(ref["f32"]))
    assert np.nanmax(d) <= 1e-4, float(np.nanmax(d))
    assert np.abs(gu.astype(np.int32) - ref["u8"].astype(np.int32)).max() <= 1
    for k in ("samples", "rays", "shadow_rays", "internal_visits", "sphere_tests", "tri_tests", "mat_fetches", "max_stack"):
        assert st[k] == ref["stats"][k], (k, st[k], ref["stats"][k])
    return st, gu


@pytest.mark.parametrize("name", [n for n in edge_scenes.ALL if n != "deep_stack"])
@pytest.mark.parametrize("spp,qnodes", [(0, 1), (1, 1), (8, 1), (8, 2)])
def test_edge_scene_matches_oracle(name, spp, qnodes):
    """(qnodes = 2: quantised records on every scene -- the wide walk on the scenes with triangles, which are far too small to get
    it by default.)"""
    st, img = run_case(edge_scenes.ALL[name](), 64, 48, spp, qnodes=qnodes)
    if name == "empty":
        assert img.max() == 0
    if name in ("plane_only", "single_sphere", "bulbs_and_planes"):
        assert img[..., 3].max() == 255


def test_stack_spill_path():
    """Stack depths beyond 24 LDS entries + the top-of-stack register need > 2^25 overlapping primitives in a balanced
    Karras tree, so the spill path is forced instead: with the option stack_lds_depth = 2 every entry below the top three
    goes to the global spill area (max depth on this scene: 12).  Results and counters must not change."""
    st, img = run_case(edge_scenes.deep_stack(3000), 24, 18, 4, stack_lds_depth=2)
    assert st["max_stack"] >= 10, st["max_stack"]
    run_case(edge_scenes.deep_stack(500), 16, 12, 1, stack_lds_depth=0)


def test_quantised_nodes_are_conservative_for_far_ray_origins():
    """Sphere-only scene, camera 30 000 scene sizes away: the quantised records give the bytes of the 64-byte float records
    walked in the reference's order (and of the oracle), with the oracle's visit counters; they are larger boxes -- more visits
    than the exact boxes in the same near-first order (traversal = 2 on the exact records), fewer than the reference's order."""
    text = edge_scenes.far_camera()
    st_q, img_q = run_case(text, 96, 72, 4)                       # default: quantised nodes, near child first
    st_l, img_l = run_case(text, 96, 72, 4, qnodes=0)             # exact records: the reference's order
    st_f, img_f = run_case(text, 96, 72, 4, qnodes=0, traversal=2)
    assert np.array_equal(img_q, img_l) and img_q[..., 3].max() == 255
    assert st_q["rays"] == st_l["rays"] == st_f["rays"] and st_f["internal_visits"] <= st_q["internal_visits"] < st_l["internal_visits"]


def test_a_scene_far_from_the_origin_is_walked_over_the_exact_records():
    """edge_scenes.far_from_origin: coordinates 450 times the scene's extent -- the grid of the quantised records is coarser than
    the rounding of the primitives' own box planes, so they are not used (grid_ok, lbvh_build.hip): the default options walk the
    exact records in the reference's order -- counters equal to the oracle's mirror of exactly that (no QNODES, no ORDERED), and
    to the counters of qnodes = 0."""
    text = edge_scenes.far_from_origin()
    o = ol.OracleScene(pyscene.parse_lines(text.split("\n")), bounds_mode=0)
    assert not o.grid_ok()
    o.close()
    st_q, img_q = run_case(text, 96, 72, 4)
    st_l, img_l = run_case(text, 96, 72, 4, qnodes=0)
    assert np.array_equal(img_q, img_l) and all(st_q[k] == st_l[k] for k in ("rays", "internal_visits", "sphere_tests"))


@pytest.mark.parametrize("name", ["fisheye", "single_triangle", "glass_gi_dof"])
def test_general_kernels_on_scenes_that_have_specialised_ones(name):
    """specialise = 0: the kernels compiled with every feature render the scenes that by default get a kernel without point
    lights / transparency / gi (render.hip, SPEC_*): same oracle pixels and counters."""
    run_case(edge_scenes.ALL[name](), 64, 48, 4, specialise=0)


@pytest.mark.parametrize("seed,triangles,qnodes", [(31, 0.0, 1), (32, 0.0, 1), (33, 1.0, 1), (33, 1.0, 2), (34, 3.0, 2)])
def test_random_scenes_match_the_oracle_with_equal_counters(seed, triangles, qnodes):
    """25 random scenes per seed from tools/fuzz_modes.py's generator (scales 1e-3 .. 1e6, cameras inside / outside / far away,
    fisheye / panorama, depth of field, some with point lights, glass and gi; seeds 33, 34: with triangles -- on the exact records,
    and with qnodes = 2 over the wide quantised ones): colours within 1e-4 of the oracle's, 8-bit image within one level, the
    tree's child links and every ray / node / leaf counter equal."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_modes
    rng = np.random.default_rng(seed)
    for i in range(25):
        text = fuzz_modes.scene_text(rng, triangles)
        spp = int(rng.choice([0, 1, 3]))
        try:
            run_case(text, 48, 32, spp, qnodes=qnodes)
        except AssertionError as e:
            raise AssertionError(f"scene {i} of seed {seed} (spp {spp}): {e}") from e


def test_options_are_validated():
    stl = m.parseText(edge_scenes.single_sphere())
    raw = m.initRawConfigFromStl(stl, 0)
    assert raw.get_option("traversal") == 1 and raw.get_option("wavefront") == 0
    for name, bad in (("reps", 9), ("traversal", 3), ("no_such_option", 1)):
        with pytest.raises(m.MirtError) as e:
            raw.set_option(name, bad)
        assert e.value.status == 3
    raw.close()


def test_null_arrays_and_negative_counts_are_rejected():
    """mirt_scene_create validates the descriptor instead of dereferencing it."""
    import ctypes as C
    stl = m.parseText(edge_scenes.single_sphere())
    d = api.SceneDesc.from_buffer_copy(stl.desc)
    d.spheres = None
    h = C.c_void_p()
    assert m.lib().mirt_scene_create(C.byref(d), 0, C.byref(h)) == 3 and not h.value
    d = api.SceneDesc.from_buffer_copy(stl.desc)
    d.num_planes = -1
    assert m.lib().mirt_scene_create(C.byref(d), 0, C.byref(h)) == 3 and not h.value
    p = api.render_params(16, 16, 5000)
    raw = m.initRawConfigFromStl(stl, 0)
    m.build_lbvh_karas(raw)
    img = torch.empty(16 * 16 * 4, dtype=torch.uint8, device="cuda")
    with pytest.raises(m.MirtError) as e:
        m.render(img, 16, 16, 5000, raw, params=p)
    assert e.value.status == 3
    raw.close()


def test_more_than_64_lights_is_rejected():
    text = edge_scenes.HEADER + "".join(f"sun {i + 1} 1 1\n" for i in range(65)) + "sphere 0 0 -2 1\n"
    stl = m.parseText(text)
    with pytest.raises(m.MirtError):
        m.initRawConfigFromStl(stl, 0)


def test_render_before_build_is_an_error():
    stl = m.parseText(edge_scenes.single_sphere())
    raw = m.initRawConfigFromStl(stl, 0)
    img = torch.empty(16 * 16 * 4, dtype=torch.uint8, device="cuda")
    with pytest.raises(m.MirtError) as e:
        m.render(img, 16, 16, 1, raw)
    assert e.value.status == 6
    raw.close()
