"""Host logic: the product's C++ scene parser (parse.cpp:16-222 rewritten) against the independent Python restatement
in tests/pyscene.py, bit for bit, plus the grammar's edge cases and the reference's error behaviour.  No GPU needed."""
import numpy as np
import pytest

import cuda_ray_tracer_amd as m
import pyscene
from conftest import scene_path

PAIRS = (("spheres", "spheres"), ("triangles", "triangles"), ("prim_refs", "refs"), ("planes", "planes"), ("suns", "suns"), ("bulbs", "bulbs"))


def assert_same(stl, py):
    arr = py.arrays()
    for a, b in PAIRS:
        assert stl.array(a).tobytes() == arr[b].tobytes(), a
    for f in ("forward", "right", "up", "eye"):
        assert np.array_equal(np.array(getattr(stl.desc, f).tolist(), np.float32).view(np.uint32), getattr(py, f).view(np.uint32)), f
    for f in ("width", "height", "bounces", "aa", "gi"):
        assert getattr(stl, f) == getattr(py, f), f
    assert np.float32(stl.dof_focus) == py.dof_focus and np.float32(stl.dof_lens) == py.dof_lens
    assert (np.isinf(stl.expose) and np.isinf(py.expose)) or np.float32(stl.expose) == py.expose
    assert bool(stl.fisheye) == py.fisheye and bool(stl.panorama) == py.panorama
    assert stl.filename == py.filename


@pytest.mark.parametrize("name,counts", [("tri", (3, 2, 0, 1)), ("redchair", (2, 1715, 1, 2)), ("spiral", (5000, 0, 1, 2)), ("tenthousand", (10000, 0, 1, 2))])
def test_bundled_scenes_parse_identically(name, counts):
    stl = m.parseInput(scene_path(name))
    assert (stl.num_spheres, stl.num_triangles, stl.num_planes, stl.num_suns) == counts
    assert_same(stl, pyscene.parse_file(scene_path(name)))


GRAMMAR = """png 320 200 out.png
bounces 7
eye 1 2 3
forward -.3 -.75 -.8
up 0 1 0.1
expose 2.5
dof 2.25 0.01
aa 9
gi 3
color 0.25 0.5 0.75
shininess 0.3
roughness 0.05
sphere 0 0 -1 0.5
shininess 0.1 0.2 0.3
transparency 0.4
ior 1.33
xyz 0 0 0
xyz 1 0 0
xyz 0 1 0
xyz 0 0 1
tri 1 2 3
tri -1 -2 -3
tri 1 -1 2
transparency 0.1 0.2 0.3
plane 1 2 3 4
color 1 1 1
sun 1 1 1
bulb 0 5 0

sphere 1 1 1 1
"""


def test_grammar_state_machine_and_relative_indices():
    stl = m.parseText(GRAMMAR)
    py = pyscene.parse_lines(GRAMMAR.split("\n"))
    assert_same(stl, py)
    tri = stl.array("triangles")
    verts = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1)]
    assert tri["p0"][1].tolist() == list(verts[3]) and tri["p1"][1].tolist() == list(verts[2]) and tri["p2"][1].tolist() == list(verts[1])
    sp = stl.array("spheres")
    assert np.allclose(sp["mat"]["shininess"][0], 0.3) and np.allclose(sp["mat"]["trans"][0], 0.0)
    assert np.allclose(sp["mat"]["trans"][1], [0.1, 0.2, 0.3]) and np.isclose(sp["mat"]["ior"][1], 1.33)
    assert stl.array("prim_refs")["type"].tolist() == [0, 1, 1, 1, 0]
    assert stl.array("suns")["color"][0].tolist() == [1, 1, 1]


def test_defaults():
    stl = m.parseText("png 10 10 a.png\nsphere 0 0 -1 1\n")
    assert stl.bounces == 4 and stl.aa == 0 and stl.gi == 0 and np.isinf(stl.expose)
    assert stl.desc.forward.tolist() == [0, 0, -1] and stl.desc.up.tolist() == [0, 1, 0] and stl.desc.right.tolist() == [1, 0, 0]
    mat = stl.array("spheres")["mat"][0]
    assert mat["color"].tolist() == [1, 1, 1] and np.isclose(mat["ior"], 1.458) and mat["roughness"] == 0


@pytest.mark.parametrize("text", ["foo 1 2\n", "sphere 1 2 3\n", "png 1 2\n", "tri 1 2 9\nxyz 0 0 0\n", "color a b c\n", "aa\n"])
def test_invalid_lines_are_rejected_like_the_reference(text):
    with pytest.raises(m.MirtError) as e:
        m.parseText(text)
    assert e.value.status == 2 and e.value.message == "One of the lines are not valid."


def test_missing_file_message():
    with pytest.raises(m.MirtError) as e:
        m.parseInput("/nonexistent/scene.txt")
    assert e.value.status == 1 and e.value.message == "Error opening file..."


def test_synthetic_scene_is_deterministic_and_shaped():
    a = m.syntheticScene(2000, 3000, seed=1234)
    b = m.syntheticScene(2000, 3000, seed=1234)
    assert (a.num_spheres, a.num_triangles, a.num_prims, a.num_planes, a.num_suns) == (2000, 3000, 5000, 1, 2)
    assert a.array("spheres").tobytes() == b.array("spheres").tobytes() and a.array("triangles").tobytes() == b.array("triangles").tobytes()
    s = a.array("spheres")
    assert s["c"][:, 0].min() >= -50 and s["c"][:, 0].max() <= 50 and s["c"][:, 2].min() >= -150 and s["c"][:, 2].max() <= -50
    assert s["r"].min() >= 0.05 and s["r"].max() <= 0.25
    assert m.syntheticScene(10, 10, seed=1).array("spheres").tobytes() != m.syntheticScene(10, 10, seed=2).array("spheres").tobytes()
