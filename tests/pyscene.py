"""Test-side scene parser: an independent Python restatement of the reference grammar
(parse.cpp:41-222, object.cuh:136-141,177-191) in float32 arithmetic,
one rounding per source operation.  It feeds the oracle directly (so oracle tests do not depend
on the product's C++ parser) and cross-checks that parser bit for bit in test_parser.py.
"""
import math
import numpy as np

f32 = np.float32

MAT = np.dtype([("color", "<f4", 3), ("shininess", "<f4", 3), ("trans", "<f4", 3), ("ior", "<f4"), ("roughness", "<f4")])
SPHERE = np.dtype([("c", "<f4", 3), ("r", "<f4"), ("mat", MAT)])
TRIANGLE = np.dtype([("p0", "<f4", 3), ("p1", "<f4", 3), ("p2", "<f4", 3), ("nor", "<f4", 3), ("e1", "<f4", 3), ("e2", "<f4", 3), ("mat", MAT)])
PLANE = np.dtype([("abcd", "<f4", 4), ("nor", "<f4", 3), ("point", "<f4", 3), ("mat", MAT)])
LIGHT = np.dtype([("v", "<f4", 3), ("color", "<f4", 3)])
PRIMREF = np.dtype([("type", "<u4"), ("id", "<u4")])
assert MAT.itemsize == 44 and SPHERE.itemsize == 60 and TRIANGLE.itemsize == 116 and PLANE.itemsize == 84
assert LIGHT.itemsize == 24 and PRIMREF.itemsize == 8


def _v(x, y, z):
    return np.array([x, y, z], dtype=f32)


def _dot(a, b):
    return f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2]))


def _cross(a, b):
    return _v(f32(a[1] * b[2]) - f32(a[2] * b[1]), f32(a[2] * b[0]) - f32(a[0] * b[2]), f32(a[0] * b[1]) - f32(a[1] * b[0]))


def _fequal(a, b, eps=f32(1e-6)):
    diff = f32(abs(f32(a - b)))
    largest = f32(max(abs(a), abs(b)))
    if largest < f32(1e-6):
        return diff < eps
    return f32(diff / largest) < eps


def _normalize(a):
    mag = f32(np.sqrt(f32(f32(f32(a[0] * a[0]) + f32(a[1] * a[1])) + f32(a[2] * a[2]))))
    if _fequal(mag, f32(0.0)):
        return _v(0, 0, 0)
    inv = f32(f32(1.0) / mag)
    return _v(a[0] * inv, a[1] * inv, a[2] * inv)


def _stof(w):
    return f32(float(w))


class PyScene:
    def __init__(self):
        self.width = 0
        self.height = 0
        self.filename = "file.txt"
        self.color = _v(1, 1, 1)
        self.bounces = 4
        self.aa = 0
        self.dof_focus = f32(0)
        self.dof_lens = f32(0)
        self.forward = _v(0, 0, -1)
        self.right = _v(1, 0, 0)
        self.up = _v(0, 1, 0)
        self.eye = _v(0, 0, 0)
        self.target_up = _v(0, 1, 0)
        self.expose = f32(np.inf)
        self.fisheye = False
        self.panorama = False
        self.ior = f32(1.458)
        self.rough = f32(0)
        self.gi = 0
        self.trans = _v(0, 0, 0)
        self.shine = _v(0, 0, 0)
        self.spheres = []
        self.triangles = []
        self.refs = []
        self.planes = []
        self.suns = []
        self.bulbs = []
        self.vertices = []

    def _mat(self, color):
        m = np.zeros((), dtype=MAT)
        m["color"] = color
        m["shininess"] = self.shine
        m["trans"] = self.trans
        m["ior"] = self.ior
        m["roughness"] = self.rough
        return m

    def arrays(self):
        def arr(lst, dt):
            a = np.zeros(len(lst), dtype=dt)
            for i, x in enumerate(lst):
                a[i] = x
            return a
        return dict(spheres=arr(self.spheres, SPHERE), triangles=arr(self.triangles, TRIANGLE),
                    refs=arr(self.refs, PRIMREF), planes=arr(self.planes, PLANE),
                    suns=arr(self.suns, LIGHT), bulbs=arr(self.bulbs, LIGHT))


def parse_lines(lines):
    s = PyScene()
    for line in lines:
        w = line.split()
        if not w:
            continue
        k, n = w[0], len(w)
        if k == "png" and n == 4:
            s.width, s.height, s.filename = int(w[1]), int(w[2]), w[3]
        elif k == "bounces" and n == 2:
            s.bounces = int(w[1])
        elif k == "forward" and n == 4:
            s.forward = _v(_stof(w[1]), _stof(w[2]), _stof(w[3]))
            s.right = _normalize(_cross(s.forward, s.up))
            s.up = _normalize(_cross(s.right, s.forward))
        elif k == "up" and n == 4:
            s.target_up = _v(_stof(w[1]), _stof(w[2]), _stof(w[3]))
            s.right = _normalize(_cross(s.forward, s.target_up))
            s.up = _normalize(_cross(s.right, s.forward))
        elif k == "eye" and n == 4:
            s.eye = _v(_stof(w[1]), _stof(w[2]), _stof(w[3]))
        elif k == "expose" and n == 2:
            s.expose = _stof(w[1])
        elif k == "dof" and n == 3:
            s.dof_focus, s.dof_lens = _stof(w[1]), _stof(w[2])
        elif k == "aa" and n == 2:
            s.aa = int(w[1])
        elif k == "panorama" and n == 1:
            s.panorama = True
        elif k == "fisheye" and n == 1:
            s.fisheye = True
        elif k == "gi" and n == 2:
            s.gi = int(w[1])
        elif k == "color" and n == 4:
            s.color = _v(_stof(w[1]), _stof(w[2]), _stof(w[3]))
        elif k == "roughness" and n == 2:
            s.rough = _stof(w[1])
        elif k == "shininess" and n == 2:
            s.shine = _v(_stof(w[1]), _stof(w[1]), _stof(w[1]))
        elif k == "shininess" and n == 4:
            s.shine = _v(_stof(w[1]), _stof(w[2]), _stof(w[3]))
        elif k == "transparency" and n == 2:
            s.trans = _v(_stof(w[1]), _stof(w[1]), _stof(w[1]))
        elif k == "transparency" and n == 4:
            s.trans = _v(_stof(w[1]), _stof(w[2]), _stof(w[3]))
        elif k == "ior" and n == 2:
            s.ior = _stof(w[1])
        elif k == "sphere" and n == 5:
            sp = np.zeros((), dtype=SPHERE)
            sp["c"] = _v(_stof(w[1]), _stof(w[2]), _stof(w[3]))
            sp["r"] = _stof(w[4])
            sp["mat"] = s._mat(s.color)
            s.spheres.append(sp)
            s.refs.append(np.array((0, len(s.spheres) - 1), dtype=PRIMREF))
        elif k == "plane" and n == 5:
            a, b, c, d = (_stof(x) for x in w[1:5])
            pl = np.zeros((), dtype=PLANE)
            pl["abcd"] = [a, b, c, d]
            pl["nor"] = _normalize(_v(a, b, c))
            # object.cuh:139: host pow(float,int) is double; the sum is rounded to float at operator/
            den = f32(float(a) ** 2 + float(b) ** 2 + float(c) ** 2)
            nd = f32(-d)
            pl["point"] = _v(f32(f32(a * nd) / den), f32(f32(b * nd) / den), f32(f32(c * nd) / den))
            pl["mat"] = s._mat(s.color)
            s.planes.append(pl)
        elif k == "xyz" and n == 4:
            s.vertices.append(_v(_stof(w[1]), _stof(w[2]), _stof(w[3])))
        elif k == "tri" and n == 4:
            size = len(s.vertices)
            idx = []
            for x in w[1:4]:
                v = int(x)
                idx.append(v - 1 if v > 0 else size + v)
            p0, p1, p2 = (s.vertices[i] for i in idx)
            tr = np.zeros((), dtype=TRIANGLE)
            tr["p0"], tr["p1"], tr["p2"] = p0, p1, p2
            nor = _normalize(_cross(p1 - p0, p2 - p0))
            a1 = _cross(p2 - p0, nor)
            a2 = _cross(p1 - p0, nor)
            k1 = f32(f32(1) / _dot(a1, p1 - p0))
            k2 = f32(f32(1) / _dot(a2, p2 - p0))
            tr["nor"] = nor
            tr["e1"] = _v(a1[0] * k1, a1[1] * k1, a1[2] * k1)
            tr["e2"] = _v(a2[0] * k2, a2[1] * k2, a2[2] * k2)
            tr["mat"] = s._mat(s.color)
            s.triangles.append(tr)
            s.refs.append(np.array((1, len(s.triangles) - 1), dtype=PRIMREF))
        elif k == "sun" and n == 4:
            li = np.zeros((), dtype=LIGHT)
            li["v"] = _v(_stof(w[1]), _stof(w[2]), _stof(w[3]))
            li["color"] = s.color
            s.suns.append(li)
        elif k == "bulb" and n == 4:
            li = np.zeros((), dtype=LIGHT)
            li["v"] = _v(_stof(w[1]), _stof(w[2]), _stof(w[3]))
            li["color"] = s.color
            s.bulbs.append(li)
        else:
            raise ValueError("One of the lines are not valid.")
    return s


def parse_file(path):
    with open(path, "r") as f:
        return parse_lines(f.read().split("\n"))
