"""ctypes wrapper around oracle/_build/liboracle.so (the CPU oracle; test infrastructure only)."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "_build", "liboracle.so")

FLAG_ANYHIT_SHADOW = 1
FLAG_NORMAL_ZYX = 2
FLAG_ORDERED = 4        # near child first where both subtrees hold spheres only (the product: over the quantised records of a sphere-only scene)
FLAG_ORDERED_ALL = 8    # near child first everywhere
FLAG_SKIP_UNLIT = 16    # shadow rays towards lights the shading normal faces away from are not traced (their term is 0)
# what libmirt does: any-hit shadow rays and no shadow rays to unlit lights, always; near child first over
# the quantised records of a sphere-only scene, the reference's order everywhere else; the hit a quantised walk ends with vetted
# -- the oracle mirrors each so that the visit counters can be compared with ==
FLAG_QNODES = 32        # quantised node records (single-kernel path, traversal >= 1; triangle hits outside their exact leaf box re-walk the exact boxes)
FLAG_WIDE = 64          # wide walk over the quantised records: grandchildren tested per step, reference order (scenes with triangles)
FLAG_REACH = 128        # over quantised boxes: the sphere hit a nearest-hit query ends with (shadow queries to point lights included) is vetted; a ray whose hit the reference may not reach is walked literally
PRODUCT_ALWAYS = FLAG_ANYHIT_SHADOW | FLAG_SKIP_UNLIT | FLAG_REACH      # (the defaults of the scene options shadow_anyhit / skip_unlit; the single-kernel path)
PRODUCT_FLAGS = PRODUCT_ALWAYS | FLAG_ORDERED | FLAG_QNODES   # default options, single-kernel path, a scene WITHOUT triangles
PRODUCT_FLAGS_TRI = PRODUCT_FLAGS | FLAG_WIDE                 # ... a scene with triangles and 65536 primitives or more (or qnodes = 2)
PRODUCT_FLAGS_SMALL_TRI = PRODUCT_ALWAYS                      # ... a smaller scene with triangles: the exact records, the reference's order
REFERENCE_WALK = 0      # {traversal: 0, shadow_anyhit: 0, skip_unlit: 0, qnodes: 0}: draw.cu:292-377 + bvh_traversal.cu:92-183 verbatim


def product_flags(scene_has_triangles, traversal=1, wavefront=False, qnodes=1, shadow_anyhit=True, skip_unlit=True, nprims=0, grid_ok=True):
    """The oracle flags that mirror what libmirt does for an option set (for counters to compare with ==).  qnodes: the scene
    option (0 never, 1 sphere-only scenes and scenes with triangles of 65536 primitives or more, 2 every scene) -- honoured only
    if the grid of the quantised records resolves the scene's coordinates (grid_ok: OracleScene.grid_ok()).  traversal = 1
    reorders only the walk over the quantised records of a sphere-only scene; everything else keeps the reference's order."""
    quantised = grid_ok and bool(qnodes) and not wavefront and not scene_has_triangles and traversal >= 1
    wide = grid_ok and bool(qnodes) and not wavefront and scene_has_triangles and traversal == 1 and (qnodes >= 2 or nprims >= 65536)
    f = {0: 0, 1: FLAG_ORDERED if (quantised or wide) else 0, 2: FLAG_ORDERED_ALL}[traversal]
    if shadow_anyhit:
        f |= FLAG_ANYHIT_SHADOW
    if skip_unlit:
        f |= FLAG_SKIP_UNLIT
    if not wavefront:
        f |= FLAG_REACH
    if quantised:
        f |= FLAG_QNODES
    elif wide:
        f |= FLAG_QNODES | FLAG_WIDE
    return f


class V3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class SceneDesc(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("bounces", C.c_int32), ("aa", C.c_int32),
        ("dof_focus", C.c_float), ("dof_lens", C.c_float),
        ("forward", V3), ("right", V3), ("up", V3), ("eye", V3),
        ("expose", C.c_float),
        ("fisheye", C.c_int32), ("panorama", C.c_int32), ("gi", C.c_int32),
        ("num_spheres", C.c_int32), ("num_triangles", C.c_int32), ("num_prims", C.c_int32),
        ("num_planes", C.c_int32), ("num_suns", C.c_int32), ("num_bulbs", C.c_int32),
        ("spheres", C.c_void_p), ("triangles", C.c_void_p), ("prim_refs", C.c_void_p),
        ("planes", C.c_void_p), ("suns", C.c_void_p), ("bulbs", C.c_void_p),
    ]


NODE = np.dtype([("xmin", "<f4"), ("xmax", "<f4"), ("ymin", "<f4"), ("ymax", "<f4"), ("zmin", "<f4"), ("zmax", "<f4"),
                 ("left", "<u4"), ("right", "<u4"), ("prim_offset", "<u4"), ("count", "<u4")])
HIT = np.dtype([("t", "<f4"), ("kind", "<u4"), ("id", "<u4"), ("n", "<f4", 3)])
STAT_FIELDS = ["samples", "rays", "shadow_rays", "node_iters", "internal_visits", "sphere_tests", "tri_tests",
               "mat_fetches", "max_stack", "prim_hits", "overflow", "qn_retraces", "traversals"]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in STAT_FIELDS]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in STAT_FIELDS}


def build_oracle():
    """Compile the oracle if its .so is missing or stale (building the checker is not using it)."""
    src = [os.path.join(ORACLE_DIR, f) for f in ("oracle.cpp", "omath.h", "Makefile")]
    if os.path.exists(ORACLE_SO) and all(os.path.getmtime(ORACLE_SO) >= os.path.getmtime(s) for s in src):
        return ORACLE_SO
    subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return ORACLE_SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build_oracle()
        L = C.CDLL(ORACLE_SO)
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [C.POINTER(SceneDesc)]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_build_lbvh.argtypes = [C.c_void_p, C.c_int]
        L.orc_num_nodes.argtypes = [C.c_void_p]
        for fn in ("orc_get_nodes", "orc_get_codes", "orc_get_refs"):
            getattr(L, fn).argtypes = [C.c_void_p, C.c_void_p]
        L.orc_get_bounds.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Stats), C.c_uint32, C.c_int]
        L.orc_render_subsample.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Stats), C.c_uint32,
                                           C.c_int, C.POINTER(C.c_float)]
        L.orc_render_accumulate.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_void_p, C.c_uint32, C.c_int]
        L.orc_finalize.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_finalize.restype = None
        L.orc_xorwow_seq.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p]
        L.orc_xorwow_state.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p]
        L.orc_xorwow_floats.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_void_p]
        L.orc_jump_matrix.argtypes = [C.c_int, C.c_void_p]
        L.orc_math_probe.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_morton.restype = C.c_uint32
        L.orc_morton.argtypes = [C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.orc_sizeof.argtypes = [C.c_int]
        _lib = L
    return _lib


def _vec(v):
    return V3(float(v[0]), float(v[1]), float(v[2]))


def make_desc(sc, arrays, desc_cls=SceneDesc):
    """Fill a scene descriptor from a tests.pyscene.PyScene (or anything with the same attributes)."""
    d = desc_cls()
    d.width, d.height, d.bounces, d.aa = sc.width, sc.height, sc.bounces, sc.aa
    d.dof_focus, d.dof_lens = float(sc.dof_focus), float(sc.dof_lens)
    d.forward, d.right, d.up, d.eye = _vec(sc.forward), _vec(sc.right), _vec(sc.up), _vec(sc.eye)
    d.expose = float(sc.expose)
    d.fisheye, d.panorama, d.gi = int(sc.fisheye), int(sc.panorama), sc.gi
    d.num_spheres, d.num_triangles, d.num_prims = len(arrays["spheres"]), len(arrays["triangles"]), len(arrays["refs"])
    d.num_planes, d.num_suns, d.num_bulbs = len(arrays["planes"]), len(arrays["suns"]), len(arrays["bulbs"])
    for name, key in (("spheres", "spheres"), ("triangles", "triangles"), ("prim_refs", "refs"),
                      ("planes", "planes"), ("suns", "suns"), ("bulbs", "bulbs")):
        a = arrays[key]
        setattr(d, name, a.ctypes.data if len(a) else None)
    return d


class OracleScene:
    """Owns an oracle scene handle.  bounds_mode 0 = true scene bounds, 1 = as shipped (all Morton codes 0)."""

    def __init__(self, pyscene, bounds_mode=0, build=True):
        self.sc = pyscene
        self.arrays = pyscene.arrays()
        self.desc = make_desc(pyscene, self.arrays)
        self.h = lib().orc_scene_create(C.byref(self.desc))
        self.n = len(self.arrays["refs"])
        if build:
            assert lib().orc_build_lbvh(self.h, bounds_mode) == 0

    def close(self):
        if self.h:
            lib().orc_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def nodes(self):
        n = lib().orc_num_nodes(self.h)
        out = np.zeros(n, dtype=NODE)
        lib().orc_get_nodes(self.h, out.ctypes.data)
        return out

    def codes(self):
        out = np.zeros(self.n, dtype=np.uint32)
        lib().orc_get_codes(self.h, out.ctypes.data)
        return out

    def refs(self):
        out = np.zeros(self.n, dtype=[("type", "<u4"), ("id", "<u4")])
        lib().orc_get_refs(self.h, out.ctypes.data)
        return out

    def grid_ok(self):
        """lbvh_build.hip: every axis of the scene box has its coordinates below 64 extents (float arithmetic as there)."""
        mn, mx = self.bounds()
        ext = (mx - mn).astype(np.float32)
        return bool(np.all(np.maximum(np.abs(mn), np.abs(mx)) <= np.float32(64.0) * ext))

    def bounds(self):
        mn = np.zeros(3, np.float32)
        mx = np.zeros(3, np.float32)
        lib().orc_get_bounds(self.h, mn.ctypes.data, mx.ctypes.data)
        return mn, mx

    def render(self, width, height, spp, tile=None, flags=0, nthreads=1, want_aov=False):
        x0, y0, tw, th = tile if tile else (0, 0, width, height)
        f = np.zeros((th, tw, 4), np.float32)
        u = np.zeros((th, tw, 4), np.uint8)
        aov = np.zeros((th, tw), HIT) if want_aov else None
        st = Stats()
        rc = lib().orc_render(self.h, width, height, spp, x0, y0, tw, th, f.ctypes.data, u.ctypes.data,
                              aov.ctypes.data if want_aov else None, C.byref(st), flags, nthreads)
        assert rc == 0
        return dict(f32=f, u8=u, aov=aov, stats=st.as_dict())

    def render_accumulate(self, accum, width, height, first, count, tile=None, flags=0, nthreads=1):
        """Adds samples [first, first + count) of every pixel of the tile to accum ([th, tw, 4] float32, in place)."""
        x0, y0, tw, th = tile if tile else (0, 0, width, height)
        assert accum.shape == (th, tw, 4) and accum.dtype == np.float32 and accum.flags["C_CONTIGUOUS"]
        assert lib().orc_render_accumulate(self.h, width, height, x0, y0, tw, th, first, count, accum.ctypes.data, flags, nthreads) == 0

    @staticmethod
    def finalize(accum, total):
        out = np.zeros(accum.shape[:-1] + (4,), np.uint8)
        lib().orc_finalize(np.ascontiguousarray(accum).ctypes.data, accum.size // 4, total, out.ctypes.data)
        return out

    def render_subsample(self, width, height, spp, step, flags=0, nthreads=1):
        st = Stats()
        cs = C.c_float(0)
        rc = lib().orc_render_subsample(self.h, width, height, spp, step, C.byref(st), flags, nthreads, C.byref(cs))
        assert rc == 0
        return st.as_dict(), float(cs.value)


class ArrayScene:
    """Adapter: builds an oracle scene from a product StlConfig (arrays + scalars), e.g. the synthetic scene, so that
    the oracle sees exactly the bytes the product uploaded."""

    def __init__(self, stl):
        self.width, self.height, self.bounces, self.aa = stl.width, stl.height, stl.bounces, stl.aa
        self.dof_focus, self.dof_lens, self.expose = stl.dof_focus, stl.dof_lens, stl.expose
        self.fisheye, self.panorama, self.gi = bool(stl.fisheye), bool(stl.panorama), stl.gi
        d = stl.desc
        self.forward, self.right, self.up, self.eye = d.forward.tolist(), d.right.tolist(), d.up.tolist(), d.eye.tolist()
        self._arrays = dict(spheres=stl.array("spheres"), triangles=stl.array("triangles"), refs=stl.array("prim_refs"),
                            planes=stl.array("planes"), suns=stl.array("suns"), bulbs=stl.array("bulbs"))

    def arrays(self):
        return self._arrays
