"""Python host side of the MI355X ray tracer: a thin ctypes binding over the C ABI in include/mirt.h.

The functions mirror the reference's host interface for the hot path (main.cu:25-94), same names and
argument meaning:

    parseInput(path)                        -> StlConfig          parse.hpp:10
    initRawConfigFromStl(stl) +
    copyConfigDataToDevice(stl, raw)        -> RawConfig          config_utils.cuh:11-17
    build_lbvh_karas(raw, morton_bits=30)                         lbvh_builder.cuh:14
    render(d_image, w, h, aa, raw)                                draw.cuh:10
    freeRawConfigDeviceMemory(raw)                                config_utils.cuh:20

Device memory, streams and torch.distributed come from PyTorch (plumbing only); every computation runs in the
hand-written HIP kernels of libmirt.so.  There is no CPU fallback: if the library is missing or no GPU is present
the calls raise.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# MIRT_LIB: load another build of the same library (tools/ab.py builds A/B variants next to the default one)
LIB_PATH = os.environ.get("MIRT_LIB") or os.path.join(HERE, "_build", "libmirt.so")

MIRT_RENDER_COUNTERS = 1


class MirtError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"libmirt status {status}: {message}")
        self.status = status
        self.message = message


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]

    def tolist(self):
        return [self.x, self.y, self.z]


class SceneDesc(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("bounces", C.c_int32), ("aa", C.c_int32),
        ("dof_focus", C.c_float), ("dof_lens", C.c_float),
        ("forward", Vec3), ("right", Vec3), ("up", Vec3), ("eye", Vec3),
        ("expose", C.c_float),
        ("fisheye", C.c_int32), ("panorama", C.c_int32), ("gi", C.c_int32),
        ("num_spheres", C.c_int32), ("num_triangles", C.c_int32), ("num_prims", C.c_int32),
        ("num_planes", C.c_int32), ("num_suns", C.c_int32), ("num_bulbs", C.c_int32),
        ("spheres", C.c_void_p), ("triangles", C.c_void_p), ("prim_refs", C.c_void_p),
        ("planes", C.c_void_p), ("suns", C.c_void_p), ("bulbs", C.c_void_p),
    ]


class RenderParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_int32),
                ("stripe_rows", C.c_int32), ("num_parts", C.c_int32), ("part", C.c_int32), ("flags", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("samples", "rays", "shadow_rays", "internal_visits", "sphere_tests",
                                          "tri_tests", "mat_fetches", "max_stack", "overflow_events")] + \
               [("trace_kernel_ms", C.c_float), ("render_ms", C.c_float), ("build_ms", C.c_float), ("num_nodes", C.c_int32),
                ("trace_kernel_ms_mean", C.c_float), ("frames_timed", C.c_int32), ("trace_launches", C.c_int32), ("node_record_bytes", C.c_int32),
                ("rays_traversed", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


MULTI_MAX_GPUS = 16


class MultiStats(C.Structure):
    _fields_ = [("num_gpus", C.c_int32), ("build_ms", C.c_float), ("render_ms", C.c_float * MULTI_MAX_GPUS),
                ("gather_ms", C.c_float), ("frame_ms", C.c_float)]


class TreeNode(C.Structure):
    _fields_ = [("xmin", C.c_float), ("xmax", C.c_float), ("ymin", C.c_float), ("ymax", C.c_float),
                ("zmin", C.c_float), ("zmax", C.c_float),
                ("left", C.c_uint32), ("right", C.c_uint32), ("prim_offset", C.c_uint32), ("count", C.c_uint32)]


EXPORTS = [
    "mirt_last_error", "mirt_version", "mirt_parse_scene_file", "mirt_parse_scene_text", "mirt_synthetic_scene",
    "mirt_host_scene_destroy", "mirt_host_scene_desc", "mirt_host_scene_filename", "mirt_scene_create",
    "mirt_scene_destroy", "mirt_scene_set_option", "mirt_scene_get_option", "mirt_build_lbvh", "mirt_render_num_pixels", "mirt_render", "mirt_render_accumulate", "mirt_finalize", "mirt_scatter_part",
    "mirt_get_stats", "mirt_get_tree", "mirt_probe_math", "mirt_probe_xorwow", "mirt_write_png",
    "mirt_multi_create", "mirt_multi_destroy", "mirt_multi_num_parts", "mirt_multi_set_option", "mirt_render_frame_multi",
    "mirt_multi_submit", "mirt_multi_wait", "mirt_render_frames_multi", "mirt_multi_get_stats", "mirt_part_pixel_xy",
]

_lib = None


def lib():
    """Load libmirt.so (fails loudly if it has not been built: `python -m cuda_ray_tracer_amd.build`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MirtError(-1, f"{LIB_PATH} not found: build it with `python -m cuda_ray_tracer_amd.build` "
                            "(there is no fallback implementation)")
    try:
        import torch  # noqa: F401  (loads the process-wide HIP runtime first so both share one instance)
    except Exception:
        pass
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    L.mirt_last_error.restype = C.c_char_p
    L.mirt_parse_scene_file.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
    L.mirt_parse_scene_text.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p)]
    L.mirt_synthetic_scene.argtypes = [C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.mirt_host_scene_destroy.argtypes = [C.c_void_p]
    L.mirt_host_scene_destroy.restype = None
    L.mirt_host_scene_desc.argtypes = [C.c_void_p, C.POINTER(SceneDesc)]
    L.mirt_host_scene_filename.argtypes = [C.c_void_p]
    L.mirt_host_scene_filename.restype = C.c_char_p
    L.mirt_scene_create.argtypes = [C.POINTER(SceneDesc), C.c_int, C.POINTER(C.c_void_p)]
    L.mirt_scene_destroy.argtypes = [C.c_void_p]
    L.mirt_scene_destroy.restype = None
    L.mirt_scene_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    L.mirt_scene_get_option.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]
    L.mirt_build_lbvh.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
    L.mirt_render_num_pixels.argtypes = [C.POINTER(RenderParams)]
    L.mirt_render_num_pixels.restype = C.c_int64
    L.mirt_render.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.c_void_p, C.c_void_p]
    L.mirt_render_accumulate.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.mirt_finalize.argtypes = [C.POINTER(RenderParams), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.mirt_scatter_part.argtypes = [C.POINTER(RenderParams), C.c_void_p, C.c_void_p, C.c_void_p]
    L.mirt_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
    L.mirt_get_tree.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mirt_probe_math.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.mirt_probe_xorwow.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.mirt_write_png.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
    L.mirt_multi_create.argtypes = [C.POINTER(SceneDesc), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]
    L.mirt_multi_destroy.argtypes = [C.c_void_p]
    L.mirt_multi_destroy.restype = None
    L.mirt_multi_num_parts.argtypes = [C.c_void_p]
    L.mirt_multi_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    L.mirt_render_frame_multi.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(MultiStats)]
    L.mirt_multi_submit.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_uint64)]
    L.mirt_multi_wait.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(MultiStats)]
    L.mirt_render_frames_multi.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(MultiStats), C.POINTER(C.c_float)]
    L.mirt_multi_get_stats.argtypes = [C.c_void_p, C.c_int, C.POINTER(Stats)]
    L.mirt_part_pixel_xy.argtypes = [C.POINTER(RenderParams), C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise MirtError(rc, lib().mirt_last_error().decode("utf-8", "replace"))


# ------------------------------------------------------------------------------------------------------
# StlConfig: the parsed scene on the host (config.hpp:24-73)
# ------------------------------------------------------------------------------------------------------
class StlConfig:
    def __init__(self, handle):
        self._h = C.c_void_p(handle)
        self.desc = SceneDesc()
        _check(lib().mirt_host_scene_desc(self._h, C.byref(self.desc)))
        self.filename = lib().mirt_host_scene_filename(self._h).decode()
        for name in ("width", "height", "bounces", "aa", "dof_focus", "dof_lens", "expose", "fisheye", "panorama", "gi",
                     "num_spheres", "num_triangles", "num_prims", "num_planes", "num_suns", "num_bulbs"):
            setattr(self, name, getattr(self.desc, name))

    def close(self):
        if self._h:
            lib().mirt_host_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def array(self, which):
        """Host arrays as numpy structured views (spheres, triangles, prim_refs, planes, suns, bulbs)."""
        import numpy as np
        from . import layouts
        dt, cnt, ptr = {
            "spheres": (layouts.SPHERE, self.desc.num_spheres, self.desc.spheres),
            "triangles": (layouts.TRIANGLE, self.desc.num_triangles, self.desc.triangles),
            "prim_refs": (layouts.PRIMREF, self.desc.num_prims, self.desc.prim_refs),
            "planes": (layouts.PLANE, self.desc.num_planes, self.desc.planes),
            "suns": (layouts.LIGHT, self.desc.num_suns, self.desc.suns),
            "bulbs": (layouts.LIGHT, self.desc.num_bulbs, self.desc.bulbs),
        }[which]
        if cnt == 0:
            return np.zeros(0, dtype=dt)
        buf = (C.c_char * (cnt * dt.itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dt, count=cnt).copy()


def parseInput(path):
    """parseInput(argv, StlConfig&), parse.cpp:16-39.  Raises MirtError with the reference's messages
    ("Error opening file...", "One of the lines are not valid.") where the reference prints them and exits."""
    h = C.c_void_p()
    _check(lib().mirt_parse_scene_file(os.fsencode(path), C.byref(h)))
    return StlConfig(h.value)


def parseText(text):
    data = text.encode()
    h = C.c_void_p()
    _check(lib().mirt_parse_scene_text(data, len(data), C.byref(h)))
    return StlConfig(h.value)


def syntheticScene(num_spheres=1_000_000, num_triangles=1_000_000, seed=1234):
    """The synthetic stress scene of BASELINE config 5."""
    h = C.c_void_p()
    _check(lib().mirt_synthetic_scene(seed, num_spheres, num_triangles, C.byref(h)))
    return StlConfig(h.value)


# ------------------------------------------------------------------------------------------------------
# RawConfig: the device-resident scene (config.hpp:75-126)
# ------------------------------------------------------------------------------------------------------
class RawConfig:
    def __init__(self, stl_or_desc, device=0):
        desc = stl_or_desc.desc if hasattr(stl_or_desc, "desc") else stl_or_desc
        self._keep = stl_or_desc
        self.desc = desc
        self.device = device
        h = C.c_void_p()
        _check(lib().mirt_scene_create(C.byref(desc), device, C.byref(h)))
        self._h = h
        self.build_ms = None

    def close(self):
        if getattr(self, "_h", None):
            lib().mirt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stats(self):
        """MirtStats as a dict.  Raises MirtError (status 6) when a capacity overflow was recorded during a render."""
        st = Stats()
        _check(lib().mirt_get_stats(self._h, C.byref(st)))
        return st.as_dict()

    def set_option(self, name, value):
        """mirt_scene_set_option: the build option "bounds_as_shipped" takes effect at the next build_lbvh_karas, every other one at the next render."""
        _check(lib().mirt_scene_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name):
        v = C.c_int(0)
        _check(lib().mirt_scene_get_option(self._h, name.encode(), C.byref(v)))
        return v.value

    def tree(self):
        """(nodes, codes, refs, bounds) in the reference's numbering -- for parity tests."""
        import numpy as np
        from . import layouts
        n = self.desc.num_prims
        nodes = np.zeros(max(2 * n - 1, 0), dtype=layouts.TREENODE)
        codes = np.zeros(n, dtype=np.uint32)
        refs = np.zeros(n, dtype=layouts.PRIMREF)
        bounds = np.zeros(6, dtype=np.float32)
        _check(lib().mirt_get_tree(self._h, nodes.ctypes.data if n else None, codes.ctypes.data if n else None,
                                   refs.ctypes.data if n else None, bounds.ctypes.data))
        return nodes, codes, refs, bounds


def initRawConfigFromStl(stl, device=0):
    """initRawConfigFromStl + copyConfigDataToDevice (config_utils.cu:18-199): uploads the scene."""
    return RawConfig(stl, device)


copyConfigDataToDevice = initRawConfigFromStl


def freeRawConfigDeviceMemory(raw):
    raw.close()


def part_pixel_xy(params, local):
    """mirt_part_pixel_xy: frame coordinates of local pixel `local` of a part's compact buffer (host arithmetic, no GPU)."""
    x, y = C.c_int32(0), C.c_int32(0)
    _check(lib().mirt_part_pixel_xy(C.byref(params), local, C.byref(x), C.byref(y)))
    return x.value, y.value


class MultiGpu:
    """mirt_multi_*: one process, several GPUs, RCCL framebuffer gather (include/mirt.h)."""

    def __init__(self, stl, ngpu=1, devices=None):
        self._keep = stl
        dv = (C.c_int * ngpu)(*devices) if devices is not None else None
        h = C.c_void_p()
        _check(lib().mirt_multi_create(C.byref(stl.desc), ngpu, dv, C.byref(h)))
        self._h = h

    def set_option(self, name, value):
        """mirt_multi_set_option: the option on every device's scene."""
        _check(lib().mirt_multi_set_option(self._h, name.encode(), int(value)))

    def render_frame(self, width, height, spp, stripe_rows=4):
        import numpy as np
        out = np.zeros((height, width, 4), np.uint8)
        st = MultiStats()
        _check(lib().mirt_render_frame_multi(self._h, width, height, spp, stripe_rows, out.ctypes.data, C.byref(st)))
        n = st.num_gpus
        return out, dict(num_gpus=n, build_ms=st.build_ms, render_ms=list(st.render_ms)[:n], gather_ms=st.gather_ms, frame_ms=st.frame_ms)

    @staticmethod
    def _stats(st):
        n = st.num_gpus
        return dict(num_gpus=n, build_ms=st.build_ms, render_ms=list(st.render_ms)[:n], gather_ms=st.gather_ms, frame_ms=st.frame_ms)

    def submit(self, width, height, spp, stripe_rows=4, out=None):
        """mirt_multi_submit: issues a frame, returns its ticket.  `out` (numpy uint8 [height, width, 4], kept alive by the caller
        until wait) receives the frame."""
        t = C.c_uint64(0)
        _check(lib().mirt_multi_submit(self._h, width, height, spp, stripe_rows, out.ctypes.data if out is not None else None, C.byref(t)))
        return t.value

    def wait(self, ticket):
        st = MultiStats()
        _check(lib().mirt_multi_wait(self._h, ticket, C.byref(st)))
        return self._stats(st)

    def render_frames(self, width, height, spp, nframes, in_flight=2, stripe_rows=4):
        """mirt_render_frames_multi: nframes frames back to back, `in_flight` of them in flight; (last frame, its stats, ms per frame)."""
        import numpy as np
        out = np.zeros((height, width, 4), np.uint8)
        st = MultiStats()
        ms = C.c_float(0)
        _check(lib().mirt_render_frames_multi(self._h, width, height, spp, stripe_rows, nframes, in_flight, out.ctypes.data, C.byref(st), C.byref(ms)))
        return out, self._stats(st), ms.value

    def stats(self, part):
        st = Stats()
        _check(lib().mirt_multi_get_stats(self._h, part, C.byref(st)))
        return st.as_dict()

    def close(self):
        if getattr(self, "_h", None):
            lib().mirt_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _stream_ptr(stream):
    if stream is None:
        try:
            import torch
            return C.c_void_p(torch.cuda.current_stream().cuda_stream)
        except Exception:
            return None
    if hasattr(stream, "cuda_stream"):
        return C.c_void_p(stream.cuda_stream)
    return C.c_void_p(int(stream))


def build_lbvh_karas(raw, morton_bits=30, stream=None):
    """build_lbvh_karas(RawConfig&, int morton_bits = 30), lbvh_builder.cu:401-521.  morton_bits is accepted and
    ignored exactly as in the reference (10 bits per axis are hard-coded there, lbvh_utils.cu:84)."""
    ms = C.c_float(0)
    _check(lib().mirt_build_lbvh(raw._h, _stream_ptr(stream), C.byref(ms)))
    raw.build_ms = ms.value
    return ms.value


def render_params(width, height, aa, stripe_rows=None, num_parts=1, part=0, counters=False):
    p = RenderParams()
    p.width, p.height, p.spp = width, height, aa
    p.stripe_rows = stripe_rows if stripe_rows else height
    p.num_parts, p.part = num_parts, part
    p.flags = MIRT_RENDER_COUNTERS if counters else 0
    return p


def num_pixels(params):
    n = lib().mirt_render_num_pixels(C.byref(params))
    if n < 0:
        raise MirtError(3, "bad render parameters")
    return n


def render(d_image, img_width, img_height, aa, raw, d_float=None, params=None, stream=None):
    """render(pixel_t* d_image, w, h, aa, RawConfig*), draw.cu:215-239.

    d_image: a CUDA/HIP uint8 tensor (or raw device pointer) of num_pixels*4 bytes, RGBA.
    d_float: optional float32 tensor of num_pixels*4 -- the linear sample mean before sRGB/quantisation.
    params : optional RenderParams selecting one part of a striped frame (multi-GPU); default = whole frame.
    Asynchronous on `stream` (default: torch's current stream)."""
    p = params if params is not None else render_params(img_width, img_height, aa)
    img_ptr = d_image.data_ptr() if hasattr(d_image, "data_ptr") else int(d_image)
    f_ptr = None
    if d_float is not None:
        f_ptr = d_float.data_ptr() if hasattr(d_float, "data_ptr") else int(d_float)
    _check(lib().mirt_render(raw._h, C.byref(p), C.c_void_p(img_ptr), C.c_void_p(f_ptr) if f_ptr else None, _stream_ptr(stream)))


def render_accumulate(d_accum, img_width, img_height, sample_first, sample_count, raw, params=None, stream=None):
    """render_kernel_atomic_aa, draw.cu:49-92: adds samples [sample_first, sample_first + sample_count) of every pixel to the
    float32 accumulation buffer d_accum (num_pixels * 4, zeroed by the caller before the first call)."""
    p = params if params is not None else render_params(img_width, img_height, max(sample_first + sample_count, 2))
    _check(lib().mirt_render_accumulate(raw._h, C.byref(p), C.c_void_p(d_accum.data_ptr()), int(sample_first), int(sample_count), _stream_ptr(stream)))


def finalize(d_image, d_accum, img_width, img_height, total_samples, params=None, stream=None):
    """finalize_kernel, draw.cu:13-47: mean over total_samples, sRGB, 8-bit with rounding."""
    p = params if params is not None else render_params(img_width, img_height, max(total_samples, 2))
    _check(lib().mirt_finalize(C.byref(p), C.c_void_p(d_accum.data_ptr()), int(total_samples), C.c_void_p(d_image.data_ptr()), _stream_ptr(stream)))


def scatter_part(params, d_part, d_frame, stream=None):
    _check(lib().mirt_scatter_part(C.byref(params), C.c_void_p(d_part.data_ptr()), C.c_void_p(d_frame.data_ptr()), _stream_ptr(stream)))


def write_png(path, rgba_u8_host, width, height):
    """Image::save, libpng.cpp:73-107."""
    import numpy as np
    a = np.ascontiguousarray(rgba_u8_host, dtype=np.uint8)
    assert a.size == width * height * 4
    _check(lib().mirt_write_png(os.fsencode(path), a.ctypes.data, width, height))


def probe_math(which, x, device=0):
    import numpy as np
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.zeros_like(x)
    _check(lib().mirt_probe_math(device, which, x.size, x.ctypes.data, out.ctypes.data))
    return out


def probe_xorwow(spp, num_streams, draws, device=0):
    import numpy as np
    out = np.zeros((num_streams, draws), dtype=np.uint32)
    _check(lib().mirt_probe_xorwow(device, spp, num_streams, draws, out.ctypes.data))
    return out
