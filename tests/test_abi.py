"""The C-ABI library loads and exports every symbol include/mirt.h declares; host-only entry points behave.
No compute calls are made here (no GPU in this container)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import cuda_ray_tracer_amd as m
from cuda_ray_tracer_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "mirt.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mirt_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = m.lib()
    syms = declared_symbols()
    assert len(syms) >= 19
    for s in syms:
        assert hasattr(L, s), s
    assert sorted(api.EXPORTS) == syms
    assert L.mirt_version() == 3


def test_exported_symbols_are_unmangled_c():
    out = subprocess.check_output(["nm", "-D", "--defined-only", api.LIB_PATH]).decode()
    names = {line.split()[-1] for line in out.splitlines() if line.strip()}
    for s in declared_symbols():
        assert s in names, s


def test_pod_struct_sizes_match_the_reference_layout():
    assert C.sizeof(api.SceneDesc) == 4 * 4 + 2 * 4 + 4 * 12 + 4 + 3 * 4 + 6 * 4 + 6 * 8
    assert C.sizeof(api.RenderParams) == 28
    assert C.sizeof(api.TreeNode) == 40


@pytest.mark.parametrize("w,h,rows,parts", [(1920, 1080, 4, 8), (200, 121, 8, 2), (50, 7, 5, 3), (17, 9, 4, 8), (64, 36, 36, 1)])
def test_stripe_partition_covers_the_frame_exactly_once(w, h, rows, parts):
    from cuda_ray_tracer_amd.tiles import StripePartition
    sp = StripePartition(w, h, rows, parts)
    seen = np.zeros(h, np.int32)
    for p in range(parts):
        n = api.num_pixels(api.render_params(w, h, 16, rows, parts, p))
        assert n == sp.num_pixels(p)
        for r in sp.rows(p):
            seen[r] += 1
    assert np.all(seen == 1)


def test_bad_render_params_are_rejected():
    assert m.lib().mirt_render_num_pixels(C.byref(api.render_params(10, 10, 1, 4, 2, 2))) == -1
    assert m.lib().mirt_render_num_pixels(C.byref(api.render_params(0, 10, 1, 4, 1, 0))) == -1


def test_png_writer_roundtrip(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53, 4), dtype=np.uint8)
    path = str(tmp_path / "x.png")
    m.write_png(path, img, 53, 37)
    back = np.array(Image.open(path))
    assert back.shape == (37, 53, 4) and np.array_equal(back, img)


def test_product_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    stl = m.parseText("png 4 4 a.png\nsphere 0 0 -1 1\n")
    with pytest.raises(m.MirtError) as e:
        m.initRawConfigFromStl(stl, 0)
    assert e.value.status == 5 and "no CPU path" in e.value.message


def test_product_does_not_reference_the_oracle():
    """The product tree must not include, link or load anything under oracle/."""
    pkg = os.path.join(ROOT, "cuda_ray_tracer_amd")
    for dirpath, _, files in os.walk(pkg):
        if "_build" in dirpath or "__pycache__" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.lower(), os.path.join(dirpath, f)
    needed = subprocess.check_output(["objdump", "-p", api.LIB_PATH]).decode()
    assert "liboracle" not in needed


@pytest.mark.parametrize("nparts,width,height,rows", [(2, 7, 10, 4), (8, 5, 37, 4), (8, 3, 20, 1), (3, 4, 9, 2), (8, 6, 5, 4)])
def test_part_pixel_arithmetic_tiles_the_frame_like_tiles_py(nparts, width, height, rows):
    """mirt_render_num_pixels / mirt_part_pixel_xy -- the host side of the partition arithmetic mirt_multi_* and the kernels use
    (scatter, sample -> pixel) -- against cuda_ray_tracer_amd/tiles.py's StripePartition for N = 2, 3, 8: every part's local
    pixels, in order, are its stripes' rows; together they cover the frame exactly once (ragged last stripe, parts with no
    stripe at all)."""
    from cuda_ray_tracer_amd import api
    from cuda_ray_tracer_amd.tiles import StripePartition
    sp = StripePartition(width, height, rows, nparts)
    seen = set()
    for part in range(nparts):
        p = api.render_params(width, height, 1, rows, nparts, part)
        n = api.num_pixels(p)
        assert n == sp.num_pixels(part)
        want = [(x, y) for y in sp.rows(part) for x in range(width)]
        got = [api.part_pixel_xy(p, i) for i in range(n)]
        assert got == want
        seen.update(got)
        with pytest.raises(api.MirtError):
            api.part_pixel_xy(p, n)
    assert seen == {(x, y) for y in range(height) for x in range(width)}
