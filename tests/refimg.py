"""Statistics against the three renders the reference ships (docs/{tenthousand,spiral,redchair}.png, copied as DATA
to tests/golden/ref_docs/): the only outputs of the reference that exist.  They were made on a CUDA machine with
cuRAND and CUDA's libm, so agreement can only be statistical: per-channel image means and PSNR at several scales
(the box-filtered images average the Monte-Carlo noise of both sides away, so a structural error -- a missing GI
term, a wrong refraction, exposure on/off -- shows up as a PSNR of 20-28 dB where a faithful render gives 38-44 dB).
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "golden", "ref_docs")
REF_SIZE = 800

# Per-channel mean of the 8-bit image, in 8-bit units ("per-channel mean within 1.0/255")
MEAN_TOL = 1.0
# PSNR floors in dB: (full comparison size, 100x100 box-filtered).  Measured with the oracle at 400x400 x 8 spp:
# tenthousand 27.3 / 38.2, spiral 31.7 / 40.5, redchair (exposure off) 29.9 / 38.4; the negative controls below
# reach at most 26.2 / 27.9.
PSNR_FLOOR = {"tenthousand": (25.0, 35.0), "spiral": (29.0, 37.0), "redchair": (27.5, 35.0)}
# At the reference's native 800x800 with each scene's own `aa` (oracle: 34.8 / 51.3, 38.7 / 52.0, 35.6 / 50.7 dB) and, with the
# truncating quantiser the shipped images were written with, per-channel means within 0.05/255 (oracle: <= 0.046).
PSNR_FLOOR_NATIVE = {"tenthousand": (32.0, 46.0), "spiral": (35.0, 46.0), "redchair": (32.0, 46.0)}
MEAN_TOL_TRUNCATED_NATIVE = 0.15


def load_reference(name):
    """docs/<name>.png as float64 [800, 800, 4] (RGBA, 0..255)."""
    from PIL import Image
    im = np.asarray(Image.open(os.path.join(REF_DIR, name + ".png")).convert("RGBA")).astype(np.float64)
    assert im.shape == (REF_SIZE, REF_SIZE, 4)
    return im


def box(a, size):
    k = a.shape[0] // size
    assert k * size == a.shape[0] and a.shape[0] == a.shape[1]
    return a.reshape(size, k, size, k, a.shape[-1]).mean(axis=(1, 3))


def psnr(a, b):
    mse = float(np.mean((a - b) ** 2))
    return float("inf") if mse == 0 else 10.0 * np.log10(255.0 ** 2 / mse)


def compare(name, u8):
    """u8: [S, S, 4] render of scene `name` (S divides 800).  Returns dict(dmean[4], psnr_full, psnr_100)."""
    u = np.asarray(u8).astype(np.float64)
    s = u.shape[0]
    ref = box(load_reference(name), s)
    return dict(dmean=(u.mean(axis=(0, 1)) - ref.mean(axis=(0, 1))),
                psnr_full=psnr(u[..., :3], ref[..., :3]),
                psnr_100=psnr(box(u, 100)[..., :3], box(ref, 100)[..., :3]))


def srgb_truncated(f32):
    """Linear float RGBA [.., 4] -> 8-bit with the TRUNCATING quantiser of render_kernel (draw.cu:129-132) instead of the
    rounding one of render_kernel_warp_aa (draw.cu:9-11).  The shipped images sit 0.5 x coverage below a rounded render
    and within 0.05/255 of a truncated one: they were written by a revision that truncated."""
    f = np.asarray(f32).astype(np.float64)
    l = f[..., :3]
    s = np.where(l < 0.0031308, 12.92 * l, 1.055 * np.power(np.maximum(l, 1e-30), 1 / 2.4) - 0.055)
    rgb = np.floor(np.clip(s, 0.0, 1.0) * 255.0)
    a = np.floor(np.clip(f[..., 3:], 0.0, 1.0) * 255.0)
    return np.concatenate([rgb, a], axis=-1).astype(np.uint8)


def passes(name, c):
    lo_full, lo_100 = PSNR_FLOOR[name]
    return bool(np.all(np.abs(c["dmean"]) <= MEAN_TOL)) and c["psnr_full"] >= lo_full and c["psnr_100"] >= lo_100


def strip_expose(text):
    """The scene text without its `expose` line(s): docs/redchair.png shows no exposure."""
    return "\n".join(l for l in text.split("\n") if not l.strip().startswith("expose")) + "\n"
