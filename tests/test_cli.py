"""The user-facing contract of the reference (main.cu:25-94): `raytracer scene.txt` writes the PNG the scene's `png W H name`
line names, prints the phase lines, and exits 1 with the reference's messages for a missing file / a bad line
(parse.cpp:22-25, 218-221).  The binary (cuda_ray_tracer_amd/_build/raytracer, csrc/raytracer_main.cpp) runs as a fresh child
process over libmirt's C ABI."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import scene_path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "cuda_ray_tracer_amd", "_build", "raytracer")


def run(args, cwd):
    return subprocess.run([CLI] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)


def test_missing_scene_file_exits_1_with_the_reference_message(tmp_path):
    r = run([str(tmp_path / "nope.txt")], tmp_path)
    assert r.returncode == 1 and r.stdout.strip() == "Error opening file..."          # parse.cpp:22-25
    r = run([], tmp_path)
    assert r.returncode == 1


def test_bad_scene_line_exits_1_with_the_reference_message(tmp_path):
    bad = tmp_path / "bad.txt"
    bad.write_text("png 8 8 bad.png\nsphere 0 0 -1\n")                                  # a sphere line needs four numbers
    r = run([str(bad)], tmp_path)
    assert r.returncode == 1 and r.stdout.strip() == "One of the lines are not valid."  # parse.cpp:218-221
    assert not (tmp_path / "bad.png").exists()


def test_unknown_option_is_rejected(tmp_path):
    r = run([scene_path("tri"), "--frobnicate"], tmp_path)
    assert r.returncode == 2 and "unknown option" in r.stderr


def test_no_gpu_is_a_loud_failure(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = run([scene_path("tri")], tmp_path)
    assert r.returncode == 1 and "no HIP device" in r.stderr and not (tmp_path / "tri.png").exists()


PHASES = [r"Initialize raw config time: [0-9.e+-]+ seconds", r"LBVH Build time \(N=5\): [0-9.]+ ms",
          r"LBVH Build \(Karas algorithm\) complete\. Total nodes: 9", r"Malloc and transfer to device time: [0-9.e+-]+ seconds",
          r"Render time: [0-9.e+-]+ seconds", r"Transfer to host time: [0-9.e+-]+ seconds", r"hipFree time: [0-9.e+-]+ seconds"]


@pytest.mark.gpu
def test_cli_renders_tri_txt_to_the_png_the_scene_names(tmp_path, oracle_scenes):
    """BASELINE config 1's scene through the CLI: tri.txt at 256x256, aa 0 -> ./tri.png (the scene's `png` line names it),
    the reference's phase lines in order (main.cu:39,61,71,80,93, lbvh_builder.cu:489), and the oracle's bytes."""
    import oracle_lib as ol
    from PIL import Image
    r = run([scene_path("tri"), "--width", "256", "--height", "256", "--spp", "0"], tmp_path)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == len(PHASES), r.stdout
    for line, pat in zip(lines, PHASES):
        assert re.fullmatch(pat, line), (line, pat)
    img = np.array(Image.open(tmp_path / "tri.png"))
    assert img.shape == (256, 256, 4)
    ref = oracle_scenes("tri").render(256, 256, 0, flags=ol.PRODUCT_FLAGS)["u8"]
    assert np.abs(img.astype(np.int32) - ref.astype(np.int32)).max() <= 1
    # SURVEY.md 8c pixel values (x, y)
    assert img[128, 64].tolist() == [188, 138, 0, 255] and img[128, 128].tolist() == [238, 238, 238, 255]
    assert img[192, 128].tolist() == [0, 0, 0, 255] and img[64, 64].tolist() == [0, 0, 0, 0]
    assert int((img[..., 3] > 0).sum()) == 20555


@pytest.mark.gpu
def test_cli_scene_defaults_and_out_override(tmp_path, oracle_scenes):
    """No overrides: the scene's own size and aa (tri.txt: 100x100, aa 0); --out names the file; --traversal 0 is the
    reference's visiting order (same bytes)."""
    import oracle_lib as ol
    from PIL import Image
    r = run([scene_path("tri"), "--out", "x.png", "--traversal", "0"], tmp_path)
    assert r.returncode == 0, r.stderr
    img = np.array(Image.open(tmp_path / "x.png"))
    assert img.shape == (100, 100, 4) and int(img.astype(np.uint64).sum()) == 2561600      # SURVEY.md 8c byte sum
    assert not (tmp_path / "tri.png").exists()


@pytest.mark.gpu
def test_multi_gpu_entry_point_with_one_gpu_gives_the_single_gpu_frame(gpu_scenes):
    """mirt_multi_create / mirt_render_frame_multi with one device (all this box has): the frame equals mirt_render's; the
    stripe partition, gather and re-interleave for N > 1 are covered on CPU (tests/test_tiles_gloo.py, test_abi.py)."""
    import torch
    import cuda_ray_tracer_amd as m
    from cuda_ray_tracer_amd import api
    stl, raw = gpu_scenes("tenthousand")
    w, h, spp = 160, 90, 16
    img = torch.empty(w * h * 4, dtype=torch.uint8, device="cuda")
    m.render(img, w, h, spp, raw)
    torch.cuda.synchronize()
    mg = api.MultiGpu(stl, 1)
    frame, st = mg.render_frame(w, h, spp)
    assert np.array_equal(frame.reshape(-1), img.cpu().numpy())
    assert st["num_gpus"] == 1 and st["render_ms"][0] > 0
    mg.set_option("traversal", 0)                      # the reference's visiting order on every device: same bytes
    frame0, _ = mg.render_frame(w, h, spp)
    assert np.array_equal(frame0, frame)
    with pytest.raises(m.MirtError):
        mg.set_option("traversal", 7)
    mg.close()
    with pytest.raises(m.MirtError):
        api.MultiGpu(stl, 2, devices=[0, 0])
