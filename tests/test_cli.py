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


NODE = (r"  Node %d: num_primitives_in_leaf=%d, primitive_offset=%d, left_child_offset=\d+, right_child_offset=\d+, visited_atomic_counter=%d, "
        r"bbox=\(min: -?[0-9.]+, -?[0-9.]+, -?[0-9.]+, max: -?[0-9.]+, -?[0-9.]+, -?[0-9.]+\)")
# main.cu:39,61,71,80,93 and lbvh_builder.cu:489-520 (tri.txt has 5 primitives: the "Node Info" dump of N <= 16 is printed)
PHASES = ([r"Initialize raw config time: [0-9.e+-]+ seconds", r"LBVH Build time \(N=5\): [0-9.]+ ms",
           r"LBVH Build \(Karas algorithm\) complete\. Total nodes: 9", r"Node Info:"]
          + [NODE % (i, 0, 0, 2) for i in range(4)] + [NODE % (4 + j, 1, j, 0) for j in range(5)]
          + [r"Malloc and transfer to device time: [0-9.e+-]+ seconds",
             r"Render time: [0-9.e+-]+ seconds", r"Transfer to host time: [0-9.e+-]+ seconds", r"cudaFree time: [0-9.e+-]+ seconds"])


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
def test_cli_node_dump_of_the_shipped_tree_is_the_surveys(tmp_path):
    """--bounds-as-shipped builds the tree of the shipped reference (every Morton code 0, parse.cpp:28); its "Node Info" dump
    (lbvh_builder.cu:496-520, the reference's only known-answer hook) must be the nine nodes SURVEY.md 8c recorded from the
    reference's own code: node0{L=3,R=8} node1{L=4,R=5} node2{L=6,R=7} node3{L=1,R=2}, root box (-0.8,-0.7,-1.3)..(0.9,0.6,-0.8),
    leaf 7 (the first triangle) (-0.7,-0.6,-1.2)..(0.8,0.5,-0.9).  With --spp 4 the many-samples kernel announces itself as
    draw.cu:232 does."""
    r = run([scene_path("tri"), "--bounds-as-shipped", "--width", "64", "--height", "64", "--spp", "4"], tmp_path)
    assert r.returncode == 0, r.stderr
    out = r.stdout.splitlines()
    i = out.index("Node Info:")
    nodes = out[i + 1:i + 10]
    kids = {0: (3, 8), 1: (4, 5), 2: (6, 7), 3: (1, 2)}
    for k, (l, rr) in kids.items():
        assert f"Node {k}: num_primitives_in_leaf=0, primitive_offset=0, left_child_offset={l}, right_child_offset={rr}, visited_atomic_counter=2" in nodes[k], nodes[k]
    assert nodes[0].endswith("bbox=(min: -0.80, -0.70, -1.30, max: 0.90, 0.60, -0.80)"), nodes[0]
    assert "Node 7: num_primitives_in_leaf=1, primitive_offset=3," in nodes[7] and nodes[7].endswith("bbox=(min: -0.70, -0.60, -1.20, max: 0.80, 0.50, -0.90)"), nodes[7]
    assert "[DEBUG Render] Launching AA Kernel. Total Threads: 16384, Grid: 128, Block: 128" in out


@pytest.mark.gpu
def test_cli_renders_several_frames(tmp_path):
    r = run([scene_path("tri"), "--width", "64", "--height", "64", "--spp", "2", "--frames", "3"], tmp_path)
    assert r.returncode == 0, r.stderr
    assert re.search(r"frames: 3 \([0-9.]+ ms per frame\)", r.stdout)


@pytest.mark.gpu
def test_cli_multi_gpu_branch_with_parts_sharing_one_gpu(tmp_path):
    """`raytracer scene.txt --gpus 3 --frames 4` -- the CLI's mirt_multi_* branch (frames in flight, stripe parts, gather, per-frame
    time) -- rehearsed on this one-GPU box with MIRT_MULTI_GATHER=copy (the three parts time-share GPU 0): same PNG bytes as the
    single-GPU run."""
    from PIL import Image
    env = dict(os.environ, MIRT_MULTI_GATHER="copy")
    r = subprocess.run([CLI, scene_path("tri"), "--width", "96", "--height", "70", "--spp", "4", "--gpus", "3", "--frames", "4", "--out", "m.png"],
                       cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    assert re.search(r"GPUs: 3, frames: 4 \([0-9.]+ ms per frame\)", r.stdout), r.stdout
    r1 = run([scene_path("tri"), "--width", "96", "--height", "70", "--spp", "4", "--out", "s.png"], tmp_path)
    assert r1.returncode == 0, r1.stderr
    assert np.array_equal(np.array(Image.open(tmp_path / "m.png")), np.array(Image.open(tmp_path / "s.png")))


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
    # frames in flight behind the ABI: tickets, per-device statistics, a capacity check when the pipeline is empty
    bufs = [np.zeros((h, w, 4), np.uint8) for _ in range(3)]
    mg.set_option("traversal", 1)
    tickets = [mg.submit(w, h, spp, out=b) for b in bufs]
    for t, b in zip(tickets, bufs):
        stt = mg.wait(t)
        assert np.array_equal(b, frame) and stt["frame_ms"] > 0
    with pytest.raises(m.MirtError):
        mg.wait(tickets[0])                            # collected already
    last, stl_, ms = mg.render_frames(w, h, spp, nframes=5, in_flight=2)
    assert np.array_equal(last, frame) and ms > 0
    assert mg.stats(0)["overflow_events"] == 0
    mg.close()
    with pytest.raises(m.MirtError):
        api.MultiGpu(stl, 2, devices=[0, 0])


@pytest.mark.gpu
@pytest.mark.parametrize("nparts", [2, 4])
def test_multi_gpu_rehearsal_parts_sharing_one_gpu_give_the_single_gpu_frame(nparts, gpu_scenes, monkeypatch):
    """MIRT_MULTI_GATHER=copy: the N > 1 code path of mirt_multi_* end to end on this one-GPU box -- N scenes built concurrently
    (one host thread each), N stripe sets rendered on their own streams, gathered to part 0 with peer copies on the
    communication streams, re-interleaved, frames in flight -- with every part time-sharing GPU 0.  A rehearsal: RCCL's
    send/recv branch itself has never run (no multi-GPU node was available); nothing here says anything about scaling."""
    import torch
    import cuda_ray_tracer_amd as m
    from cuda_ray_tracer_amd import api
    stl, raw = gpu_scenes("tenthousand")
    w, h, spp = 200, 111, 8                            # 111 rows: ragged last stripe, parts of unequal size
    img = torch.empty(w * h * 4, dtype=torch.uint8, device="cuda")
    m.render(img, w, h, spp, raw)
    torch.cuda.synchronize()
    want = img.cpu().numpy().reshape(h, w, 4)
    monkeypatch.setenv("MIRT_MULTI_GATHER", "copy")
    mg = api.MultiGpu(stl, nparts, devices=[0] * nparts)
    frame, st = mg.render_frame(w, h, spp)
    assert np.array_equal(frame, want) and st["num_gpus"] == nparts
    last, st2, ms = mg.render_frames(w, h, spp, nframes=6, in_flight=3)
    assert np.array_equal(last, want)
    bufs = [np.zeros((h, w, 4), np.uint8) for _ in range(4)]
    tickets = [mg.submit(w, h, spp, stripe_rows=1 + i, out=b) for i, b in enumerate(bufs)]      # a different partition per frame
    with pytest.raises(m.MirtError):
        mg.submit(w, h, spp)                           # MIRT_MULTI_MAX_IN_FLIGHT frames are in flight
    for t, b in zip(tickets, bufs):
        mg.wait(t)
        assert np.array_equal(b, want)
    for part in range(nparts):
        assert mg.stats(part)["overflow_events"] == 0
    mg.close()
