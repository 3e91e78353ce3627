"""The N > 1 path on CPU: two gloo ranks partition a frame into interleaved stripes, each produces its part buffer,
the framebuffer is gathered to rank 0 and re-interleaved with the SAME code bench.py runs on GPUs
(cuda_ray_tracer_amd.tiles).  The per-rank pixels come from the CPU oracle here (test-only stand-in for the kernel);
the result must equal the oracle's whole-frame render bit for bit (tile-split invariance)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SPP, ROWS = 40, 27, 16, 4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib
    import pyscene
    from cuda_ray_tracer_amd.tiles import StripePartition, FrameGatherer
    part = StripePartition(W, H, ROWS, world)
    g = FrameGatherer(part, rank, world, torch.device("cpu"))
    buf = g.new_part_buffer(torch.device("cpu"))
    o = oracle_lib.OracleScene(pyscene.parse_file(os.path.join(ROOT, "scenes", "tenthousand.txt")), bounds_mode=0)
    rows = part.rows(rank)
    if rows:      # (with 8 ranks and 7 stripes the last rank owns nothing)
        pix = np.concatenate([o.render(W, H, SPP, tile=(0, r, W, 1))["u8"].reshape(-1) for r in rows])
        buf[: pix.size] = torch.from_numpy(pix)
    frame = g.gather(buf)
    if rank == 0:
        np.save(out_path, frame.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3, 8])
def test_gloo_frame_gather_equals_whole_frame(world, tmp_path):
    """7 stripes of 4 rows (the last one 3 rows) over 2, 3 or 8 ranks (the driver's scale run uses 8; here one rank owns
    nothing): ragged parts, padded buffers."""
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out).reshape(H, W, 4)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    import pyscene
    o = oracle_lib.OracleScene(pyscene.parse_file(os.path.join(ROOT, "scenes", "tenthousand.txt")), bounds_mode=0)
    want = o.render(W, H, SPP, nthreads=4)["u8"]
    assert np.array_equal(got, want)


def test_uneven_parts_are_padded():
    sys.path.insert(0, ROOT)
    from cuda_ray_tracer_amd.tiles import StripePartition, FrameGatherer
    part = StripePartition(8, 10, 4, 2)          # stripes: rows 0-3 (part 0), 4-7 (part 1), 8-9 (part 0)
    assert part.rows(0) == [0, 1, 2, 3, 8, 9] and part.rows(1) == [4, 5, 6, 7]
    g = FrameGatherer(part, 0, 1, torch.device("cpu"))
    assert g.max_bytes == 6 * 8 * 4


def _gpu_worker(rank, world, port, out_path, w, h, spp, rows):
    """Rehearsal of bench.py's N > 1 path on a one-GPU box: every rank renders its stripes through libmirt on GPU 0, the
    framebuffer gather runs over gloo (RCCL refuses several ranks on one device; tiles.FrameGatherer stages through the host
    for gloo)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cuda_ray_tracer_amd as m
    from cuda_ray_tracer_amd.tiles import StripePartition, FrameGatherer
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    stl = m.parseInput(os.path.join(ROOT, "scenes", "tenthousand.txt"))
    raw = m.initRawConfigFromStl(stl, 0)
    m.build_lbvh_karas(raw)
    part = StripePartition(w, h, rows, world)
    g = FrameGatherer(part, rank, world, dev)
    buf = g.new_part_buffer(dev)
    m.render(buf, w, h, spp, raw, params=part.params(rank, spp))
    torch.cuda.synchronize()
    frame = g.gather(buf)
    if rank == 0:
        np.save(out_path, frame.cpu().numpy())
    dist.barrier()
    raw.close()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_kernel_plus_gather_over_four_ranks_equals_the_single_rank_frame(tmp_path):
    """Kernel + gather end to end (the gloo test above feeds oracle pixels): four ranks share this box's one GPU."""
    sys.path.insert(0, ROOT)
    import cuda_ray_tracer_amd as m
    w, h, spp, rows, world = 200, 121, 16, 4, 4
    out = str(tmp_path / "frame.npy")
    mp.spawn(_gpu_worker, args=(world, _free_port(), out, w, h, spp, rows), nprocs=world, join=True)
    got = np.load(out)
    stl = m.parseInput(os.path.join(ROOT, "scenes", "tenthousand.txt"))
    raw = m.initRawConfigFromStl(stl, 0)
    m.build_lbvh_karas(raw)
    img = torch.empty(w * h * 4, dtype=torch.uint8, device="cuda")
    m.render(img, w, h, spp, raw)
    torch.cuda.synchronize()
    raw.close()
    assert np.array_equal(got, img.cpu().numpy())
