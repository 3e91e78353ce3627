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
    pix = np.concatenate([o.render(W, H, SPP, tile=(0, r, W, 1))["u8"].reshape(-1) for r in rows])
    buf[: pix.size] = torch.from_numpy(pix)
    frame = g.gather(buf)
    if rank == 0:
        np.save(out_path, frame.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_gloo_frame_gather_equals_whole_frame(world, tmp_path):
    """7 stripes of 4 rows (the last one 3 rows) over 2 or 3 ranks: ragged parts, padded buffers."""
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
