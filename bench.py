#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X ray tracer.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scene tenthousand] [--width 1920 --height 1080 --spp 16]

One step = one frame of the hot path (trace + resolve, scene and BVH resident in HBM, synthetic-free bundled scene):
BASELINE.json's headline workload, `tenthousand.txt` at 1920x1080, 16 samples per pixel.  With N > 1 (launched by
torch.distributed.run, one process per GPU) the frame is cut into interleaved row stripes, every rank renders its
stripes with a replicated BVH, and the 8-bit framebuffer is gathered to rank 0 over RCCL and re-interleaved there; the
total work per frame is fixed ("strong" scaling).  Consecutive frames are kept in flight on alternating streams (two per
GPU; `--serial` for one): the drain of a frame is a single lane's bounce chain, and the next
frame's workgroups use the CUs it frees.  Rank 0 prints ONE JSON line.

value        = rays of the whole frame / max-over-ranks wall time per frame  (Mrays/s; a ray = one hitNearest call with
               bounce != 0), counted by the kernel's counters variant in an untimed pass.
roofline     = algorithmic bytes of rank 0's trace-kernel launch (64 B per internal-node visit + 16 B per sphere test
               + 48 B per triangle test + 44 B per material fetch, all counted) / its mean duration from HIP events on
               the launch stream, against 8 TB/s HBM.
cpu_baseline = the CPU oracle (oracle/, a port of the reference's algorithm) on one host core over a bounded
               sub-sample of the same workload (every `step`-th pixel in x and y), N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes(st):
    return st["internal_visits"] * 64 + st["sphere_tests"] * 16 + st["tri_tests"] * 48 + st["mat_fetches"] * 44


def measured_traffic(workload):
    """HBM bytes per trace-kernel launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, see
    profiles/*traffic*.json); bench.py cannot run the profiler on itself, so this is the last profiled value or None."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload") == workload:
            best = d.get("hbm_bytes_per_launch")
    return best


def cpu_baseline(scene_file, width, height, spp, step):
    """Times the oracle (test infrastructure, used here only as the reported CPU baseline) on one core."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    import pyscene
    o = oracle_lib.OracleScene(pyscene.parse_file(scene_file), bounds_mode=0)
    t0 = time.perf_counter()
    st, _ = o.render_subsample(width, height, spp, step, flags=0, nthreads=1)
    dt = time.perf_counter() - t0
    o.close()
    return {"value": st["rays"] / dt / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port",
            "sample": f"every {step}th pixel in x and y of the {width}x{height} frame at {spp} spp "
                      f"({st['samples']} samples, {st['rays']} rays, {dt:.1f} s, full nearest-hit shadow rays as in the reference)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", default="tenthousand")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--stripe-rows", type=int, default=4)
    ap.add_argument("--cpu-step", type=int, default=3, help="sub-sampling step of the CPU baseline (0 = skip)")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="consecutive frames overlapped on separate streams (1..4; default 2 on one GPU, 4 on several)")
    ap.add_argument("--serial", action="store_true", help="one frame in flight (no overlap of consecutive frames)")
    ap.add_argument("--share-of", type=int, default=0,
                    help="diagnostic: one process renders only part 0 of an N-way stripe partition (what one rank of N does, without "
                         "the gather); the line then reports that share's rays and time, not a whole job")
    ap.add_argument("--png", default=None, help="write the last frame here (rank 0)")
    ap.add_argument("--headline-only", action="store_true", help="skip the extra configurations (profiling runs)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import cuda_ray_tracer_amd as m
    from cuda_ray_tracer_amd import api

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one process per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    # MIRT_BENCH_REHEARSE=1: several ranks share GPU 0 over gloo -- exercises the N > 1 code path on a one-GPU box
    rehearse = os.environ.get("MIRT_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    scene_file = os.path.join(ROOT, "scenes", args.scene + ".txt")
    W, H, SPP = args.width, args.height, args.spp
    stl = m.parseInput(scene_file)
    raw = m.initRawConfigFromStl(stl, local_rank)       # BVH replicated: every rank builds the identical tree
    build_ms = m.build_lbvh_karas(raw)

    from cuda_ray_tracer_amd.tiles import StripePartition, FrameGatherer
    pworld, prank = (args.share_of, 0) if (args.share_of > 1 and world == 1) else (world, rank)
    stripe_rows = args.stripe_rows if pworld > 1 else H
    partition = StripePartition(W, H, stripe_rows, pworld)
    mine = partition.params(prank, SPP)
    # Frames in flight: consecutive frames go to alternating streams, each with its own part buffer and (on rank 0) its own
    # gather buffers and frame, so the next frame's workgroups fill the CUs the draining frame frees (the drain of a frame
    # is one lane's 16-bounce chain, ~8 ms of latency).
    nfl = args.frames_in_flight if args.frames_in_flight > 0 else 2       # 3 is ~3 % better over 24+ steps, worse over 10
    nfl = 1 if args.serial else max(1, min(4, nfl))
    streams = [torch.cuda.Stream(device=dev) for _ in range(nfl)]
    gatherers = [FrameGatherer(partition, prank, pworld, dev) for _ in range(nfl)]
    parts = [g.new_part_buffer(dev) for g in gatherers]
    part = parts[0]
    frame_no = [0]

    def step():
        i = frame_no[0] % nfl
        frame_no[0] += 1
        with torch.cuda.stream(streams[i]):
            m.render(parts[i], W, H, SPP, raw, params=mine)
            if world > 1:
                return gatherers[i].gather(parts[i])   # RCCL gather over xGMI (<= 1.04 MB per rank at 1080p) + re-interleave on rank 0
        return parts[i]                                # N = 1: the part already is the whole row-major frame

    # untimed counting pass (same rays every frame: the RNG is keyed by pixel and sample index only)
    cparams = partition.params(prank, SPP, counters=True)
    m.render(part, W, H, SPP, raw, params=cparams)
    torch.cuda.synchronize()
    cst = raw.stats()
    cdev = torch.device("cpu") if rehearse else dev
    counts = torch.tensor([cst[k] for k in ("rays", "internal_visits", "sphere_tests", "tri_tests", "mat_fetches", "samples")],
                          dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(counts)
    total_rays = float(counts[0].item())

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    raw.stats()                                            # reset the library's running mean of trace-kernel times
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kst = raw.stats()
    tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if args.png:                      # one more frame (a collective on several GPUs: every rank takes part)
        fr = step()
        torch.cuda.synchronize()
        if rank == 0:
            m.write_png(args.png, fr[: W * H * 4].cpu().numpy(), W, H)

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        mean_kernel_ms = kst["trace_kernel_ms_mean"]        # HIP events around every timed frame's trace kernel, on its launch stream
        my_bytes = algorithmic_bytes(cst)
        achieved = my_bytes / (mean_kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "Mrays/sec, tenthousand.txt 1080p@16spp" if (args.scene, W, H, SPP) == ("tenthousand", 1920, 1080, 16)
                      else f"Mrays/sec, {args.scene}.txt {W}x{H}@{SPP}spp",
            "value": total_rays * args.steps / dt / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "bundled scene file scenes/%s.txt (the reference's own input); no synthetic substitution" % args.scene,
            "config": {"workload": f"{args.scene}.txt {W}x{H} {SPP}spp", "rays_per_frame": int(total_rays),
                       "samples_per_frame": int(counts[5].item()), "parallelism": f"image stripes x{world}" if world > 1 else (f"DIAGNOSTIC: part 0 of {pworld} stripe sets on one GPU" if pworld > 1 else "single GPU"), "frames_in_flight": nfl,
                       "stripe_rows": stripe_rows, "lbvh_build_ms": build_ms},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": measured_traffic(f"{args.scene}.txt {W}x{H} {SPP}spp") if world == 1 else None, "kernel": "trace_kernel", "kernel_ms": mean_kernel_ms,
                         "algorithmic_bytes_per_launch": int(my_bytes),
                         "per_ray": {"internal_visits": cst["internal_visits"] / max(cst["rays"], 1),
                                     "sphere_tests": cst["sphere_tests"] / max(cst["rays"], 1),
                                     "tri_tests": cst["tri_tests"] / max(cst["rays"], 1)}},
        }
        if world == 1 and args.cpu_step > 0:
            out["cpu_baseline"] = cpu_baseline(scene_file, W, H, SPP, args.cpu_step)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)

    raw.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
