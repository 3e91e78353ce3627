#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X ray tracer.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scene tenthousand] [--width 1920 --height 1080 --spp 16]

One step = one frame of the hot path (trace + resolve, scene and BVH resident in HBM, the reference's own scene file):
BASELINE.json's headline workload, `tenthousand.txt` at 1920x1080, 16 samples per pixel.  With N > 1 (launched by
torch.distributed.run, one process per GPU) the frame is cut into interleaved row stripes, every rank renders its
stripes with a replicated BVH, and the 8-bit framebuffer is gathered to rank 0 over RCCL and re-interleaved there; the
total work per frame is fixed ("strong" scaling).  Consecutive frames are kept in flight on alternating streams (one to
three per GPU, whichever an untimed calibration finds fastest; `--serial` for one): the drain of a frame is a single lane's bounce chain, and the next frame's workgroups use the CUs
it frees.  Rank 0 prints ONE JSON line.

value        = rays of the whole frame / max-over-ranks wall time per frame  (Mrays/s; a ray = one hitNearest call with
               bounce != 0), counted by the kernel's counters variant in an untimed pass.
roofline     = the bytes of the records rank 0's trace-kernel launch requests -- internal-node visits x the record size of the walk
               in use (32 B quantised records on a sphere-only scene, 64 B otherwise) + 16 B per sphere test + 48 B per triangle
               test + 44 B per material fetch, all counted -- / the launch's mean duration from HIP events on its stream (one event
               pair per launch; a frame of several slabs is several launches), against 8 TB/s HBM: `achieved`, `frac`.
               `frac_contract` prices every node visit at SURVEY.md 8d's 64 B whatever the record (round 1-2's `frac`).  The BVH
               of the bundled scenes lives in L1/L2, so both are delivered record bandwidth, not DRAM utilisation; what bounds the
               kernel is in `limiter`: the busiest unit of the committed rocprofv3 --pmc passes of THIS build
               (profiles/*pmc_trace_kernel*.json carry a hash of csrc/; a summary of another build is marked stale and not used
               for `traffic`).
configs      = (N = 1) the other BASELINE configurations on this GPU, a few frames each: spiral 1080p16, redchair 4K64,
               the synthetic 1 M spheres + 1 M triangles scene at 4K x 256 spp (whole frame on one GPU) -- ms/frame,
               Mrays/s, node visits per ray, algorithmic roofline fraction.
cpu_baseline = the CPU oracle (oracle/, a port of the reference's algorithm) on one host core over a bounded
               sub-sample of the same workload (every `step`-th pixel in x and y), N = 1 only.
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
HEADLINE = ("tenthousand", 1920, 1080, 16)


def algorithmic_bytes(st):
    return st["internal_visits"] * 64 + st["sphere_tests"] * 16 + st["tri_tests"] * 48 + st["mat_fetches"] * 44


def source_hash():
    from cuda_ray_tracer_amd import build as B
    return B.source_hash()


def committed_pmc(workload):
    """The committed counter summary for this workload (tools/pmc_profile.py -> profiles/*pmc_trace_kernel*.json; bench.py cannot
    run the profiler on itself): the newest one taken on THIS build (its csrc_sha16 is the tree's), else the newest one of any
    build, which the caller then reports as stale.  Returns (file name, summary, stale)."""
    sha = source_hash()
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_trace_kernel*.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload") != workload:
            continue
        fresh = d.get("csrc_sha16") == sha
        if best is None or fresh or best[2]:
            best = (os.path.basename(f), d, not fresh)
    return best


def committed_limiter(workload):
    """What bounds the kernel on `workload`, from its committed counter passes (None when there are none): per launch of the
    trace kernel (one slab of a frame that takes several).  `bound` names the busiest unit; a summary taken on another build is
    kept for orientation only and says so (`stale`)."""
    got = committed_pmc(workload)
    if not got:
        return None
    name, d, stale = got
    dv = d.get("derived", {})
    # (per-launch duration from the launch's own cycle counter: a frame of several slabs is several launches)
    launch_ms = dv.get("gui_active_ms")
    out = {"pmc_source": "profiles/" + name, "pmc_workload": d.get("workload"), "pmc_csrc_sha16": d.get("csrc_sha16"), "pmc_git_head": d.get("git_head"),
           "stale": stale, "pmc_launch_ms": launch_ms}
    for k_out, k_in in (("valu_issue_frac", "valu_issue_frac"), ("ta_busy", "ta_busy"), ("l1_requests_per_launch", "l1_requests"), ("ta_floor_ms", "ta_floor_ms"),
                        ("l1_hit_rate", "l1_hit_rate"), ("l2_hit_rate", "l2_hit_rate"), ("active_lanes_per_valu_inst", "active_lanes"),
                        ("wave_wait_frac", "wave_wait_frac"), ("beyond_l2_read_bytes_low", "hbm_read_bytes_low"),
                        ("beyond_l2_read_bytes_high", "hbm_read_bytes_high"), ("beyond_l2_write_bytes", "hbm_write_bytes")):
        if k_in in dv:
            out[k_out] = dv[k_in]
    if launch_ms:
        w = dv.get("hbm_write_bytes", 0.0)
        # read bytes by request size (tools/pmc_profile.py, group rdreq): a 64-byte tally stands for 64 or 128 bytes -> a bracket;
        # the fraction claimed is the LOW one.  Summaries from before round 3 only have FETCH_SIZE (x 2: the HIGH reading).
        if "hbm_read_bytes_low" in dv:
            out["measured_hbm_frac"] = (dv["hbm_read_bytes_low"] + w) / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            out["measured_hbm_frac_high"] = (dv["hbm_read_bytes_high"] + w) / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            out["beyond_l2_bytes_per_launch"] = dv["hbm_read_bytes_low"] + w
        elif "hbm_bytes_per_launch" in dv:
            out["measured_hbm_frac_high"] = dv["hbm_bytes_per_launch"] / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            out["beyond_l2_bytes_per_launch"] = dv["hbm_bytes_per_launch"]
        if "ta_floor_ms" in dv:
            out["frac_of_ta_floor"] = dv["ta_floor_ms"] / launch_ms
    units = {"valu_issue": out.get("valu_issue_frac"), "address_units": out.get("ta_busy"), "hbm": out.get("measured_hbm_frac")}
    units = {k: v for k, v in units.items() if v is not None}
    if units:
        out["bound"] = max(units, key=units.get)
        out["frac"] = units[out["bound"]]
    return out


def record_bytes(st, node_bytes):
    """Bytes of the records a launch requests: what its own record layout reads per counted event."""
    return st["internal_visits"] * node_bytes + st["sphere_tests"] * 16 + st["tri_tests"] * 48 + st["mat_fetches"] * 44


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
            "sample": f"every pixel whose x and y are multiples of {step} of the {width}x{height} frame at {spp} spp "
                      f"({st['samples']} samples, {st['rays']} rays, {dt:.1f} s; the reference's own traversal: left-first order, "
                      f"full nearest-hit shadow rays to every light)"}


def time_config(m, api, torch, raw, w, h, spp, steps, label, options=None, pmc_workload=None):
    """A few serial frames of one more configuration on this GPU (N = 1): counters in an untimed pass, then `steps` timed frames."""
    old = {}
    for k, v in (options or {}).items():
        old[k] = raw.get_option(k)
        raw.set_option(k, v)
    img = torch.empty(w * h * 4, dtype=torch.uint8, device="cuda")
    m.render(img, w, h, spp, raw, params=api.render_params(w, h, spp, counters=True))
    torch.cuda.synchronize()
    st = raw.stats()
    if w * h * max(spp, 1) <= (1 << raw.get_option("slab_log2")):
        m.render(img, w, h, spp, raw)                  # warm-up: a one-slab frame runs with the chunk order measured on the previous one
        torch.cuda.synchronize()
    raw.stats()
    t0 = time.perf_counter()
    for _ in range(steps):
        m.render(img, w, h, spp, raw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ks = raw.stats()
    for k, v in old.items():
        raw.set_option(k, v)
    del img
    ab = algorithmic_bytes(st)
    kms = ks["trace_kernel_ms_mean"]                     # per frame: the sum over the frame's trace launches
    launches = max(ks.get("trace_launches", 1), 1)
    rb = record_bytes(st, ks.get("node_record_bytes", 64))
    out = {"workload": label, "ms_per_frame": dt * 1e3, "trace_kernel_ms": kms, "trace_launches_per_frame": launches,
           "trace_kernel_ms_per_launch": kms / launches, "Mrays_per_s": st["rays"] / dt / 1e6, "rays_per_frame": st["rays"],
           "rays_traversed_per_frame": st.get("rays_traversed"),
           "node_visits_per_ray": st["internal_visits"] / max(st["rays"], 1), "node_record_bytes": ks.get("node_record_bytes", 64),
           "leaf_tests_per_ray": (st["sphere_tests"] + st["tri_tests"]) / max(st["rays"], 1),
           "requested_record_GBps": rb / (kms * 1e-3) / 1e9, "frac": rb / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "frac_contract": ab / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, "frames_timed": steps,
           "options": options or {}}
    lim = committed_limiter(pmc_workload or label) if not options else None      # (counter passes exist for the default options only)
    if lim:
        out["limiter"] = lim
    if out["frac"] > 1.0:
        out["frac_note"] = ("above 1: the records this launch requests are answered by the L1 / L2s (a small tree, the lanes of a wave on samples of one "
                            "kind), not by HBM -- delivered record bandwidth; limiter.bound names what the kernel is bound by")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", default=HEADLINE[0])
    ap.add_argument("--width", type=int, default=HEADLINE[1])
    ap.add_argument("--height", type=int, default=HEADLINE[2])
    ap.add_argument("--spp", type=int, default=HEADLINE[3])
    ap.add_argument("--stripe-rows", type=int, default=4)
    ap.add_argument("--cpu-step", type=int, default=3, help="sub-sampling step of the CPU baseline (0 = skip)")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="consecutive frames overlapped on separate streams (1..4; default 2)")
    ap.add_argument("--serial", action="store_true", help="one frame in flight (no overlap of consecutive frames)")
    ap.add_argument("--share-of", type=int, default=0,
                    help="diagnostic: one process renders only part 0 of an N-way stripe partition (what one rank of N does, without "
                         "the gather); the line then reports that share's rays and time, not a whole job")
    ap.add_argument("--png", default=None, help="write the last frame here (rank 0)")
    ap.add_argument("--headline-only", action="store_true", help="skip the extra configurations (profiling runs)")
    ap.add_argument("--config5-spp", type=int, default=256, help="samples per pixel of the synthetic 2 M-primitive configuration (BASELINE: 256)")
    args = ap.parse_args()

    # (the pool's host driver only supports dmabuf IPC: without this RCCL fails with `hipIpcGetMemHandle: invalid argument`)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
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
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        backend = dist.get_backend()

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
    auto_fif = args.frames_in_flight <= 0 and not args.serial      # default: whichever of 1 / 2 frames in flight is faster here (calibrated below)
    nfl = args.frames_in_flight if args.frames_in_flight > 0 else 3       # (auto: 1, 2 and 3 are calibrated below)
    nfl = 1 if args.serial else max(1, min(4, nfl))
    streams = [torch.cuda.Stream(device=dev) for _ in range(nfl)]
    gatherers = [FrameGatherer(partition, prank, pworld, dev) for _ in range(nfl)]
    parts = [g.new_part_buffer(dev) for g in gatherers]
    part = parts[0]
    frame_no = [0]
    # per-frame device times of this rank: its part's render and (N > 1) the gather + re-interleave that follows it
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(nfl)]
    render_ms, gather_ms = [], []
    pending = [None] * nfl

    def collect(i):
        if pending[i]:
            e0, e1, e2 = ev[i]
            e2.synchronize()
            render_ms.append(e0.elapsed_time(e1))
            gather_ms.append(e1.elapsed_time(e2))
            pending[i] = None

    active = [nfl]      # frames in flight in use (<= nfl)

    def step(timed=False):
        i = frame_no[0] % active[0]
        frame_no[0] += 1
        if timed:
            collect(i)
        with torch.cuda.stream(streams[i]):
            e0, e1, e2 = ev[i]
            if timed:
                e0.record()
            m.render(parts[i], W, H, SPP, raw, params=mine)
            if timed:
                e1.record()
            out = parts[i]                                 # N = 1: the part already is the whole row-major frame
            if world > 1:
                out = gatherers[i].gather(parts[i])        # RCCL gather over xGMI (<= 1.04 MB per rank at 1080p) + re-interleave on rank 0
            if timed:
                e2.record()
                pending[i] = True
        return out

    # untimed counting pass (same rays every frame: the RNG is keyed by pixel and sample index only)
    cparams = partition.params(prank, SPP, counters=True)
    m.render(part, W, H, SPP, raw, params=cparams)
    torch.cuda.synchronize()
    cst = raw.stats()
    cdev = torch.device("cpu") if rehearse else dev
    counts = torch.tensor([cst[k] for k in ("rays", "internal_visits", "sphere_tests", "tri_tests", "mat_fetches", "samples", "rays_traversed")],
                          dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(counts)
    total_rays = float(counts[0].item())

    for _ in range(args.warmup):
        step()
    fif_ms = None
    serial_kernel_ms = [None]
    if auto_fif:
        # untimed calibration: a whole frame on this kernel is faster alone than overlapped with the next one (round 3: 21.9 vs
        # 23.3 ms), a stripe share of it is not (a 1/8 share: 3.21 / 2.93 / 2.89 ms at 1 / 2 / 3 in flight) -- take whichever is
        # fastest on this box, every rank the same
        fif_ms = {}
        for k in (1, 2, 3):
            active[0] = k
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            tc = time.perf_counter()
            for _ in range(6):
                step()
            torch.cuda.synchronize()
            t = torch.tensor([time.perf_counter() - tc], dtype=torch.float64, device=torch.device("cpu") if rehearse else dev)
            if world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            fif_ms[k] = float(t.item()) / 6 * 1e3
            if k == 1:
                serial_kernel_ms[0] = raw.stats()["trace_kernel_ms_mean"]      # this rank's trace kernel with the GPU to itself
            else:
                raw.stats()
        active[0] = min(fif_ms, key=fif_ms.get)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    raw.stats()                                            # reset the library's running mean of trace-kernel times
    for _ in range(args.steps):
        step(timed=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for i in range(nfl):
        collect(i)
    kst = raw.stats()
    tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    # every rank's mean render / gather time (diagnostics of the N > 1 line)
    mine_ms = torch.tensor([sum(render_ms) / max(len(render_ms), 1), sum(gather_ms) / max(len(gather_ms), 1), kst["trace_kernel_ms_mean"]],
                           dtype=torch.float64, device=cdev)
    all_ms = [mine_ms]
    if world > 1:
        all_ms = [torch.zeros_like(mine_ms) for _ in range(world)]
        dist.all_gather(all_ms, mine_ms)

    if args.png:                      # one more frame (a collective on several GPUs: every rank takes part)
        fr = step()
        torch.cuda.synchronize()
        if rank == 0:
            m.write_png(args.png, fr[: W * H * 4].cpu().numpy(), W, H)

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        mean_kernel_ms = kst["trace_kernel_ms_mean"]        # HIP events around every trace launch of the timed frames, on their launch stream
        launches = max(kst.get("trace_launches", 1), 1)
        node_bytes = kst.get("node_record_bytes", 64)
        my_bytes = algorithmic_bytes(cst)
        req_bytes = record_bytes(cst, node_bytes)
        achieved = req_bytes / (mean_kernel_ms * 1e-3) / 1e9
        workload = f"{args.scene}.txt {W}x{H} {SPP}spp"
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": None, "kernel": "trace_kernel", "kernel_ms": mean_kernel_ms / launches, "launches_per_frame": launches,
                "requested_record_bytes_per_launch": int(req_bytes / launches), "node_record_bytes": node_bytes,
                "frac_contract": my_bytes / (mean_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch_contract": int(my_bytes / launches),
                "per_ray": {"internal_visits": cst["internal_visits"] / max(cst["rays"], 1),
                            "sphere_tests": cst["sphere_tests"] / max(cst["rays"], 1),
                            "tri_tests": cst["tri_tests"] / max(cst["rays"], 1)},
                "note": "achieved / frac: bytes of the records the launch requests (node visits x node_record_bytes + 16 B per sphere test + 48 B per "
                        "triangle test + 44 B per material fetch, all counted) / HIP-event kernel time / 8 TB/s; frac_contract prices a node visit "
                        "at SURVEY.md 8d's 64 B whatever the record.  The bundled scenes' BVH is served from L1/L2, so neither is DRAM "
                        "utilisation: `traffic` (bytes beyond the L2s per launch, request-size bracket's low end) and limiter.measured_hbm_frac are; "
                        "what bounds the kernel is limiter.bound"}
        if serial_kernel_ms[0] and active[0] > 1:
            # with frames in flight a launch shares the GPU with its neighbour: its duration says less about the kernel than the
            # duration of the same launch alone (the calibration's serial frames)
            roof["kernel_ms_alone"] = serial_kernel_ms[0] / launches
            roof["frac_alone"] = req_bytes / (serial_kernel_ms[0] * 1e-3) / 1e9 / HBM_PEAK_GBS
        lim = committed_limiter(workload) if world == 1 and pworld == 1 else None
        if lim:
            roof["limiter"] = lim
            if not lim["stale"]:
                roof["traffic"] = lim.get("beyond_l2_bytes_per_launch")
        out = {
            "metric": "Mrays/sec, tenthousand.txt 1080p@16spp" if (args.scene, W, H, SPP) == HEADLINE
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
            "config": {"workload": workload, "rays_per_frame": int(total_rays), "rays_traversed_per_frame": int(counts[6].item()),
                       "samples_per_frame": int(counts[5].item()), "parallelism": f"image stripes x{world}" if world > 1 else (f"DIAGNOSTIC: part 0 of {pworld} stripe sets on one GPU" if pworld > 1 else "single GPU"), "frames_in_flight": active[0], "frames_in_flight_calibration_ms": fif_ms,
                       "stripe_rows": stripe_rows, "lbvh_build_ms": build_ms, "traversal": raw.get_option("traversal")},
            "roofline": roof,
        }
        if world > 1:
            out["multi_gpu"] = {"backend": backend, ("ranks" if rehearse else "rccl_ranks"): dist.get_world_size(),
                                "physical_gpus": 1 if rehearse else world,
                                "per_rank_render_ms": [float(t[0]) for t in all_ms], "per_rank_gather_ms": [float(t[1]) for t in all_ms],
                                "per_rank_trace_kernel_ms": [float(t[2]) for t in all_ms],
                                "note": "render = this rank's part (device time on its stream); gather = from the end of its render to the end of "
                                        "the framebuffer gather (+ re-interleave on rank 0), i.e. it includes waiting for the slowest rank"}
        if rehearse:
            # never to be read as a multi-GPU measurement: the ranks time-share ONE device
            out["n_gpus"] = 1
            out["rehearsal"] = (f"MIRT_BENCH_REHEARSE=1: {world} ranks share GPU 0 and gather over gloo -- exercises the N > 1 code path "
                                "end to end; value / ms_per_step say nothing about scaling")
            out["config"]["parallelism"] = f"REHEARSAL: image stripes x{world} ranks on one GPU (gloo)"
        if world == 1 and pworld == 1 and not args.headline_only and (args.scene, W, H, SPP) == HEADLINE:
            # the other BASELINE configurations on this GPU (a few frames each, serial)
            extra = []
            raw.close()
            raw = None
            for scene, w, h, spp, steps in (("spiral", 1920, 1080, 16, 5), ("redchair", 3840, 2160, 64, 2)):
                s2 = m.parseInput(os.path.join(ROOT, "scenes", scene + ".txt"))
                r2 = m.initRawConfigFromStl(s2, local_rank)
                m.build_lbvh_karas(r2)
                extra.append(time_config(m, api, torch, r2, w, h, spp, steps, f"{scene}.txt {w}x{h} {spp}spp"))
                r2.close()
            s5 = m.syntheticScene(1_000_000, 1_000_000, seed=1234)
            r5 = m.initRawConfigFromStl(s5, local_rank)
            b5 = m.build_lbvh_karas(r5)
            lab = f"synthetic 1M spheres + 1M triangles 3840x2160 {args.config5_spp}spp (whole frame on this GPU)"
            # (its counter passes were taken on the same scene and frame size at 8 spp: one slab = one launch of the same kernel)
            c5 = time_config(m, api, torch, r5, 3840, 2160, args.config5_spp, 1, lab,
                             pmc_workload="synthetic 1M spheres + 1M triangles 3840x2160 8spp (BASELINE config 5 scene, one slab)")
            c5["lbvh_build_ms"] = b5
            c5["note"] = ("default options: wide quantised records (four grandchild boxes per step, the reference's order; a triangle hit the reference's "
                          "walk may not reach re-walks the exact records) -- same bytes as the exact-record walk on every tested scene, counters == the oracle's mirror")
            extra.append(c5)
            c5b = time_config(m, api, torch, r5, 3840, 2160, args.config5_spp, 1, lab, options={"traversal": 2})
            c5b["lbvh_build_ms"] = b5
            c5b["note"] = ("option traversal = 2: near-child-first at every node; identical bytes on this scene (tests/test_gpu_parity.py), but a "
                           "triangle-silhouette sample may differ where the reference's own result depends on its visiting order")
            extra.append(c5b)
            r5.close()
            out["configs"] = extra
        if world == 1 and args.cpu_step > 0:
            out["cpu_baseline"] = cpu_baseline(scene_file, W, H, SPP, args.cpu_step)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)

    if raw is not None:
        raw.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
