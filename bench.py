#!/usr/bin/env python3
"""Throughput of the per-pixel Monte-Carlo loop on MI355X.

One step = one complete render of the benchmark workload (Cornell box,
1024x1024, depth 6, 1000 samples per pixel, IS+ES+RR) with the scene resident
on the GPU.  With N GPUs the image rows are dealt out in bands, every rank
renders its rows, and the accumulators are gathered on rank 0 (RCCL) inside
the timed region.  Prints one JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

# Algorithmic work per path for this workload: 80 (primary ray) + 6.82 rays x 489 FLOP (seven spheres, one plane,
# hit attributes) + 4.08 bounces x 120 FLOP (shading) + 10 (accumulate).
FLOP_PER_PATH = 3.9e3
PEAK_FP32_TFLOPS = 157.3
PEAK_FP64_TFLOPS = 78.6


def recorded_hbm_traffic(rows, sx):
    """HBM bytes of one launch of the path-tracing kernel on this workload, from the newest counter summary of it
    under profiles/ (written by tools/collect_profiles.sh: FETCH_SIZE doubled as gfx950 needs, plus WRITE_SIZE).
    The counters cannot be read from inside this process, so the figure is the recorded one, scaled to this rank's
    share of the pixels (the traffic is the traced samples, 32 bytes each, and spilled registers)."""
    import glob

    best = None
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "*_pmc.json"))):
        try:
            with open(path) as f:
                rec = json.load(f)
            if rec.get("kernel") != "k_trace_samples" or rec.get("workload", "bench") != "bench":
                continue
            # FP64 operations executed per launch, as the counters saw them: (2 FMA + ADD + MUL wave-instructions) x the
            # average number of active lanes per vector instruction
            flop = rec.get("f64_flop_counted", 0) * rec.get("active_lanes_per_valu_inst", 0) / max(1, rec.get("launches_per_run", 1))
            best = (rec["hbm_bytes_per_launch"]["total"], os.path.basename(path), flop, rec.get("library_sha16"))
        except (OSError, ValueError, KeyError):
            continue
    if best is None:
        return None, None, None, None
    share = (rows * sx) / float(1024 * 1024)
    return best[0] * share, best[1], best[2] * share, best[3]


def library_stamp():
    """First 16 hex digits of the SHA-256 of the kernel library this run loads (tools/pmc_summary.py stamps the counter
    summaries the same way)."""
    import hashlib

    try:
        with open(os.path.join(REPO, "pathtracer_amd", "lib", "libptx_hip.so"), "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()[:16]
    except OSError:
        return None


def cpu_baseline(sx, depth, flags, seconds_target=15.0):
    """Times the CPU side on a bounded sample of the same workload: the compiled reference if it was built
    (oracle/_ref), else the CPU oracle."""
    import ctypes as C

    import numpy as np

    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    ref_lib = os.path.join(REPO, "oracle", "_ref", f"libptref_f{flags}.so")
    small = 256
    dp = C.POINTER(C.c_double)
    if os.path.exists(ref_lib):
        lib = C.CDLL(ref_lib)
        lib.ref_select_scene.argtypes = [C.c_char_p, C.c_int]
        lib.ref_render.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, dp, dp, C.c_int]
        lib.ref_rng_counter.argtypes = [C.c_uint64]
        lib.ref_select_scene(b"cornell", depth)
        lib.ref_rng_counter(1)
        rgb, w = np.zeros(small * small * 3), np.ones(small * small)

        def run(k0, nk):
            lib.ref_render(small, k0, nk, 0.0, rgb.ctypes.data_as(dp), w.ctypes.data_as(dp), threads)

        kind = "reference"
    else:
        import pathtracer_amd as pt
        from harness import Oracle

        orc = Oracle(pt.Scene.from_file(os.path.join(REPO, "tests", "golden", "cornell.ptxscene")))
        prm = pt.make_params(small, depth, flags, 0.0, seed=1)
        rgb, w = np.zeros(small * small * 3), np.ones(small * small)

        def run(k0, nk):
            orc.render(prm, k0, nk, rgb, w, threads)

        kind = "port"
    t0 = time.perf_counter()
    run(0, 4)
    t4 = time.perf_counter() - t0
    spp = max(4, min(2000, int(seconds_target / max(t4 / 4, 1e-6))))
    t0 = time.perf_counter()
    run(4, spp)
    dt = time.perf_counter() - t0
    return {
        "value": small * small * spp / dt,
        "unit": "paths/s",
        "cores": threads,
        "kind": kind,
        "sample": f"Cornell {small}x{small}, depth {depth}, {spp} spp, flags IS+ES+RR, {dt:.1f} s of CPU work, "
                  f"rows shared by {threads} OpenMP threads",
    }


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=1000)
    ap.add_argument("--depth", type=int, default=6)
    ap.add_argument("--chunk", type=int, default=0, help="passes per kernel launch (0 = all of them in one launch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse", action="store_true",
                    help="run the N-rank code path on however many GPUs there are (ranks share GPUs, gloo instead of "
                         "RCCL): a check of the multi-rank plumbing, not a measurement")
    return ap.parse_args()


def main():
    args = parse_args()
    from pathtracer_amd import launch

    if args.gpus > 1 and not launch.under_launcher():
        # plain `python bench.py --gpus N`: this process starts the N ranks (it has not touched the GPU) and hands on
        # rank 0's line.  Under torch.distributed.run the ranks exist already.
        # (bounded: a rank that hangs must not hold the GPUs; PTX_BENCH_TIMEOUT seconds, default 25 minutes)
        rc, out = launch.launch_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus,
                                      timeout=float(os.environ.get("PTX_BENCH_TIMEOUT", "1500")))
        sys.stdout.write(out)
        sys.stdout.flush()
        sys.exit(rc)
    run(args)


def run(args):
    import torch
    import torch.distributed as dist

    import pathtracer_amd as pt
    from pathtracer_amd import parallel

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}")
    if args.rehearse:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    torch.cuda.init()
    torch.zeros(1, device=torch.device("cuda", local_rank))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank)

    flags = pt.abi.USE_IS | pt.abi.USE_ES | pt.abi.USE_RR
    sx, spp, depth = args.size, args.spp, args.depth
    chunk = args.chunk if args.chunk > 0 else spp
    band = parallel.band_rows_for(sx, world)
    scene = pt.Scene.from_file(os.path.join(REPO, "tests", "golden", "cornell.ptxscene"))
    dev = pt.Device(local_rank)
    # The library launches on PyTorch's stream, so that the collective that follows is ordered after the render
    # without a host synchronisation.  That is valid only if both use one copy of the HIP runtime; with two copies
    # the library keeps its own stream and the step waits for it before the gather.
    runtimes = parallel.hip_runtime_copies()
    shared_stream = len(runtimes) == 1
    if shared_stream:
        dev.set_stream(torch.cuda.current_stream().cuda_stream)
    dev.scene_upload(scene)
    prm = pt.make_params(sx, depth, flags, 0.0, seed=2024, rank=rank, n_ranks=world, band_rows=band)
    gat = parallel.Gatherer(sx, rank, world, band, device)  # accumulators, receive buffers and the whole image, once
    rows = gat.rows
    kernel_ms = []
    gather_ev = []  # pairs of events around the gathers of the timed steps, on the stream the gather runs on

    def step(record):
        gat.reset()
        if not shared_stream:
            # the reset is queued on PyTorch's stream, the render on the library's own: order them on the host
            torch.cuda.current_stream().synchronize()
        dev.render_begin(prm, gat.rgb_local.data_ptr(), gat.wght_local.data_ptr())
        k = 0
        while k < spp:
            nk = min(chunk, spp - k)
            dev.render_passes(k, nk)
            k += nk
            # the kernel's own time comes from the library's events around its launches; they are read (which waits
            # for the kernel) only after the gather has been queued behind it, unless another launch is to follow
            if record and k < spp:
                kernel_ms.append(dev.last_trace_ms() + (nk,))
        if not shared_stream:
            dev.sync()
        if record and world > 1:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
            out = gat.gather()
            ev[1].record()
            gather_ev.append(ev)
        else:
            out = gat.gather()
        if record:
            kernel_ms.append(dev.last_trace_ms() + (nk,))
        return out

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        img, _w = step(True)
    fence()
    dt = time.perf_counter() - t0
    stats = dev.stats()
    t = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.rehearse else device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    total_paths = sx * sx * spp * args.steps
    paths_per_s = total_paths / dt
    # dominant kernel: k_trace_samples; a render_passes call of nk passes launches it once per round of samples
    # that fits the sample buffer (once, here), and over its launches it traces rows*sx*nk paths
    ms_sum = sum(m for m, _, _ in kernel_ms)
    n_launches = sum(n for _, n, _ in kernel_ms)
    paths_launches = sum(rows * sx * nk for _, _, nk in kernel_ms)
    achieved_tflops = paths_launches * FLOP_PER_PATH / (ms_sum * 1e-3) / 1e12 if ms_sum > 0 else 0.0
    rays_per_path = (stats["closest_hits"] + stats["shadow_rays"]) / max(1, stats["paths"])
    same_workload = sx == 1024 and spp == 1000 and chunk == spp and depth == 6
    traffic, traffic_src, counted_flop, traffic_lib = recorded_hbm_traffic(rows, sx) if same_workload else (None, None, None, None)
    lib_stamp = library_stamp()
    # every rank's share and kernel time, so that an uneven split or a slow GPU shows in the line
    mine = torch.tensor([float(rows), ms_sum / max(1, args.steps)], dtype=torch.float64, device="cpu" if args.rehearse else device)
    if world > 1:
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
    else:
        every = [mine]
    rows_per_rank = [int(t[0].item()) for t in every]
    per_rank_kernel_ms = [float(t[1].item()) for t in every]
    executed_tflops = counted_flop / (ms_sum / max(1, n_launches) * 1e-3) / 1e12 if counted_flop and ms_sum > 0 else None
    out = {
        "metric": "paths/s (pixel-samples per second), Cornell box 1024^2 depth 6 IS+ES+RR",
        "value": paths_per_s,
        "unit": "paths/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"CornellBox {sx}x{sx}, depth {depth}, {spp} spp, IS=1 ES=1 RR=1, aperture 0",
            "passes_per_launch": chunk,
            "band_rows": band,
            "parallelism": f"rows in bands over {world} GPU(s), gather to rank 0" + (" [rehearsal: shared GPUs, gloo]" if args.rehearse else ""),
            "hip_runtime_copies": runtimes,
        },
        "mrays_per_s": paths_per_s * rays_per_path / 1e6,
        "rays_per_path": rays_per_path,
        "gather_ms_per_step": (sum(a.elapsed_time(b) for a, b in gather_ev) / len(gather_ev)) if gather_ev else 0.0,
        "rows_per_rank": rows_per_rank,
        "per_rank_kernel_ms": {"per_step": per_rank_kernel_ms, "min": min(per_rank_kernel_ms), "max": max(per_rank_kernel_ms)},
        "library_sha16": lib_stamp,
        "roofline": {
            "bound": "valu",
            "achieved": achieved_tflops,
            "peak": PEAK_FP32_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved_tflops / PEAK_FP32_TFLOPS,
            "frac_of_fp64_vector_peak": achieved_tflops / PEAK_FP64_TFLOPS,
            "executed_fp64_tflops": executed_tflops,
            "executed_frac_of_fp64_vector_peak": None if executed_tflops is None else executed_tflops / PEAK_FP64_TFLOPS,
            "traffic": traffic,
            "traffic_source": None if traffic_src is None else f"recorded: profiles/{traffic_src} (same workload, counters of an earlier run)",
            "traffic_profile_library_sha16": traffic_lib,
            "traffic_profile_is_of_this_build": None if traffic_src is None else (traffic_lib is not None and traffic_lib == lib_stamp),
            "algorithmic_hbm_bytes_per_launch": 32 * paths_launches // max(1, n_launches),
            "kernel": "k_trace_samples",
            "kernel_ms_avg": ms_sum / max(1, n_launches),
            "flop_per_path_model": FLOP_PER_PATH,
            "note": "achieved = paths x a MODEL of 3.9 kFLOP per path (not a counter) / kernel time; peak = vector FP32 "
                    "(no MFMA: the work is not a dense contraction); the arithmetic runs in FP64, whose vector peak is "
                    "78.6 TFLOP/s (frac_of_fp64_vector_peak); executed_fp64_tflops = the FP64 operations the hardware "
                    "counters recorded for one launch of this workload (traffic_source) / this run's kernel time",
        },
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sx, depth, flags)
            out["speedup_over_cpu_baseline"] = paths_per_s / out["cpu_baseline"]["value"]
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    dev.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
