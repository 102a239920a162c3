#!/usr/bin/env python3
"""bench.py -- solver-sweep throughput of the hot path on MI355X.

A *step* is one GPUMatrixFreeSolver call (edge-weight pass + K sweeps + copy-back) on one
synthetic image already resident in HBM.  Default workload = BASELINE.json configs[1]:
a single 1920x1080 image, one pyramid level, exactly 1000 Chebyshev-Jacobi sweeps.
Workloads *_rbsor_1e-4 / *_multigrid_1e-4 (BASELINE configs 3 and 5, extensions) solve to a residual instead.
With --gpus N (launched by torch.distributed.run, one rank per GPU) every rank solves its own
image of the same size (independent images shard with no collective; the only communication
is the timing barrier and a MAX-reduce of the elapsed time), so scaling is weak.

Prints ONE JSON line on rank 0; see the task contract for the fields.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    "1080p_jacobi1000": dict(rows=1080, cols=1920, iters=1000),
    "4k_jacobi1000": dict(rows=2160, cols=3840, iters=1000),
    "8k_jacobi200": dict(rows=4320, cols=7680, iters=200),
    "128x128_jacobi1000": dict(rows=128, cols=128, iters=1000),
    "120x67_jacobi1000": dict(rows=67, cols=120, iters=1000),
    "480x270_jacobi250": dict(rows=270, cols=480, iters=250),
    "240x135_jacobi500": dict(rows=135, cols=240, iters=500),
    "960x540_jacobi125": dict(rows=540, cols=960, iters=125),
    # extensions (BASELINE configs 3 and 5): solve from the cold start to a residual max|J(x)-x| <= 1e-4; `iters` is the cap
    "4k_rbsor_1e-4": dict(rows=2160, cols=3840, iters=400000, method="sor_cycles", tolerance=1e-4),
    "1080p_rbsor_1e-4": dict(rows=1080, cols=1920, iters=400000, method="sor_cycles", tolerance=1e-4),
    "8k_multigrid_1e-4": dict(rows=4320, cols=7680, iters=200, method="multigrid", tolerance=1e-4),
    "1080p_multigrid_1e-4": dict(rows=1080, cols=1920, iters=200, method="multigrid", tolerance=1e-4),
}
# Chebyshev-Jacobi x_k 4 + x_{k-1} 4 + x_{k+1} 4 + 4 weight indices 4 + mask 1 = 17 B per pixel-sweep;
# red-black sweep (no x_{k-1}) 13 B.  A V(2,2) cycle (no SURVEY figure: an extension) per image pixel: level 0 = 4 red-black
# sweeps 52 + residual 13 (x 4, indices 4, mask 1, r 4) + restriction 20 (r 4, weights 16) + prolongation 24 (weights 16,
# x read+write 8) = 109 B; the coarse levels hold 1/3 as many points, each 4 sweeps x 48 (9 coefficients 36, e 8, b 4) +
# residual 48 + restriction 20 + prolongation 24 = 284 B -> 95 B per image pixel; 204 B per cycle = 51 B per counted sweep.
ALGO_BYTES = {"jacobi": 17.0, "rbgs": 13.0, "sor_cycles": 13.0, "multigrid": 51.0}
MG_SWEEPS_PER_CYCLE = 4
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(rows, cols, method="jacobi", seconds_target=12.0):
    """Oracle (CPU port of the reference kernels) on this box's host cores, bounded sample."""
    import oracle
    from realtimedepthdiffusion_amd.synth import make_problem
    p = make_problem(rows, cols, seed=1234)
    lut = oracle.load_weights(0.4)
    threads = oracle.max_threads()
    if method in ("rbgs", "sor_cycles"):              # scalar in-place sweep: one thread
        idx = oracle.index_to_weight(p["gray"], None, 0, 0)
        d = p["depth"].copy(); n = 0; t = time.perf_counter()
        while time.perf_counter() - t < seconds_target:
            oracle.rbgs_sweep(d, idx, p["mask"], lut, 1, 1.9); n += 1
        el = time.perf_counter() - t
        return {"value": rows * cols * n / el / 1e6, "unit": "Mpixel-iterations/s", "cores": 1, "kind": "port",
                "sample": f"{n} red-black SOR sweeps of the same {cols}x{rows} problem, scalar C, {el:.1f} s"}
    if method == "multigrid":                         # scalar restatement of the V-cycle: hierarchy setup + ONE cycle
        idx = oracle.index_to_weight(p["gray"], None, 0, 0)
        d = p["depth"].copy(); t = time.perf_counter()
        oracle.mg_solve(d, idx, p["mask"], lut, 1, 1, 0.0, 1)
        el = time.perf_counter() - t
        return {"value": rows * cols * MG_SWEEPS_PER_CYCLE / el / 1e6, "unit": "Mpixel-iterations/s", "cores": 1, "kind": "port",
                "sample": f"hierarchy setup + 1 V-cycle (= {MG_SWEEPS_PER_CYCLE} level-0 sweeps) of the same {cols}x{rows} problem, scalar C, {el:.1f} s"}
    n, el = 16, 0.0
    while True:                               # grow the sample until it is ~seconds_target of CPU work
        d = p["depth"].copy()
        t = time.perf_counter(); oracle.solve(d, p["mask"], p["gray"], n, 0, 0, lut, 1, threads=threads); el = time.perf_counter() - t
        if el >= 0.6 * seconds_target or n >= 200000:
            break
        n = int(min(200000, max(2 * n, n * seconds_target / max(el, 1e-3))))
    return {"value": rows * cols * n / el / 1e6, "unit": "Mpixel-iterations/s", "cores": threads, "kind": "port",
            "sample": f"{n} sweeps of the same {cols}x{rows} problem (incl. the edge-weight pass), OpenMP over rows, {el:.1f} s"}


def estimate_ms(rt, ctx, p, rows, cols, dev, n=20):
    """ms for one whole depth estimate: the coarse-to-fine cascade of /root/reference/src/main.cpp:232-295
    (annotation pyramid, per-level solves with 1000/500/.. sweeps, pyrUp + re-injection, u8 conversion),
    warm-started like --live mode, everything on the device."""
    import numpy as np
    import torch
    bgr = np.repeat(p["gray"][..., None], 3, 2)
    ann = np.where(p["mask"] == 255, p["edited"][..., 0], 32).astype(np.uint8)
    P = ctx.pyramid_create(rows, cols)
    ctx.pyramid_set_image(rt.device_image(bgr, dev)); ctx.pyramid_set_annotation(rt.device_image(ann, dev))
    for _ in range(3):
        ctx.estimate_depth(1000)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        ctx.estimate_depth(1000)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / n * 1e3
    pxit = sum((rows >> l) * (cols >> l) * int(1000 / 2 ** (P - 1 - l)) for l in range(P))
    return {"what": f"{cols}x{rows} {P}-level cascade, {pxit / 1e6:.1f} Mpixel-iterations, device-resident", "ms": ms,
            "Mpixel_iterations_per_s": pxit / ms / 1e3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="1080p_jacobi1000", help="one of %s, or ROWSxCOLSxITERS" % ", ".join(sorted(WORKLOADS)))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-estimate", action="store_true", help="skip the whole-cascade timing leg (profiling runs)")
    ap.add_argument("--sweep-kernel", type=int, default=0)
    ap.add_argument("--temporal-depth", type=int, default=0)
    ap.add_argument("--rows-per-wave", type=int, default=0)
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--persistent", type=int, default=-1)
    ap.add_argument("--method", default=None, choices=["jacobi", "rbgs", "sor_cycles", "multigrid"],
                    help="override the workload's method; everything but jacobi is an EXTENSION (not the headline)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import realtimedepthdiffusion_amd as rt
    from realtimedepthdiffusion_amd.synth import make_problem

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if os.environ.get("RTDD_BENCH_SHARE_GPU"):     # rehearsal on a 1-GPU box: every rank on device 0 (use --persistent 0)
        local = 0
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        # The data path has NO collective (independent images); the process group only serves the timing barrier and
        # the MAX/SUM of two scalars.  RCCL ("nccl") is used when it comes up, gloo otherwise -- the result is the same.
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            dist.barrier()
        except Exception as e:                                  # noqa: BLE001
            print(f"[bench] rank {rank}: nccl unavailable ({e!r}); using gloo for the timing barrier", file=sys.stderr)
            if dist.is_initialized():
                dist.destroy_process_group()
            dist.init_process_group("gloo")

    if args.workload not in WORKLOADS:
        r_, c_, i_ = (int(v) for v in args.workload.split("x"))
        WORKLOADS[args.workload] = dict(rows=r_, cols=c_, iters=i_)
    w = WORKLOADS[args.workload]
    rows, cols, iters = w["rows"], w["cols"], w["iters"]
    method = args.method or w.get("method", "jacobi")
    tolerance = w.get("tolerance", 1e-4 if method in ("sor_cycles", "multigrid") else 0.0)
    p = make_problem(rows, cols, seed=1234 + rank)
    dev = f"cuda:{local}"
    ctx = rt.Context(local)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.GPUAllocateDeviceMemory(rows, cols, 1)
    ctx.GPULoadWeights(0.4)
    if args.sweep_kernel: ctx.set_option(rt.OPT_SWEEP_KERNEL, args.sweep_kernel)
    if args.temporal_depth: ctx.set_option(rt.OPT_TEMPORAL_DEPTH, args.temporal_depth)
    if args.rows_per_wave: ctx.set_option(rt.OPT_ROWS_PER_WAVE, args.rows_per_wave)
    if args.tile: ctx.set_option(rt.OPT_TILE, args.tile)
    if args.persistent >= 0: ctx.set_option(rt.OPT_PERSISTENT, args.persistent)
    mask = rt.device_image(p["mask"], dev); gray = rt.device_image(p["gray"], dev)
    # one pristine initial-depth image per step, uploaded before the clock starts
    depths = [rt.device_image(p["depth"], dev) for _ in range(args.steps + args.warmup)]

    executed = []                           # iterations actually run per step (residual-stopped methods), and the residual reached

    def step(i):
        if method == "rbgs":
            ctx.solve_ex(depths[i], mask, gray, rows, cols, 0, method=rt.METHOD_RED_BLACK_GS, maxIterations=iters, tolerance=0.0)
        elif method == "sor_cycles":
            executed.append(ctx.solve_ex(depths[i], mask, gray, rows, cols, 0, method=rt.METHOD_RED_BLACK_GS, maxIterations=iters, tolerance=tolerance,
                                         relaxation=rt.RELAXATION_AUTO))
        elif method == "multigrid":
            executed.append(ctx.solve_ex(depths[i], mask, gray, rows, cols, 0, method=rt.METHOD_MULTIGRID, maxIterations=iters, tolerance=tolerance))
        else:
            ctx.GPUMatrixFreeSolver(depths[i], mask, gray, rows, cols, 0.4, iters, 1e-5, 0)

    from realtimedepthdiffusion_amd import shard

    def fence():
        shard.fence(dist, torch.cuda.synchronize)

    for i in range(args.warmup):
        step(i)
    executed.clear()
    ctx.profile_enable(True)               # HIP events around the sweep launches, on the launch stream
    sweep_ms = 0.0; launches = 0; sweeps = 0
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)                  # asynchronous: nothing in the timed loop waits for the GPU
    fence()
    elapsed = time.perf_counter() - t0
    pr = ctx.profile(); sweep_ms = pr.sweep_ms; launches = pr.launches; sweeps = pr.sweeps   # events recorded inside the timed region
    if executed:                              # residual-stopped: count what actually ran (a V-cycle = its level-0 sweeps)
        per = MG_SWEEPS_PER_CYCLE if method == "multigrid" else 1
        px_iter_per_step = rows * cols * per * sum(e[0] for e in executed) / len(executed)
    else:
        px_iter_per_step = rows * cols * iters
    algo_bytes = ALGO_BYTES[method]
    agg_dev = dev if (dist is None or dist.get_backend() == "nccl") else "cpu"
    units, elapsed, thr = shard.aggregate(args.steps * px_iter_per_step, elapsed, dist, agg_dev)   # SUM of units, MAX of time
    value = thr / 1e6
    launch_us = sweep_ms * 1e3 / max(launches, 1)
    if method == "multigrid": sweeps *= MG_SWEEPS_PER_CYCLE
    sweeps_per_launch = sweeps / max(launches, 1)
    achieved = algo_bytes * rows * cols * sweeps_per_launch / (launch_us * 1e-6) / 1e9
    out = {
        "metric": "Mpixel-iterations/s (solver sweep)", "value": value, "unit": "Mpixel-iterations/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: one {cols}x{rows} image per GPU, 1 level, {iters} Chebyshev-Jacobi sweeps "
                               f"(BASELINE configs[1])" if args.workload == "1080p_jacobi1000" and method == "jacobi" else f"{args.workload} ({method})",
                   "images_per_step": world, "sweep_kernel": ctx.get_option(rt.OPT_SWEEP_KERNEL), "tile": ctx.get_option(rt.OPT_TILE), "temporal_depth": ctx.get_option(rt.OPT_TEMPORAL_DEPTH), "persistent": ctx.get_option(rt.OPT_PERSISTENT),
                   "sweeps_per_launch": sweeps_per_launch},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": None, "kernel": {"jacobi": "k_sweep_blocked", "rbgs": "k_rbgs_blocked", "sor_cycles": "k_rbgs_blocked (+ residual checks)",
                                                 "multigrid": "whole V(2,2) cycle, all levels: 204 B per image pixel and cycle"}[method], "launch_us": launch_us,
                     "algorithmic_bytes_per_launch": algo_bytes * rows * cols * sweeps_per_launch},
    }
    if executed:
        out["config"]["converged"] = {"tolerance": tolerance, "iterations": [e[0] for e in executed], "unit": "cycles" if method == "multigrid" else "sweeps",
                                      "residual": max(e[1] for e in executed), "start": "cold (depth 255 + labels)"}
    # HBM-side traffic of the sweep kernel comes from a separate rocprofv3 --pmc pass of this same command
    # (scripts/profile_round.sh; FETCH_SIZE x2 + WRITE_SIZE, MI355X_MICROARCH.md section HBM), committed under profiles/.
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))) + [os.path.join(ROOT, "profiles", "traffic_latest.json")]:
        try:
            prof = json.load(open(path))
            if prof.get("workload") == args.workload and method == "jacobi" and not (args.sweep_kernel or args.tile or args.temporal_depth or args.persistent >= 0):
                sweeps_k = [k for name, k in prof["kernels"].items() if "k_sweep" in name]
                if sweeps_k:                            # the dominant kernel of that profile; a later file (traffic_latest last) overrides an earlier one
                    out["roofline"]["traffic"] = max(k["hbm_bytes_per_launch_corrected"] for k in sweeps_k)
                    out["roofline"]["traffic_source"] = prof.get("source", os.path.relpath(path, ROOT))
        except (OSError, ValueError, KeyError):
            pass
    if rank == 0 and args.workload == "1080p_jacobi1000" and method == "jacobi" and not args.no_estimate:
        out["estimate"] = estimate_ms(rt, ctx, p, rows, cols, dev)      # second half of BASELINE's metric; outside the timed region
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(rows, cols, method)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
