#!/usr/bin/env python3
"""bench.py -- solver-sweep throughput of the hot path on MI355X.

A *step* is one pass of the hot path over one batch of synthetic input already resident in HBM: for the default workload
(BASELINE.json configs[1]) one GPUMatrixFreeSolver call -- edge-weight pass + 1000 Chebyshev-Jacobi sweeps + copy-back -- on a
single 1920x1080 image, one pyramid level.  Other workloads: --workload 4k_jacobi1000 (the north star's 4K sweep),
*_rbsor_1e-4 / *_multigrid_1e-4 (BASELINE configs 3 and 5, extensions: solve to a residual), batch64_1080p (config 4: 64
independent 1080p images, image i on rank i % N, one stream per GPU).

--gpus N: one process per GPU.  Launched by `torch.distributed.run` (RANK / LOCAL_RANK / WORLD_SIZE in the environment) this
process is one rank; launched plainly (`python3 bench.py --gpus N`) it starts the N ranks itself, before anything touches the
GPU, and relays rank 0's line.  Images are independent, so there is NO collective on the data path: the process group only
carries the timing barrier and a MAX / SUM of two scalars.  Default workload: every rank solves its own image of the same
size ("weak"); batch64_1080p splits a fixed batch ("strong").

Prints ONE JSON line on rank 0; see the task contract for the fields.  `roofline` is on the resource that binds -- VALU issue:
the tile lives in registers for all sweeps, HBM is idle -- with the 17 B/pixel-sweep HBM-equivalent
rate beside it.  Outside the timed region, on rank 0 at N = 1, the line also carries `estimate` (ms to a depth map: the whole
1080p cascade), `sweep_4k` (the 4K sweep rate) and `effects` (the per-pixel passes), and `cpu_baseline` (the oracle on the host).
"""
import argparse
import json
import os
import subprocess
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
    # BASELINE config 4: a fixed batch of 64 independent 1080p images (distinct seeds), image i on rank i % N
    "batch64_1080p": dict(rows=1080, cols=1920, iters=1000, batch=64),
    # the same batch as whole ESTIMATES (5-level cascades, src/main.cpp:232-295): a rank's images form one batched pyramid, every level of all
    # of them in the same launches (rtdd_estimate_depth_batch); image i on rank i % N
    "batch64_1080p_estimate": dict(rows=1080, cols=1920, iters=1000, batch=64, estimate=True),
    # extensions (BASELINE configs 3 and 5): solve from the cold start to a residual max|J(x)-x| <= 1e-4; `iters` is the cap
    "4k_rbsor_1e-4": dict(rows=2160, cols=3840, iters=400000, method="sor_cycles", tolerance=1e-4),
    "1080p_rbsor_1e-4": dict(rows=1080, cols=1920, iters=400000, method="sor_cycles", tolerance=1e-4),
    "8k_multigrid_1e-4": dict(rows=4320, cols=7680, iters=200, method="multigrid", tolerance=1e-4),
    "1080p_multigrid_1e-4": dict(rows=1080, cols=1920, iters=200, method="multigrid", tolerance=1e-4),
    # RTDD_METHOD_AUTO: V-cycles while they pay, then SOR cycles -- the documented entry point for config 5 (V-cycles alone stall on photographs)
    "8k_auto_1e-4": dict(rows=4320, cols=7680, iters=400000, method="auto", tolerance=1e-4),
    "1080p_auto_1e-4": dict(rows=1080, cols=1920, iters=400000, method="auto", tolerance=1e-4),
}
# Algorithmic HBM bytes per pixel-sweep: Chebyshev-Jacobi 17 (x_k 4 + x_{k-1} 4 + x_{k+1} 4 + 4 weight indices 4 + mask 1);
# red-black 13 (no x_{k-1}).  A V(2,2) cycle is counted as its 4 level-0 red-black sweeps for `value`; its algorithmic traffic per image
# pixel is: level 0 -- 4 sweeps 52 + residual 13 (x 4, indices 4, mask 1, r 4) + restriction 20 (r 4, weights 16) + prolongation 24
# (weights 16, x read+write 8) = 109 B; the coarse levels hold 1/3 as many points, each 4 sweeps x 48 (9 coefficients 36, e 8, b 4) +
# residual 48 + restriction 20 + prolongation 24 = 284 B -> 95 B per image pixel; 204 B per cycle = 51 B per counted sweep.
ALGO_BYTES = {"jacobi": 17.0, "rbgs": 13.0, "sor_cycles": 13.0, "multigrid": 51.0, "auto": 13.0}
MG_SWEEPS_PER_CYCLE = 4
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: 8.0 TB/s spec
# VALU roof: 256 CUs x 4 SIMD-32 x 32 lanes x 2.4 GHz (MI355X_MICROARCH.md chip table) = 78.6 T lane-operations/s (= 157.3 TFLOP/s / 2).
VALU_PEAK_TOPS = 256 * 4 * 32 * 2.4e9 / 1e12
# Algorithmic VALU operations per pixel-sweep, counted in the ISA of the sweep body (k_sweep_blocked<32,1024,3,true,true>, fast
# path: 362 VALU instructions per 2 sweeps x 12 pixels = 15.1; 4 fma for the sum, 3 for the divide, 1.5 for the tiny-numerator test, 6
# for clamp + update + Dirichlet select, ~0.6 bookkeeping; the lane shifts are LDS-crossbar permutes since round 3 and no longer VALU
# work: 16.3 before).  Halo redundancy is NOT counted: `achieved` is useful work.
VALU_OPS = {"jacobi": 14.5, "rbgs": 15.0, "sor_cycles": 15.0}
# (14.5 since round 5: 348 per sweep pair -- the Dirichlet select became the EXEC mask of the update's last fma; 15.1 in rounds 3-4, which
# `frac_at_round4_count` keeps comparable.  The Jacobi figure is RECOUNTED from the built object at run time where llvm-objdump is there
# (scripts/isa_count.py: the fall-through path of the sweep-pair loop of k_sweep_blocked<32,1024,3,true,true>); the constant is the fallback
# and tests/test_isa_hazards.py keeps the two equal.)
VALU_OPS_ROUND4 = 15.1
_valu_ops_source = {}


def jacobi_valu_ops():
    if "v" not in _valu_ops_source:
        try:
            sys.path.insert(0, os.path.join(ROOT, "scripts"))
            import isa_count
            c = isa_count.sweep_pair()
            _valu_ops_source.update(v=c["per_pixel_sweep"], src=f"counted now in the disassembly of the built kernel: {c['valu']} VALU instructions per sweep pair x 12 pixels (scripts/isa_count.py)")
        except Exception as e:                      # no llvm-objdump / no object on this box
            _valu_ops_source.update(v=VALU_OPS["jacobi"], src=f"recorded constant (the built object could not be disassembled here: {type(e).__name__})")
    return _valu_ops_source["v"], _valu_ops_source["src"]
# What the sweep's instruction MIX can issue at: the kernel's own hot block replayed as a micro-benchmark (scripts/ubench/gen_block_bench.py,
# profiles/r03_block_replay.txt) issues at 2.66 cycles per wave-instruction and SIMD with every CU busy at 4 waves per SIMD (2.27 with one
# CU busy: the chip clocks down under load) -- clamp, compare and select forms run at half rate -- against the 2 cycles `peak` assumes.
# Reported beside `frac`, never instead of it.
VALU_MIX_CYCLES = 2.66


def free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def launch_ranks(n, argv, timeout_s=1500.0):
    """`python3 bench.py --gpus N` without a launcher: start the N ranks (one process per GPU) and relay rank 0's JSON line.
    Nothing in this process has touched the GPU (no torch import yet).  Every child is watched: the first rank that exits non-zero
    (or the deadline) ends the others -- fresh children, never an exec -- so a rank that dies before the rendezvous cannot leave the
    rest waiting in init_process_group."""
    import tempfile
    port = free_port()
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=out0 if r == 0 else subprocess.DEVNULL, text=True))
    deadline = time.time() + timeout_s
    failed = 0
    while True:
        rcs = [p.poll() for p in procs]
        bad = [rc for rc in rcs if rc not in (None, 0)]
        if bad or time.time() > deadline:
            failed = abs(bad[0]) if bad else 124
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t_end = time.time() + 10
            for p in procs:
                try:
                    p.wait(timeout=max(0.1, t_end - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
            print(f"[bench] launcher: a rank exited with {failed} (or the deadline passed); the other ranks were stopped", file=sys.stderr)
            break
        if all(rc == 0 for rc in rcs):
            break
        time.sleep(0.05)
    out0.seek(0)
    sys.stdout.write(out0.read())
    sys.stdout.flush()
    return failed


def cpu_baseline(rows, cols, method="jacobi", seconds_target=12.0):
    """Oracle (CPU port of the reference kernels) on this box's host cores, bounded sample."""
    import oracle
    from realtimedepthdiffusion_amd.synth import make_problem
    p = make_problem(rows, cols, seed=1234)
    lut = oracle.load_weights(0.4)
    threads = oracle.max_threads()
    if method in ("rbgs", "sor_cycles"):              # the in-place red-black sweep, each colour's rows over the host cores (orc_rbgs_sweeps_mt)
        idx = oracle.index_to_weight(p["gray"], None, 0, 0)
        d = p["depth"].copy(); n = 0; t = time.perf_counter()
        while time.perf_counter() - t < seconds_target:
            oracle.rbgs_sweeps_mt(d, idx, p["mask"], lut, 1, 1.9, 8, threads=threads); n += 8
        el = time.perf_counter() - t
        return {"value": rows * cols * n / el / 1e6, "unit": "Mpixel-iterations/s", "cores": threads, "kind": "port",
                "sample": f"{n} red-black SOR sweeps of the same {cols}x{rows} problem, OpenMP over the rows of each colour, {el:.1f} s"}
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
    ctx.synchronize()
    ms = (time.perf_counter() - t) / n * 1e3
    # like for like with the reference's frame (src/main.cpp:239-259 rebuilds the annotation pyramid and re-injects the coarsest level
    # EVERY frame): the same estimates with the annotation declared changed in front of each
    t = time.perf_counter()
    for _ in range(n):
        ctx.pyramid_annotation_changed(); ctx.estimate_depth(1000)
    ctx.synchronize()
    ms_ann = (time.perf_counter() - t) / n * 1e3
    pxit = sum((rows >> l) * (cols >> l) * int(1000 / 2 ** (P - 1 - l)) for l in range(P))
    kernels = []
    for l in range(1):                          # what the finest level ran (rtdd_last_solve_info reports the last solve: level 0)
        i = ctx.last_solve_info(); kernels.append(f"level 0: tile {i.tile}, {i.temporal_depth} sweeps per launch, persistent {i.persistent}")
    # The region the reference itself clocks (src/main.cpp:234-293): upload of the host's scribble + edited images (:236-237), the
    # cascade, convertTo + download of the u8 map (:290-291) -- page-locked host images, rtdd_live_submit / rtdd_live_wait.  One frame
    # at a time = that region as it stands; two frames in flight (the copies on a second stream) = what a live loop sustains.
    scr = rt.host_image((rows, cols)); ed = rt.host_image((rows, cols, 3)); outs = [rt.host_image((rows, cols)) for _ in range(2)]
    scr.a[...] = ctx.pyramid_download(rt.IMG_SCRIBBLE, 0); ed.a[...] = ctx.pyramid_download(rt.IMG_EDITED, 0)
    for _ in range(3):
        ctx.live_submit(scr.a, ed.a, outs[0].a, 1000); ctx.live_wait()
    t = time.perf_counter()
    for _ in range(n):
        ctx.live_submit(scr.a, ed.a, outs[0].a, 1000); ctx.live_wait()
    ms_e2e = (time.perf_counter() - t) / n * 1e3
    frames = max(200, 2 * n)                     # (long enough that the first upload and the last download, which overlap nothing, do not show)
    t = time.perf_counter()
    for f in range(frames):
        if f >= 2:
            ctx.live_wait()
        ctx.live_submit(scr.a, ed.a, outs[f % 2].a, 1000)
    while ctx.live_pending():
        ctx.live_wait()
    ms_live = (time.perf_counter() - t) / frames * 1e3
    ctx.synchronize()
    # the reference's frame WITH a sticky effect (src/main.cpp:190-230 in every iteration of the loop at :180, next to the estimate):
    # the same pipelined loop, every frame also rendering the effect from its own depth map and bringing the artistic image to the host
    arts = [rt.host_image((rows, cols, 3)) for _ in range(2)]
    live_fx = {}
    for name, code in (("defocus", rt.EFFECT_DEFOCUS), ("desaturation", rt.EFFECT_DESATURATION), ("haze", rt.EFFECT_HAZE)):
        for f in range(4):
            ctx.live_submit_ex(scr.a, ed.a, outs[f % 2].a, code, arts[f % 2].a, 1000); ctx.live_wait()
        t = time.perf_counter()
        for f in range(frames):
            if f >= 2:
                ctx.live_wait()
            ctx.live_submit_ex(scr.a, ed.a, outs[f % 2].a, code, arts[f % 2].a, 1000)
        while ctx.live_pending():
            ctx.live_wait()
        live_fx[name] = (time.perf_counter() - t) / frames * 1e3
    t = time.perf_counter()
    for _ in range(n):
        ctx.live_submit_ex(scr.a, ed.a, outs[0].a, rt.EFFECT_DEFOCUS, arts[0].a, 1000); ctx.live_wait()
    ms_e2e_fx = (time.perf_counter() - t) / n * 1e3
    ctx.synchronize()
    ctx.pyramid_destroy()
    for h in [scr, ed] + outs + arts:
        h.free()
    h2d, d2h = rows * cols * 4, rows * cols
    return {"what": f"{cols}x{rows} {P}-level cascade, {pxit / 1e6:.1f} Mpixel-iterations, device-resident, annotation unchanged between estimates: the annotation pyramid is NOT rebuilt "
                    "(ms_annotation_changed: rebuilt every estimate, as src/main.cpp:239-259 does)", "ms": ms, "ms_annotation_changed": ms_ann,
            "Mpixel_iterations_per_s": pxit / ms / 1e3, "finest_level": kernels[0],
            "ms_end_to_end": ms_e2e, "end_to_end_is": f"src/main.cpp:234-293 as the reference clocks it: H2D of scribble + edited ({h2d / 1e6:.1f} MB, page-locked), annotation pyramid, cascade, "
                                                      f"D2H of the u8 map ({d2h / 1e6:.1f} MB), host waits for every frame",
            "live_ms_per_frame": ms_live, "live_fps": 1e3 / ms_live, "live_is": "the same frames, two in flight: copies on a second stream overlap the other frame's arithmetic (rtdd_live_submit)",
            "live_ms_per_frame_defocus": live_fx["defocus"], "live_fps_defocus": 1e3 / live_fx["defocus"],
            "live_ms_per_frame_desaturation": live_fx["desaturation"], "live_ms_per_frame_haze": live_fx["haze"], "ms_end_to_end_defocus": ms_e2e_fx,
            "live_effect_is": f"the reference's frame with a sticky effect (src/main.cpp:190-230 + 232-295; rtdd_live_submit_ex): the same pipelined frames, each also rendering the effect from its own "
                              f"depth map and downloading the artistic image ({rows * cols * 3 / 1e6:.1f} MB); ms_end_to_end_defocus: one such frame at a time"}


def estimate_dataset(rt, dev, n=10):
    """ms per estimate on the twelve bundled photographs at their own resolution (tests/golden/dataset: the reference's dataset/ pairs,
    decoded): device-resident, warm-started, the annotation pyramid rebuilt every estimate as src/main.cpp:239-259 does.  Real photographs
    cost more than the synthetic image of `estimate` (waves that meet denormal edge weights take the IEEE divide): the line says how much."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        import dataset_util
    except Exception as e:                       # (no PIL on this box, or the fixtures are absent)
        return {"skipped": f"{type(e).__name__}: {e}"}
    per = {}
    ctx = rt.Context(int(dev.split(":")[1])); ctx.set_stream(torch.cuda.current_stream().cuda_stream); ctx.GPULoadWeights(0.4)
    for name in dataset_util.PAIRS:
        bgr, ann, _ = dataset_util.load_pair(name)
        rows, cols = ann.shape
        ctx.pyramid_create(rows, cols)
        ctx.pyramid_set_image(rt.device_image(bgr, dev)); ctx.pyramid_set_annotation(rt.device_image(ann, dev))
        for _ in range(3):
            ctx.estimate_depth(1000)
        ctx.synchronize(); t = time.perf_counter()
        for _ in range(n):
            ctx.pyramid_annotation_changed(); ctx.estimate_depth(1000)
        ctx.synchronize()
        per[name] = {"ms": round((time.perf_counter() - t) / n * 1e3, 4), "size": f"{cols}x{rows}"}
    ctx.close()
    ms = [v["ms"] for v in per.values()]
    lo, hi = min(per, key=lambda k: per[k]["ms"]), max(per, key=lambda k: per[k]["ms"])
    return {"min_ms": per[lo]["ms"], "min_is": f"{lo} {per[lo]['size']}", "max_ms": per[hi]["ms"], "max_is": f"{hi} {per[hi]['size']}", "mean_ms": sum(ms) / len(ms), "pairs": per,
            "what": "the twelve bundled image / annotation pairs at their own resolution, device-resident whole estimates (annotation pyramid rebuilt every estimate), warm-started"}


def dropin_frame(rt, dev, p, rows, cols, n=20):
    """What an UNCHANGED main.cpp pays per depth estimate (src/main.cpp:239-291): the reference's ten functions through their Itanium-mangled
    symbols (csrc/dropin.cpp, the process-global context on the null stream), one call after the other in main.cpp's order, the caller's own
    pitched device images, GPUMatrixFreeSolver returning synchronised at every level (src/GPUSolver.cu:314) -- 4 x GPUPyrDownAnnotation,
    GPUConvertToFloat, 5 x (GPUMatrixFreeSolver, pyrUp, GPUConvertToFloat), convertTo.  cv::cuda::pyrUp (:273) and GpuMat::convertTo (:290) are
    OpenCV's on the reference side; the library's rtdd_pyrup_depth / rtdd_depth_to_u8 on the shim's context stand in for them.  Device-resident
    like `estimate.ms` (no upload / download), warm-started, timed host-side over n frames.  Reported beside the fused rtdd_estimate_depth."""
    import ctypes as C
    import math
    import numpy as np
    import torch
    L = rt.lib()
    vp, sz, i32, f32 = C.c_void_p, C.c_size_t, C.c_int, C.c_float

    def fn(name):
        f = getattr(L, name); f.restype = None
        return f

    def img(t):
        return vp(t.data_ptr()), sz(t.stride(0) * t.element_size())
    P = int(math.log2(max(min(cols, rows) // 45, 1)) + 1)
    sizes = [(int(np.float32(rows) / np.float32(2.0) ** l), int(np.float32(cols) / np.float32(2.0) ** l)) for l in range(P)]
    L.rtdd_dropin_context.restype = vp
    h = vp(L.rtdd_dropin_context())
    fn("_Z23GPUAllocateDeviceMemoryiii")(i32(rows), i32(cols), i32(P))
    fn("_Z14GPULoadWeightsf")(f32(0.4))
    # the caller's images (main.cpp:117-137), gray pyramid built once through the library's restated cv::pyrDown
    gray = [rt.device_image(p["gray"], dev)]
    for l in range(1, P):
        gr, gc = gray[-1].shape
        gray.append(rt.device_image(np.zeros(((gr + 1) // 2, (gc + 1) // 2), np.uint8), dev))
        a, ap = img(gray[-2]); b, bp = img(gray[-1])
        assert L.rtdd_pyrdown_gray(h, a, ap, i32(gr), i32(gc), b, bp) == 0
    gray = [g[:s[0], :s[1]] for g, s in zip(gray, sizes)]                      # (the ceil chain's images, used at the floor sizes like main.cpp does)
    edited = [rt.device_image(np.zeros(s + (3,), np.uint8), dev) for s in sizes]; scribble = [rt.device_image(np.zeros(s, np.uint8), dev) for s in sizes]
    edited[0] = rt.device_image(p["edited"], dev); scribble[0] = rt.device_image(p["mask"], dev)
    depth = [rt.device_image(np.full(s, 255.0, np.float32), dev) for s in sizes]
    u8 = rt.device_image(np.zeros(sizes[0], np.uint8), dev)
    pyrdown = fn("_Z20GPUPyrDownAnnotationPhmS_miiS_mS_mii"); convert = fn("_Z17GPUConvertToFloatPhmPfmS_mii"); solve = fn("_Z19GPUMatrixFreeSolverPfmPhmS0_miififi")

    def frame():
        for l in range(1, P):                                                   # :249-253
            pyrdown(*img(scribble[l - 1]), *img(edited[l - 1]), i32(sizes[l - 1][0]), i32(sizes[l - 1][1]), *img(scribble[l]), *img(edited[l]), i32(sizes[l][0]), i32(sizes[l][1]))
        convert(*img(edited[P - 1]), *img(depth[P - 1]), *img(scribble[P - 1]), i32(sizes[P - 1][0]), i32(sizes[P - 1][1]))      # :257
        for l in range(P - 1, -1, -1):
            iters = int(np.float32(1000) / np.float32(2.0) ** ((P - 1) - l))
            solve(*img(depth[l]), *img(scribble[l]), *img(gray[l]), i32(sizes[l][0]), i32(sizes[l][1]), f32(0.4), i32(iters), f32(1e-5), i32(l))     # :266, returns synchronised
            if l > 0:
                a, ap = img(depth[l]); b, bp = img(depth[l - 1])
                assert L.rtdd_pyrup_depth(h, a, ap, i32(sizes[l][0]), i32(sizes[l][1]), b, bp, i32(sizes[l - 1][0]), i32(sizes[l - 1][1])) == 0       # cv::cuda::pyrUp, :273
                convert(*img(edited[l - 1]), *img(depth[l - 1]), *img(scribble[l - 1]), i32(sizes[l - 1][0]), i32(sizes[l - 1][1]))                  # :281
        a, ap = img(depth[0]); b, bp = img(u8)
        assert L.rtdd_depth_to_u8(h, a, ap, b, bp, i32(rows), i32(cols)) == 0      # convertTo, :290
    def timed():
        for _ in range(3):
            frame()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n):
            frame()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3
    ms = timed()
    info = rt.SolveInfo(); L.rtdd_last_solve_info(h, C.byref(info))
    sha = __import__("hashlib").sha256(rt.to_host(u8).tobytes()).hexdigest()[:16]
    fn("_Z19GPUFreeDeviceMemoryi")(i32(P))
    return {"ms": ms, "what": f"the ten mangled symbols of include/*.h in main.cpp's order (src/main.cpp:239-291), {cols}x{rows}, {P} levels, device-resident, the caller's own pitched images, "
                              "every GPUMatrixFreeSolver returning synchronised; rtdd_pyrup_depth / rtdd_depth_to_u8 stand in for cv::cuda::pyrUp / convertTo",
            "calls_per_frame": (P - 1) + 1 + P + 2 * (P - 1) + 1, "device_syncs_per_frame": P, "finest_level": f"tile {info.tile}, persistent {info.persistent}", "u8_sha256_16": sha}


def batch_estimates(rt, dev, rows=1080, cols=1920, images=64, reps=3):
    """BASELINE configs[3] on ONE GPU as estimates: `images` independent images, every pyramid level of all of them in the same launches
    (rtdd_estimate_depth_batch) against the same estimates one after the other (rtdd_pyramid_select + rtdd_estimate_depth).  Images are
    device-resident and warm-started (fixed sweep counts: the work per estimate does not depend on the start)."""
    import numpy as np
    import torch
    from realtimedepthdiffusion_amd.synth import make_problem
    ctx = rt.Context(int(dev.split(":")[1])); ctx.set_stream(torch.cuda.current_stream().cuda_stream); ctx.GPULoadWeights(0.4)
    P = ctx.pyramid_create_batch(rows, cols, images)
    for b in range(images):
        p = make_problem(rows, cols, seed=1234 + b)
        ctx.pyramid_select(b)
        ctx.pyramid_set_image(rt.device_image(np.repeat(p["gray"][..., None], 3, 2), dev))
        ctx.pyramid_set_annotation(rt.device_image(np.where(p["mask"] == 255, p["edited"][..., 0], 32).astype(np.uint8), dev))
    ctx.estimate_depth_batch(1000); ctx.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        ctx.estimate_depth_batch(1000)
    ctx.synchronize()
    ms_b = (time.perf_counter() - t) / reps * 1e3
    t = time.perf_counter()
    for _ in range(reps):
        for b in range(images):
            ctx.pyramid_select(b); ctx.estimate_depth(1000)
    ctx.synchronize()
    ms_s = (time.perf_counter() - t) / reps * 1e3
    ctx.close()
    pxit = sum((rows >> l) * (cols >> l) * int(1000 / 2 ** (P - 1 - l)) for l in range(P)) * images
    return {"what": f"{images} independent {cols}x{rows} images, {P}-level cascades ({pxit / 1e6:.0f} Mpixel-iterations), device-resident, one GPU",
            "batched_ms": ms_b, "batched_estimates_per_s": images / ms_b * 1e3, "batched_ms_per_estimate": ms_b / images,
            "sequential_ms": ms_s, "sequential_estimates_per_s": images / ms_s * 1e3, "speedup": ms_s / ms_b,
            "Mpixel_iterations_per_s": pxit / ms_b / 1e3}


def verify_estimates(rt, ctx, problems, my_images, iters, levels, dev, world, sample=(0, 1, 7, 8, 31, 32, 62, 63)):
    """The checker of the estimate workloads (outside the timed region): a sample of this rank's images -- the batch positions VERDICT r5
    names where the rank owns them, else its first and last ones, 8 at most -- is put back into its cold state (rtdd_pyramid_set_image
    resets the warm start), the WHOLE batch runs once more (same launches, same per-level choices as the timed steps: the choice depends on
    the level's size and the batch size only), and every level + the u8 map of the sampled images is compared with the oracle's cascade
    (src/main.cpp:232-295); then once more warm-started.  Returns (global indices that differ, global indices checked)."""
    import numpy as np
    import oracle
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cascade_ref import Cascade
    lut = oracle.load_weights(0.4)
    local = [k for k, g in enumerate(my_images) if g in sample]
    if len(local) < min(4, len(my_images)):
        local = sorted(set(local) | set(range(min(2, len(my_images)))) | set(range(max(0, len(my_images) - 2), len(my_images))))
    local = local[:8]
    refs = {}
    for k in local:
        p = problems[k]
        bgr = np.repeat(p["gray"][..., None], 3, 2); ann = np.where(p["mask"] == 255, p["edited"][..., 0], 32).astype(np.uint8)
        ctx.pyramid_select(k); ctx.pyramid_set_image(rt.device_image(bgr, dev)); ctx.pyramid_set_annotation(rt.device_image(ann, dev))
        refs[k] = Cascade(oracle, bgr, ann, lut, 1, threads=max(1, oracle.max_threads() // world))
    bad = set()
    for _ in range(2):
        ctx.estimate_depth_batch(iters); ctx.synchronize()
        for k in local:
            refs[k].estimate(iters)
            ctx.pyramid_select(k)
            same = np.array_equal(ctx.pyramid_download(rt.IMG_DEPTH_U8), refs[k].depth_u8)
            for l in range(levels):
                same = same and np.array_equal(ctx.pyramid_download(rt.IMG_DEPTH, l).view(np.uint32), refs[k].depth[l].view(np.uint32))
            if not same:
                bad.add(my_images[k])
    return sorted(bad), [my_images[k] for k in local]


def sustained(step, sync, px_iter_per_step, seconds=3.0, window=0.5):
    """>= `seconds` of back-to-back steps (the headline's solve), the rate per `window`: the clocks settle in the first windows, and a
    sampler that looks every few seconds sees a busy GPU.  Reported, never `value`."""
    rates, n_total = [], 0
    t_end = time.perf_counter() + seconds
    while time.perf_counter() < t_end:
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < window:
            for _ in range(8):
                step()
            n += 8
            sync()
        rates.append(n * px_iter_per_step / (time.perf_counter() - t0) / 1e6)
        n_total += n
    return {"seconds": seconds, "window_s": window, "steps": n_total, "unit": "Mpixel-iterations/s", "mean": sum(rates) / len(rates), "min": min(rates), "max": max(rates),
            "first_window": rates[0], "last_window": rates[-1], "windows": [round(r, 1) for r in rates],
            "what": "the headline's solve back to back (each group of 8 solves synchronised), rate per window: DVFS settling is the drift from the first windows to the last"}


def photo_problem(rows, cols):
    """A photograph-like problem of any size: the bundled Dog pair (tests/golden/Dog_full.npz: 672 x 624, decoded) tiled, every other
    copy mirrored so that the seams are no edges.  Thin high-contrast structures at pixel scale -- what the synthetic images lack and
    what stalls the V-cycle."""
    import numpy as np
    g = np.load(os.path.join(ROOT, "tests", "golden", "Dog_full.npz"), allow_pickle=False)
    bgr, ann = g["bgr"], g["annotation"]

    def tile(a):
        a2 = np.concatenate([a, a[:, ::-1]], 1); a4 = np.concatenate([a2, a2[::-1]], 0)
        ry, rx = -(-rows // a4.shape[0]), -(-cols // a4.shape[1])
        return np.ascontiguousarray(np.tile(a4, (ry, rx) + (1,) * (a.ndim - 2))[:rows, :cols])
    bgr, ann = tile(bgr), tile(ann)
    gray = ((bgr[..., 0].astype(np.int32) * 1868 + bgr[..., 1].astype(np.int32) * 9617 + bgr[..., 2].astype(np.int32) * 4899 + 8192) >> 14).astype(np.uint8)
    mask = np.where(ann != 32, 255, 32).astype(np.uint8)                       # src/main.cpp:160-168
    depth = np.where(ann != 32, ann, 255).astype(np.float32)
    return {"gray": gray, "mask": mask, "depth": depth}


def photo_record(rt, dev, rows, cols):
    """The same residual-stopped solves on the photograph-like image, once each, cold start: plain V-cycles (capped at 60) and
    RTDD_METHOD_AUTO -- so that the line shows what the synthetic image hides: where the V-cycle stalls and what the fall-back costs."""
    import torch
    p = photo_problem(rows, cols)
    ctx = rt.Context(int(dev.split(":")[1])); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.GPUAllocateDeviceMemory(rows, cols, 1); ctx.GPULoadWeights(0.4)
    m, g = rt.device_image(p["mask"], dev), rt.device_image(p["gray"], dev)
    out = {"image": f"tests/golden/Dog_full.npz (672 x 624) tiled with mirroring to {cols}x{rows}, scribbles on {float((p['mask'] == 255).mean()) * 100:.1f} % of the pixels"}
    for name, method, cap in (("multigrid_60_cycles", rt.METHOD_MULTIGRID, 60), ("auto", rt.METHOD_AUTO, 400000)):
        d = rt.device_image(p["depth"], dev)
        torch.cuda.synchronize(); t = time.perf_counter()
        its, res = ctx.solve_ex(d, m, g, rows, cols, 0, method=method, maxIterations=cap, tolerance=1e-4)
        ctx.synchronize()
        out[name] = {"ms": (time.perf_counter() - t) * 1e3, "cycles": ctx.last_cycles, "sweeps": its if method == rt.METHOD_AUTO else 0,
                     "residual": float(res), "converged": bool(res <= 1e-4)}
    ctx.close()
    return out


def valu_roofline(px_sweeps_per_s, method):
    ops = VALU_OPS.get(method)
    if ops is None:
        return None
    src = "recorded constant"
    if method == "jacobi":
        ops, src = jacobi_valu_ops()
    achieved = ops * px_sweeps_per_s / 1e12
    r = {"achieved": achieved, "peak": VALU_PEAK_TOPS, "frac": achieved / VALU_PEAK_TOPS, "ops_per_pixel_sweep": ops, "ops_source": src,
         "mix_issue_cycles_measured": VALU_MIX_CYCLES, "frac_of_mix_issue_rate": achieved / VALU_PEAK_TOPS * VALU_MIX_CYCLES / 2.0}
    if method == "jacobi":
        r["frac_at_round4_count"] = VALU_OPS_ROUND4 * px_sweeps_per_s / 1e12 / VALU_PEAK_TOPS      # the same throughput priced at rounds 3-4's 15.1 operations
    return r


def sweep_4k(rt, dev, steps=10, rows=2160, cols=3840, iters=1000, name="4k_jacobi1000"):
    """The north star's 4K sweep (3840x2160 x 1000 Chebyshev-Jacobi sweeps), timed like the headline, outside its timed region; and
    (rows = 4320, cols = 7680, iters = 200) the one Jacobi size whose working set (564 MB at 17 B/px) leaves the Infinity Cache: the
    HBM-resident case of SURVEY 8(d)'s cache caveat."""
    import torch
    from realtimedepthdiffusion_amd.synth import make_problem
    p = make_problem(rows, cols, seed=1234)
    ctx = rt.Context(int(dev.split(":")[1]))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.GPUAllocateDeviceMemory(rows, cols, 1); ctx.GPULoadWeights(0.4)
    m = rt.device_image(p["mask"], dev); g = rt.device_image(p["gray"], dev)
    ds = [rt.device_image(p["depth"], dev) for _ in range(steps + 2)]
    for _ in range(3):                                                     # warm-up (the timed solves below take fresh copies)
        ctx.GPUMatrixFreeSolver(ds[0], m, g, rows, cols, 0.4, iters, 1e-5, 0)
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(steps):
        ctx.GPUMatrixFreeSolver(ds[1 + i], m, g, rows, cols, 0.4, iters, 1e-5, 0)
    ctx.synchronize()
    el = (time.perf_counter() - t) / steps
    # the per-launch duration comes from one more solve with HIP events around every launch (125 launches here: the events cost
    # ~2 us per launch, so that solve is not the one `value` is taken from)
    ctx.profile_enable(True)
    ctx.GPUMatrixFreeSolver(ds[steps + 1], m, g, rows, cols, 0.4, iters, 1e-5, 0)
    ctx.synchronize()
    pr = ctx.profile()
    info = ctx.last_solve_info()
    ctx.close()
    rate = rows * cols * iters / el
    launch_us = pr.sweep_ms * 1e3 / max(pr.launches, 1)
    sweeps_per_launch = pr.sweeps / max(pr.launches, 1)
    hbm_eq = 17.0 * rows * cols * sweeps_per_launch / (launch_us * 1e-6) / 1e9
    res = {"workload": f"{name}: one {cols}x{rows} image, 1 level, {iters} Chebyshev-Jacobi sweeps", "value": rate / 1e6, "unit": "Mpixel-iterations/s",
           "ms_per_step": el * 1e3, "kernel": f"k_sweep_blocked tile {info.tile}, {sweeps_per_launch:g} sweeps per launch, {launch_us:.1f} us per launch",
           "launch_us": launch_us, "sweeps_per_launch": sweeps_per_launch,
           "valu": valu_roofline(rows * cols * sweeps_per_launch / (launch_us * 1e-6), "jacobi"),
           "hbm_equivalent": {"achieved": hbm_eq, "peak": HBM_PEAK_GBS, "frac": hbm_eq / HBM_PEAK_GBS, "unit": "GB/s", "bytes_per_pixel_sweep": 17.0,
                              "algorithmic_bytes_per_launch": 17.0 * rows * cols * sweeps_per_launch}}
    try:                                        # recorded counters of the same kernel instantiation (profiles/, scripts/profile_round.sh), as for the headline
        k = json.load(open(os.path.join(ROOT, "profiles", "counters_latest.json"))).get(name, {})
        if k and k.get("tile") == info.tile and k.get("persistent") == info.persistent and k.get("hbm_bytes_per_launch_corrected"):
            res["traffic"] = k["hbm_bytes_per_launch_corrected"]
            res["hbm_counter_frac"] = k["hbm_bytes_per_launch_corrected"] / (launch_us * 1e-6) / 1e9 / HBM_PEAK_GBS
            res["counters_source"] = k.get("source")
    except (OSError, ValueError, KeyError):
        pass
    return res


def effects(rt, dev):
    """The per-pixel passes of the path (GPUDepthEffect.cu, and the solver's staging passes) against their algorithmic bytes."""
    import numpy as np
    import torch
    from realtimedepthdiffusion_amd.synth import make_problem
    out = {}
    # a REAL depth map for the defocus legs: the library's own estimate of the bundled Dog pair (672 x 624), tiled with mirroring to size
    dog = None
    try:
        g = np.load(os.path.join(ROOT, "tests", "golden", "Dog_full.npz"), allow_pickle=False)
        with rt.Context(int(dev.split(":")[1])) as c0:
            c0.set_stream(torch.cuda.current_stream().cuda_stream); c0.GPULoadWeights(0.4)
            c0.pyramid_create(*g["annotation"].shape)
            c0.pyramid_set_image(rt.device_image(g["bgr"], dev)); c0.pyramid_set_annotation(rt.device_image(g["annotation"], dev))
            c0.estimate_depth(1000); c0.synchronize()
            dog = c0.pyramid_download(rt.IMG_DEPTH, 0)
    except (OSError, KeyError):
        pass

    def tiled(a, rows, cols):
        a2 = np.concatenate([a, a[:, ::-1]], 1); a4 = np.concatenate([a2, a2[::-1]], 0)
        return np.ascontiguousarray(np.tile(a4, (-(-rows // a4.shape[0]), -(-cols // a4.shape[1])))[:rows, :cols])
    for name, rows, cols in (("1080p", 1080, 1920), ("4k", 2160, 3840), ("8k", 4320, 7680)):
        p = make_problem(rows, cols, seed=1)
        rng = np.random.default_rng(0)
        orig = rng.integers(0, 256, (rows, cols, 3), dtype=np.uint8)
        depth = (p["depth"] * rng.uniform(0, 1, (rows, cols))).astype(np.float32)
        c = rt.Context(int(dev.split(":")[1]))
        c.set_stream(torch.cuda.current_stream().cuda_stream)
        c.GPULoadWeights(0.4); c.GPUAllocateDeviceMemory(rows, cols, 1)
        o, g, d, m = (rt.device_image(x, dev) for x in (orig, p["gray"], depth, p["mask"]))
        d_smooth = rt.device_image(p["gray"].astype(np.float32), dev)      # a piecewise-smooth depth map (value noise + rectangles), like a solved one
        art = rt.device_image(np.zeros_like(orig), dev)
        px = rows * cols

        def timeit(f, n=20):
            for _ in range(3):
                f()
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(n):
                f()
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / n
        res = {}
        d_dog = rt.device_image(tiled(dog, rows, cols), dev) if dog is not None else d_smooth
        legs = (("desaturation", 11, lambda: c.GPUSimulateDesaturation(o, g, d, art, rows, cols)),
                             ("haze", 10, lambda: c.GPUSimulateHaze(o, d, art, rows, cols)),
                             ("defocus", 10, lambda: c.GPUSimulateDefocus(o, d_smooth, art, rows, cols)),
                             # a real depth map: the estimate of the bundled Dog pair tiled (mirrored) to this size
                             ("defocus_dataset_depth", 10, lambda: c.GPUSimulateDefocus(o, d_dog, art, rows, cols)),
                             # per-pixel random depth: every window size side by side, no coherence between neighbouring lookups (the worst case)
                             ("defocus_random_depth", 10, lambda: c.GPUSimulateDefocus(o, d, art, rows, cols)),
                             ("prepare_and_finish", 17, lambda: c.GPUMatrixFreeSolver(d, m, g, rows, cols, 0.4, 0, 0, 0)))
        for what, bpp, f in legs:
            if name == "8k" and not what.startswith("defocus"):
                continue                                    # (8K: the defocus legs only -- the one effect whose table leaves the Infinity Cache there)
            t = timeit(f, n=20 if name != "8k" else 8)
            res[what] = {"us": t * 1e6, "algorithmic_GBs": px * bpp / t / 1e9, "bytes_per_pixel": bpp, "frac_of_hbm_peak": px * bpp / t / 1e9 / HBM_PEAK_GBS}
        if name != "1080p":
            res["defocus_table_slices"] = c.get_option(rt.OPT_DEFOCUS_LAST_SLICES)
        out[name] = res
        c.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="1080p_jacobi1000", help="one of %s, or ROWSxCOLSxITERS" % ", ".join(sorted(WORKLOADS)))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-clock-ramp", action="store_true", help="skip the untimed ~0.15 s of solves in front of the warm-up steps")
    ap.add_argument("--no-estimate", action="store_true", help="skip the legs outside the timed region (whole-cascade estimate, 4K sweep, effects): profiling runs")
    ap.add_argument("--sweep-kernel", type=int, default=0)
    ap.add_argument("--temporal-depth", type=int, default=0)
    ap.add_argument("--rows-per-wave", type=int, default=0)
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--persistent", type=int, default=-1)
    ap.add_argument("--method", default=None, choices=["jacobi", "rbgs", "sor_cycles", "multigrid", "auto"],
                    help="override the workload's method; everything but jacobi is an EXTENSION (not the headline)")
    ap.add_argument("--verify", action="store_true", help="after the timed region every rank compares the depth map of each of its images (last step) with the CPU oracle, bit for bit; "
                    "the line then carries `verified` (checker only: outside the timed region)")
    ap.add_argument("--dry-run", action="store_true", help="everything but the GPU work (launcher, rendezvous, sharding, aggregation): the N > 1 path on a CPU-only box")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:        # plain `python3 bench.py --gpus N`: be the launcher
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}, or plainly without WORLD_SIZE set")

    import numpy as np
    import torch
    from realtimedepthdiffusion_amd import shard
    from realtimedepthdiffusion_amd.synth import make_problem

    share_gpu = bool(os.environ.get("RTDD_BENCH_SHARE_GPU"))       # rehearsal on a 1-GPU box: every rank on device 0
    if share_gpu:
        local = 0
    dry = args.dry_run
    if not dry:
        import realtimedepthdiffusion_amd as rt
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback (--dry-run rehearses the multi-rank plumbing only)")
        torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        # The data path has NO collective (independent images); the process group only serves the timing barrier and
        # the MAX/SUM of two scalars.  RCCL ("nccl") is used when it comes up, gloo otherwise -- the result is the same.
        # Every rank joins a gloo group first (it always comes up); whether the timing barrier then runs over RCCL is decided by ALL
        # ranks together (shard.timing_group): a rank-by-rank fallback could leave the ranks on different backends, waiting at the
        # first barrier.  RTDD_BENCH_BACKEND=gloo keeps everything on gloo (default for --dry-run and for ranks sharing a GPU).
        import datetime
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=300))      # (well above the RCCL probe's 60 s: shard.timing_group)
        want_rccl = (os.environ.get("RTDD_BENCH_BACKEND") or ("gloo" if (dry or share_gpu) else "nccl")) == "nccl"
        tgroup, tbackend = shard.timing_group(dist, f"cuda:{local}", want_rccl)
        if want_rccl and tbackend != "nccl" and rank == 0:
            print("[bench] RCCL did not come up on every rank: the timing barrier runs over gloo (the data path has no collective either way)", file=sys.stderr)
    else:
        tgroup, tbackend = None, "none"

    if args.workload not in WORKLOADS:          # ROWSxCOLSxITERS, or batchB_ROWSxCOLSxITERS (a fixed batch of B images dealt round-robin)
        spec = args.workload
        extra = {}
        if spec.startswith("batch"):
            b_, spec = spec[5:].split("_", 1)
            extra["batch"] = int(b_)
        if spec.endswith("_estimate"):          # batchB_ROWSxCOLSxITERS_estimate: the batch as whole estimates (ITERS sweeps at the coarsest level)
            spec = spec[:-len("_estimate")]
            extra["estimate"] = True
        r_, c_, i_ = (int(v) for v in spec.split("x"))
        WORKLOADS[args.workload] = dict(rows=r_, cols=c_, iters=i_, **extra)
    w = WORKLOADS[args.workload]
    rows, cols, iters = w["rows"], w["cols"], w["iters"]
    method = args.method or w.get("method", "jacobi")
    tolerance = w.get("tolerance", 1e-4 if method in ("sor_cycles", "multigrid", "auto") else 0.0)
    batch = w.get("batch", 0)
    # which images this rank owns: a fixed batch is dealt round-robin (strong scaling); otherwise one image per rank (weak)
    my_images = shard.images_for_rank(batch, world, rank) if batch else [rank]
    dev = f"cuda:{local}"
    executed = []                           # iterations actually run per step (residual-stopped methods), and the residual reached
    import math
    levels_ = int(math.log2(max(min(cols, rows) // 45, 1)) + 1)          # src/main.cpp:95
    est_px_iter = sum((rows >> l) * (cols >> l) * int(iters / 2 ** (levels_ - 1 - l)) for l in range(levels_))     # pixel-sweeps of one whole estimate

    if dry:
        def step(i):
            time.sleep(0.001 * len(my_images))

        def fence():
            shard.fence(dist, None, tgroup)
        ctx = None
    else:
        problems = [make_problem(rows, cols, seed=1234 + i) for i in my_images]
        ctx = rt.Context(local)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        if not w.get("estimate"):
            ctx.GPUAllocateDeviceMemory(rows, cols, 1)
        ctx.GPULoadWeights(0.4)
        if args.sweep_kernel: ctx.set_option(rt.OPT_SWEEP_KERNEL, args.sweep_kernel)
        if args.temporal_depth: ctx.set_option(rt.OPT_TEMPORAL_DEPTH, args.temporal_depth)
        if args.rows_per_wave: ctx.set_option(rt.OPT_ROWS_PER_WAVE, args.rows_per_wave)
        if args.tile: ctx.set_option(rt.OPT_TILE, args.tile)
        if args.persistent >= 0: ctx.set_option(rt.OPT_PERSISTENT, args.persistent)
        elif share_gpu and world > 1: ctx.set_option(rt.OPT_PERSISTENT, 0)      # ranks sharing one GPU are not co-resident: no persistent launches
        if w.get("estimate"):
            # this rank's images as ONE batched pyramid; a step = the whole estimate of every one of them (warm-started: the sweep counts are fixed)
            assert ctx.pyramid_create_batch(rows, cols, len(problems)) == levels_
            for b, p in enumerate(problems):
                ctx.pyramid_select(b)
                ctx.pyramid_set_image(rt.device_image(np.repeat(p["gray"][..., None], 3, 2), dev))
                ctx.pyramid_set_annotation(rt.device_image(np.where(p["mask"] == 255, p["edited"][..., 0], 32).astype(np.uint8), dev))
            masks = grays = []; depths = [[] for _ in range(args.steps + args.warmup)]
        else:
            masks = [rt.device_image(p["mask"], dev) for p in problems]; grays = [rt.device_image(p["gray"], dev) for p in problems]
            # one pristine initial-depth image per image and step, uploaded before the clock starts
            depths = [[rt.device_image(p["depth"], dev) for p in problems] for _ in range(args.steps + args.warmup)]

        def step(i):
            if w.get("estimate"):
                ctx.estimate_depth_batch(iters)
                return
            for k in range(len(problems)):
                d, m, g = depths[i][k], masks[k], grays[k]
                if method == "rbgs":
                    ctx.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_RED_BLACK_GS, maxIterations=iters, tolerance=0.0)
                elif method == "sor_cycles":
                    executed.append(ctx.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_RED_BLACK_GS, maxIterations=iters, tolerance=tolerance, relaxation=rt.RELAXATION_AUTO))
                elif method == "multigrid":
                    executed.append(ctx.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_MULTIGRID, maxIterations=iters, tolerance=tolerance))
                elif method == "auto":                  # counted as the level-0 sweeps it ran: 4 per V-cycle + the SOR sweeps behind them
                    its_, res_ = ctx.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_AUTO, maxIterations=iters, tolerance=tolerance)
                    executed.append((its_ + MG_SWEEPS_PER_CYCLE * ctx.last_cycles, res_, ctx.last_cycles))
                else:
                    ctx.GPUMatrixFreeSolver(d, m, g, rows, cols, 0.4, iters, 1e-5, 0)

        def fence():
            shard.fence(dist, torch.cuda.synchronize, tgroup)

    # Untimed, before the W warm-up steps: ~0.15 s of the same solve on a scratch image, so that a GPU that has been idle (a fresh box) has
    # ramped its clocks before anything is measured -- 8 ms of warm-up steps alone leave the first timed steps up to 8 % slow.
    if ctx and not args.no_clock_ramp:
        scratch = [] if w.get("estimate") else [rt.device_image(p["depth"], dev) for p in problems]
        depths.append(scratch)
        t_ramp = time.perf_counter()
        while time.perf_counter() - t_ramp < 0.15:
            step(len(depths) - 1)
            ctx.synchronize()
        depths.pop()
        executed.clear()
    for i in range(args.warmup):
        step(i)
    executed.clear()
    if ctx: ctx.profile_enable(True)        # HIP events around the sweep launches, on the launch stream
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)                  # asynchronous: nothing in the timed loop waits for the GPU
    fence()
    elapsed = time.perf_counter() - t0
    if ctx: ctx.synchronize()                  # also surfaces RTDD_ERR_TIMEOUT of a persistent launch: a number from a failed run is no number
    sweep_ms = 0.0; launches = 0; sweeps = 0; info = None
    if ctx:
        pr = ctx.profile(); sweep_ms = pr.sweep_ms; launches = pr.launches; sweeps = pr.sweeps   # events recorded inside the timed region (last <= 64 solves)
        info = ctx.last_solve_info()
    if executed:                              # residual-stopped: count what actually ran (a V-cycle = its level-0 sweeps)
        per = MG_SWEEPS_PER_CYCLE if method == "multigrid" else 1
        px_iter_per_image = rows * cols * per * sum(e[0] for e in executed) / len(executed)
    else:
        px_iter_per_image = rows * cols * iters
    if w.get("estimate"):
        px_iter_per_image = est_px_iter
    algo_bytes = ALGO_BYTES[method]
    agg_dev = dev if tbackend == "nccl" else "cpu"
    units, elapsed, thr = shard.aggregate(args.steps * len(my_images) * px_iter_per_image, elapsed, dist, agg_dev, tgroup)   # SUM of units, MAX of time
    n_images = batch if batch else world
    value = thr / 1e6
    launch_us = sweep_ms * 1e3 / max(launches, 1)
    if method == "multigrid": sweeps *= MG_SWEEPS_PER_CYCLE
    sweeps_per_launch = sweeps / max(launches, 1)
    kernel_px_sweeps_per_s = rows * cols * sweeps_per_launch / (launch_us * 1e-6) if launch_us > 0 else 0.0
    hbm_eq = algo_bytes * kernel_px_sweeps_per_s / 1e9
    default_line = args.workload == "1080p_jacobi1000" and method == "jacobi"
    out = {
        "metric": "Mpixel-iterations/s (solver sweep)", "value": value, "unit": "Mpixel-iterations/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong" if batch else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic" if not dry else "none (--dry-run: no GPU work)",
        "config": {"workload": (f"{args.workload}: one {cols}x{rows} image per GPU, 1 level, {iters} Chebyshev-Jacobi sweeps (BASELINE configs[1])" if default_line else
                                f"{args.workload}: {batch} independent {cols}x{rows} images, whole coarse-to-fine estimates ({iters} sweeps at the coarsest level, halving per level), a rank's images in the same launches, image i on rank i % {world} (BASELINE configs[3] as estimates)" if w.get("estimate") else
                                f"{args.workload}: {batch} independent {cols}x{rows} images x {iters} Chebyshev-Jacobi sweeps, image i on rank i % {world}, one stream per GPU" + (" (BASELINE configs[3])" if args.workload == "batch64_1080p" else "") if batch else
                                f"{args.workload} ({method})"),
                   "images_per_step": n_images, "images_this_rank": len(my_images), "sweeps_per_launch": sweeps_per_launch,
                   "timing_barrier_backend": tbackend},
    }
    if info is not None:
        out["config"].update({"kernel": info.kernel, "tile": info.tile, "temporal_depth": info.temporal_depth, "persistent": info.persistent, "fp_contract": info.fp_contract})
        kname = {"jacobi": "k_sweep_blocked", "rbgs": "k_rbgs_blocked", "sor_cycles": "k_rbgs_blocked (+ residual checks)",
                 "multigrid": "whole V(2,2) cycle, all levels: 204 B per image pixel and cycle", "auto": "V(2,2) cycles, then k_rbgs_blocked"}[method]
        hbm_equivalent = {"achieved": hbm_eq, "peak": HBM_PEAK_GBS, "frac": hbm_eq / HBM_PEAK_GBS, "unit": "GB/s", "bytes_per_pixel_sweep": algo_bytes,
                          "algorithmic_bytes_per_launch": algo_bytes * rows * cols * sweeps_per_launch,
                          "note": "what the sweeps would move through HBM one launch per sweep; with temporal blocking the tile stays in registers, so this can exceed 1 and is NOT the binding roof"}
        v = valu_roofline(kernel_px_sweeps_per_s, method)
        if v is not None:
            out["roofline"] = {"bound": "valu", "achieved": v["achieved"], "peak": v["peak"], "unit": "Tlane-op/s", "frac": v["frac"], "traffic": None,
                               "kernel": kname, "launch_us": launch_us, "ops_per_pixel_sweep": v["ops_per_pixel_sweep"],
                               "ops_source": v.get("ops_source"), "frac_at_round4_count": v.get("frac_at_round4_count"),
                               "mix_issue_cycles_measured": v["mix_issue_cycles_measured"], "frac_of_mix_issue_rate": v["frac_of_mix_issue_rate"],
                               "definition": "useful pixel-sweeps/s of the kernel x VALU operations per pixel-sweep (static ISA count) / (256 CUs x 4 SIMD-32 x 32 lanes x 2.4 GHz); halo redundancy not credited",
                               "hbm_equivalent": hbm_equivalent}
        else:                                   # the V-cycle is a chain of streaming kernels: HBM is its roof
            out["roofline"] = {"bound": "hbm", "achieved": hbm_eq, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_eq / HBM_PEAK_GBS, "traffic": None, "kernel": kname, "launch_us": launch_us,
                               "algorithmic_bytes_per_launch": algo_bytes * rows * cols * sweeps_per_launch}
        # Counter evidence of the same command, from separate rocprofv3 --pmc passes (scripts/profile_round.sh; FETCH_SIZE x2 + WRITE_SIZE per
        # MI355X_MICROARCH.md section HBM; SQ_INSTS_VALU), committed under profiles/: HBM-side traffic per launch and the counted VALU issue rate.
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "counters_latest.json")))
            k = prof.get(args.workload if not (args.sweep_kernel or args.tile or args.temporal_depth or args.persistent >= 0 or args.method) else "", {})
            # recorded counters describe ONE kernel instantiation: they are attached only if this run launched the same one
            # (tile id and persistence recorded with them by scripts/make_counters_json.py), else the line says why not
            if k and method == "jacobi" and not (k.get("tile") == info.tile and k.get("persistent") == info.persistent):
                out["roofline"]["counters_source"] = f"not attached: {k.get('source')} recorded tile {k.get('tile')} persistent {k.get('persistent')}, this run launched tile {info.tile} persistent {info.persistent}"
                k = {}
            if k:
                if method == "multigrid":       # a chain of kernels: all of them together, per solve (setup + cycles + residual checks)
                    out["roofline"]["traffic"] = k.get("hbm_bytes_per_solve_all_kernels_corrected")
                    out["roofline"]["traffic_is"] = "HBM-side bytes of ALL kernels of one solve"
                    if k.get("kernel_ms_per_solve_clean_trace"):
                        out["roofline"]["hbm_counter_frac"] = k["hbm_bytes_per_solve_all_kernels_corrected"] / (k["kernel_ms_per_solve_clean_trace"] * 1e-3) / 1e9 / HBM_PEAK_GBS
                else:
                    out["roofline"]["traffic"] = k.get("hbm_bytes_per_launch_corrected")
                    if k.get("hbm_bytes_per_launch_corrected") and launch_us > 0:
                        out["roofline"]["hbm_counter_frac"] = k["hbm_bytes_per_launch_corrected"] / (launch_us * 1e-6) / 1e9 / HBM_PEAK_GBS
                if k.get("valu_issue_frac_counted") is not None:
                    out["roofline"]["valu_issue_frac_counted"] = k["valu_issue_frac_counted"]      # SQ_INSTS_VALU x 2 cycles / (duration x 1024 SIMDs x 2.4 GHz): includes halo redundancy
                out["roofline"]["counters_source"] = k.get("source")
        except (OSError, ValueError, KeyError):
            pass
    if w.get("estimate") and not dry:
        out["estimates_per_s"] = batch * args.steps / elapsed if batch else None
        out.pop("roofline", None)              # (a chain of five levels' kernels: the per-kernel roofline is the single-level workloads')
    if args.verify and not dry and w.get("estimate"):
        bad, checked = verify_estimates(rt, ctx, problems, my_images, iters, levels_, dev, world)
        flag = torch.tensor([len(bad), len(checked)], dtype=torch.float64, device=agg_dev)
        if dist is not None:
            dist.all_reduce(flag, op=dist.ReduceOp.SUM, group=tgroup)
        out["verified"] = {"against": "the oracle's cascade (tests/cascade_ref.py over oracle/), bit for bit: every pyramid level's depth image and the u8 map of the sampled images, "
                                      "in the batch the timed steps ran (same images, same per-level kernel choices): the sampled images reset to their cold state, one more batched "
                                      "estimate (cold for them), then one warm-started", "images_differing_all_ranks": int(flag[0].item()), "images_checked_all_ranks": int(flag[1].item()),
                           "rank0_images_checked": checked, "levels": levels_,
                           "rank0_level_choices": {str(l): dict(zip(("tile", "sweeps_per_launch", "persistent", "images_per_launch"), (i.tile, i.temporal_depth, i.persistent, n)))
                                                   for l in range(levels_) for i, n in [ctx.pyramid_level_info(l)]}}
    elif args.verify and not dry:
        # the checker (outside the timed region): each rank's results of the LAST step against the oracle on the host cores
        import hashlib
        import oracle
        lut = oracle.load_weights(0.4)
        bad, shas = [], []
        for k, p in enumerate(problems):
            got = rt.to_host(depths[args.warmup + args.steps - 1][k])
            if method == "jacobi":
                want = oracle.solve(p["depth"].copy(), p["mask"], p["gray"], iters, 0, 0, lut, 1, threads=max(1, oracle.max_threads() // world))
                if not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
                    bad.append(my_images[k])
            shas.append(hashlib.sha256(got.tobytes()).hexdigest()[:16])
        flag = torch.tensor([len(bad)], dtype=torch.float64, device=agg_dev)
        if dist is not None:
            dist.all_reduce(flag, op=dist.ReduceOp.SUM, group=tgroup)
        out["verified"] = {"against": "oracle.solve, bit for bit" if method == "jacobi" else "(hash only: no fixed-count oracle for this method)", "images_differing_all_ranks": int(flag.item()),
                           "rank0_images": [my_images[k] for k in range(len(problems))], "rank0_sha256_16": shas}
    if not dry and rank == 0 and world == 1 and method in ("multigrid", "auto") and not args.no_estimate:
        out["photograph_like"] = photo_record(rt, dev, rows, cols)             # outside the timed region
    if executed:
        out["config"]["converged"] = {"tolerance": tolerance, "iterations": [e[0] for e in executed], "unit": "cycles" if method == "multigrid" else "sweeps",
                                      "residual": max(e[1] for e in executed), "start": "cold (depth 255 + labels)"}
    if ctx and rank == 0 and world == 1 and default_line and not args.no_estimate:          # outside the timed region: the same solve for >= 3 s
        scratch_i = len(depths); depths.append([rt.device_image(p["depth"], dev) for p in problems])
        out["sustained"] = sustained(lambda: step(scratch_i), ctx.synchronize, len(my_images) * px_iter_per_image)
        depths.pop()
    # which numbers of this line were measured by THIS run and which are recordings read from committed files
    out["measured_live"] = ["value", "ms_per_step", "roofline.launch_us (HIP events on the launch stream)", "roofline.achieved / frac (launch_us x the VALU count of the built object)",
                            "sustained", "estimate*", "batch64_1080p_estimate", "sweep_4k / sweep_8k value and launch_us", "effects", "cpu_baseline"]
    out["recorded"] = ["roofline.traffic", "roofline.hbm_counter_frac", "roofline.valu_issue_frac_counted", "sweep_4k.traffic", "sweep_8k.traffic (rocprofv3 --pmc passes of the same command: counters_source)",
                       "roofline.mix_issue_cycles_measured (profiles/r03_block_replay.txt)"]
    if ctx: ctx.close()
    if not dry and rank == 0 and world == 1 and default_line and not args.no_estimate:      # outside the timed region
        c2 = rt.Context(local); c2.set_stream(torch.cuda.current_stream().cuda_stream); c2.GPULoadWeights(0.4)
        out["estimate"] = estimate_ms(rt, c2, problems[0], rows, cols, dev)      # second half of BASELINE's metric
        out["estimate_4k"] = estimate_ms(rt, c2, make_problem(2160, 3840, seed=1234), 2160, 3840, dev, n=10)   # the 6-level 4K cascade (src/main.cpp:95,261-288)
        c2.close()
        out["estimate_dataset"] = estimate_dataset(rt, dev)                        # what real photographs cost (beside the synthetic `estimate`)
        out["dropin_frame"] = dropin_frame(rt, dev, problems[0], rows, cols)     # what an unchanged main.cpp pays: the ten symbols one by one
        out["dropin_frame"]["vs_fused_estimate"] = out["dropin_frame"]["ms"] / out["estimate"]["ms_annotation_changed"]
        out["dropin_frame"]["fused_is"] = "estimate.ms_annotation_changed (rtdd_estimate_depth with the annotation pyramid rebuilt every frame, as main.cpp does)"
        out["sweep_4k"] = sweep_4k(rt, dev)                                      # the north star's 4K stencil sweep
        out["sweep_8k"] = sweep_4k(rt, dev, steps=5, rows=4320, cols=7680, iters=200, name="8k_jacobi200")   # HBM-resident (564 MB working set)
        out["effects"] = effects(rt, dev)
        out["batch64_1080p_estimate"] = batch_estimates(rt, dev)                 # BASELINE configs[3] at N = 1, as whole estimates
    if not dry and rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(rows, cols, method)
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
