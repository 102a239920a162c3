#!/usr/bin/env python3
"""What a synchronising call costs the drop-in shim (GPUMatrixFreeSolver returns synchronised at every pyramid level, src/GPUSolver.cu:314):
host time of rtdd_ctx_synchronize behind a short solve, with and without a status word to read, and of the pieces beside it."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import realtimedepthdiffusion_amd as rt
from realtimedepthdiffusion_amd.synth import make_problem

rows, cols = 67, 120
p = make_problem(rows, cols, seed=1)
for stream_kind, spin in (("null", 0), ("own", 0)):
    c = rt.Context(0)
    if stream_kind == "own":
        c.set_stream(torch.cuda.Stream().cuda_stream)
    c.GPUAllocateDeviceMemory(rows, cols, 1); c.GPULoadWeights(0.4)
    d, m, g = (rt.device_image(p[k], "cuda:0") for k in ("depth", "mask", "gray"))
    for iters in (0, 8, 200):
        for _ in range(20):
            c.GPUMatrixFreeSolver(d, m, g, rows, cols, 0.4, iters, 1e-5, 0); c.synchronize()
        n = 300
        t = time.perf_counter()
        for _ in range(n):
            c.GPUMatrixFreeSolver(d, m, g, rows, cols, 0.4, iters, 1e-5, 0)
        c.synchronize()
        t_async = (time.perf_counter() - t) / n
        t = time.perf_counter()
        for _ in range(n):
            c.GPUMatrixFreeSolver(d, m, g, rows, cols, 0.4, iters, 1e-5, 0); c.synchronize()
        t_sync = (time.perf_counter() - t) / n
        print(f"{stream_kind} stream, {cols}x{rows} x {iters} sweeps: back to back {t_async * 1e6:.1f} us per solve, each followed by rtdd_ctx_synchronize {t_sync * 1e6:.1f} us "
              f"(+{(t_sync - t_async) * 1e6:.1f})")
    t = time.perf_counter()
    for _ in range(1000):
        c.synchronize()
    print(f"{stream_kind} stream: rtdd_ctx_synchronize on an idle stream {(time.perf_counter() - t) / 1000 * 1e6:.2f} us")
    c.close()
