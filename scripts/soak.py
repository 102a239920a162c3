"""Soak: N back-to-back persistent 1080p solves (and a mix of the other methods); every result must equal the first run's
bits and rtdd_ctx_synchronize must never report a timeout.  usage: soak.py [N]"""
import sys, os, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import realtimedepthdiffusion_amd as rt
from realtimedepthdiffusion_amd.synth import make_problem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rows, cols = 1080, 1920
p = make_problem(rows, cols, seed=1234)
c = rt.Context(0); c.GPULoadWeights(0.4); c.GPUAllocateDeviceMemory(rows, cols, 1)
m = rt.device_image(p["mask"]); g = rt.device_image(p["gray"]); src = rt.device_image(p["depth"])
ref = {}
t = time.time()
for i in range(n):
    kind = ("jacobi", "jacobi", "jacobi", "sor", "mg")[i % 5]
    d = src.clone()
    if kind == "jacobi": c.GPUMatrixFreeSolver(d, m, g, rows, cols, 0.4, 1000, 1e-5, 0)
    elif kind == "sor": c.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_RED_BLACK_GS, maxIterations=200, relaxation=1.9)
    else: c.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_MULTIGRID, maxIterations=3)
    c.synchronize()                                   # raises on RTDD_ERR_TIMEOUT
    h = hashlib.sha1(d.cpu().numpy().tobytes()).hexdigest()
    assert ref.setdefault(kind, h) == h, (i, kind)
    if i % 500 == 499: print(i + 1, "solves ok, %.1f s" % (time.time() - t), flush=True)
print("soak ok:", n, "solves,", {k: v[:12] for k, v in ref.items()})
