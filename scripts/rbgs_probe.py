import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import realtimedepthdiffusion_amd as rt
from realtimedepthdiffusion_amd.synth import make_problem
rows, cols = int(sys.argv[1]), int(sys.argv[2])
p = make_problem(rows, cols, seed=1234)
c = rt.Context(0); c.GPULoadWeights(0.4); c.GPUAllocateDeviceMemory(rows, cols, 1)
m = rt.device_image(p["mask"]); g = rt.device_image(p["gray"])
for method, name in ((rt.METHOD_RED_BLACK_GS, "rbgs"), (rt.METHOD_CHEBYSHEV_JACOBI, "cheby")):
    d = rt.device_image(p["depth"])
    tot = 0
    t = time.perf_counter()
    for chunk in range(40):
        its, res = c.solve_ex(d, m, g, rows, cols, 0, method=method, maxIterations=2000, tolerance=1e-4 if method == rt.METHOD_RED_BLACK_GS else 3e-4, checkEvery=500)
        tot += its
        print(name, rows, cols, "sweeps", tot, "residual", res, "elapsed %.2f s" % (time.perf_counter() - t), flush=True)
        if res <= (1e-4 if method == rt.METHOD_RED_BLACK_GS else 3e-4): break
