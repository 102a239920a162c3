import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import realtimedepthdiffusion_amd as rt
from realtimedepthdiffusion_amd.synth import make_problem
rows, cols = int(sys.argv[1]), int(sys.argv[2])
p = make_problem(rows, cols, seed=1234)
bgr = np.repeat(p["gray"][..., None], 3, 2); ann = np.where(p["mask"] == 255, p["edited"][..., 0], 32).astype(np.uint8)
c = rt.Context(0); c.GPULoadWeights(0.4)
P = c.pyramid_create(rows, cols); c.pyramid_set_image(rt.device_image(bgr)); c.pyramid_set_annotation(rt.device_image(ann))
c.estimate_depth(1000); c.synchronize()
warm = c.pyramid_download(rt.IMG_DEPTH, 0)
c2 = rt.Context(0); c2.GPULoadWeights(0.4); c2.GPUAllocateDeviceMemory(rows, cols, 1)
m = rt.device_image(p["mask"]); g = rt.device_image(p["gray"])
for method, name, tol in ((rt.METHOD_RED_BLACK_GS, "rbgs", 1e-4), (rt.METHOD_CHEBYSHEV_JACOBI, "cheby", 3e-4)):
    d = rt.device_image(warm); tot = 0; t = time.perf_counter()
    for chunk in range(60):
        its, res = c2.solve_ex(d, m, g, rows, cols, 0, method=method, maxIterations=500, tolerance=tol, checkEvery=100)
        tot += its
        if chunk % 4 == 0 or res <= tol: print(name, rows, cols, "sweeps", tot, "residual", res, "elapsed %.2f s" % (time.perf_counter() - t), flush=True)
        if res <= tol: break
