"""Red-black SOR followed by a plain Gauss-Seidel polish: does the residual drop under 1e-4 (config 3)?
usage: sor_probe2.py ROWS COLS"""
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
for sched in ([(1.97, 2000)], [(1.97, 4000)], [(1.97, 8000)], [(1.99, 4000), (1.9, 1000)], [(1.97, 4000), (1.8, 500)], [(1.9, 8000)]):
    d = rt.device_image(warm); t = time.perf_counter(); tot = 0
    for om, n in sched:
        its, res = c2.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_RED_BLACK_GS, maxIterations=n, tolerance=1e-30, checkEvery=n, relaxation=om)
        tot += its
        print(sched, "after omega", om, "x", n, "residual", res, flush=True)
    for k in range(40):
        its, res = c2.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_RED_BLACK_GS, maxIterations=20, tolerance=1e-4, checkEvery=20)
        tot += its
        if res <= 1e-4 or k % 5 == 4: print("   polish", (k + 1) * 20, "residual", res, "total sweeps", tot, "elapsed %.3f s" % (time.perf_counter() - t), flush=True)
        if res <= 1e-4: break
