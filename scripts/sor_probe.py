"""How many red-black SOR sweeps does BASELINE config 3 (4K, residual 1e-4) take, warm-started from the cascade?
usage: sor_probe.py ROWS COLS [omega ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import realtimedepthdiffusion_amd as rt
from realtimedepthdiffusion_amd.synth import make_problem
rows, cols = int(sys.argv[1]), int(sys.argv[2])
omegas = [float(v) for v in sys.argv[3:]] or [1.0, 1.9, 1.97, 1.99]
p = make_problem(rows, cols, seed=1234)
bgr = np.repeat(p["gray"][..., None], 3, 2); ann = np.where(p["mask"] == 255, p["edited"][..., 0], 32).astype(np.uint8)
c = rt.Context(0); c.GPULoadWeights(0.4)
P = c.pyramid_create(rows, cols); c.pyramid_set_image(rt.device_image(bgr)); c.pyramid_set_annotation(rt.device_image(ann))
c.estimate_depth(1000); c.synchronize()
warm = c.pyramid_download(rt.IMG_DEPTH, 0)
c2 = rt.Context(0); c2.GPULoadWeights(0.4); c2.GPUAllocateDeviceMemory(rows, cols, 1)
m = rt.device_image(p["mask"]); g = rt.device_image(p["gray"])
for om in omegas:
    d = rt.device_image(warm); tot = 0; t = time.perf_counter()
    for chunk in range(40):
        its, res = c2.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_RED_BLACK_GS, maxIterations=1000, tolerance=1e-4, checkEvery=200, relaxation=om)
        tot += its
        if chunk % 4 == 0 or res <= 1e-4: print("omega", om, rows, cols, "sweeps", tot, "residual", res, "elapsed %.3f s" % (time.perf_counter() - t), flush=True)
        if res <= 1e-4: break
