import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import realtimedepthdiffusion_amd as rt
from realtimedepthdiffusion_amd.synth import make_problem
rows, cols = int(sys.argv[1]), int(sys.argv[2])
p = make_problem(rows, cols, seed=1234)
c = rt.Context(0); c.GPULoadWeights(0.4); c.GPUAllocateDeviceMemory(rows, cols, 1)
m = rt.device_image(p["mask"]); g = rt.device_image(p["gray"]); d = rt.device_image(p["depth"])
prev = None
for k in range(22):
    its, res = c.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_MULTIGRID, maxIterations=1, tolerance=1e-30)
    print(k + 1, "%.3e" % res, "" if prev is None else "%.2f" % (res / prev)); prev = res
