"""rocprofv3 target: hierarchy setup + N V-cycles at ROWS x COLS (default 8K), optional residual checks.
usage: mg_profile.py [ROWS COLS [CYCLES [CHECK]]]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import realtimedepthdiffusion_amd as rt
from realtimedepthdiffusion_amd.synth import make_problem
rows, cols = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4320, 7680)
cycles = int(sys.argv[3]) if len(sys.argv) > 3 else 10
check = int(sys.argv[4]) if len(sys.argv) > 4 else 0
p = make_problem(rows, cols, seed=1234)
c = rt.Context(0); c.GPULoadWeights(0.4); c.GPUAllocateDeviceMemory(rows, cols, 1)
m = rt.device_image(p["mask"]); g = rt.device_image(p["gray"])
for _ in range(2):
    d = rt.device_image(p["depth"])
    its, res = c.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_MULTIGRID, maxIterations=cycles, tolerance=1e-30 if check else 0.0, checkEvery=1)
    c.synchronize()
print(its, res)
