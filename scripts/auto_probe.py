"""After stalled V-cycles, how many SOR sweeps does the rest need?  usage: auto_probe.py ROWS COLS [seed]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import realtimedepthdiffusion_amd as rt
from realtimedepthdiffusion_amd.synth import make_problem
rows, cols = int(sys.argv[1]), int(sys.argv[2]); seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1234
p = make_problem(rows, cols, seed=seed)
c = rt.Context(0); c.GPULoadWeights(0.4); c.GPUAllocateDeviceMemory(rows, cols, 1)
m = rt.device_image(p["mask"]); g = rt.device_image(p["gray"])
N = max(rows, cols)
for ncyc in (0, 4, 8):
    for frac in (16, 8, 4, 2, 1):
        d = rt.device_image(p["depth"]); res0 = float("nan")
        if ncyc: _, res0 = c.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_MULTIGRID, maxIterations=ncyc, tolerance=1e-30, checkEvery=ncyc)
        n = N // frac
        c.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_RED_BLACK_GS, maxIterations=n, tolerance=0.0, relaxation=1.99)
        c.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_RED_BLACK_GS, maxIterations=n // 4, tolerance=0.0, relaxation=1.9)
        its, res = c.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_RED_BLACK_GS, maxIterations=100, tolerance=1e-4, checkEvery=20)
        print(f"{ncyc} V-cycles (residual {res0:.2e}) + {n} sweeps at 1.99 + {n//4} at 1.9 + {its} polish -> residual {res:.2e}", flush=True)
