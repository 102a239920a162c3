"""RTDD_METHOD_AUTO from the cold start to a 1e-4 residual, beside SOR cycles alone.  usage: auto_bench.py ROWS COLS [seed]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import realtimedepthdiffusion_amd as rt
from realtimedepthdiffusion_amd.synth import make_problem
rows, cols = int(sys.argv[1]), int(sys.argv[2]); seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1234
p = make_problem(rows, cols, seed=seed)
c = rt.Context(0); c.GPULoadWeights(0.4); c.GPUAllocateDeviceMemory(rows, cols, 1)
m = rt.device_image(p["mask"]); g = rt.device_image(p["gray"])
for name, kw in (("auto", dict(method=rt.METHOD_AUTO)), ("sor cycles", dict(method=rt.METHOD_RED_BLACK_GS, relaxation=rt.RELAXATION_AUTO))):
    for rep in range(2):
        d = rt.device_image(p["depth"]); torch.cuda.synchronize(); t = time.perf_counter()
        its, res = c.solve_ex(d, m, g, rows, cols, 0, maxIterations=400000, tolerance=1e-4, **kw); c.synchronize()
        ms = (time.perf_counter() - t) * 1e3
    print(json.dumps(dict(method=name, rows=rows, cols=cols, seed=seed, cycles=c.last_cycles, sweeps=its, residual=res, ms=ms)), flush=True)
