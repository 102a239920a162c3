"""Prototype: vector extrapolation between V-cycles.  When the residual ratio of two consecutive cycles has settled at
lambda, the error is dominated by one slowly decaying family, e_k ~ lambda e_{k-1}, and
x* ~ x_k + lambda/(1-lambda) (x_k - x_{k-1}) removes it.  usage: mg_extrapolation_probe.py ROWS COLS [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import realtimedepthdiffusion_amd as rt
from realtimedepthdiffusion_amd.synth import make_problem
rows, cols = int(sys.argv[1]), int(sys.argv[2]); seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1234
p = make_problem(rows, cols, seed=seed)
c = rt.Context(0); c.GPULoadWeights(0.4); c.GPUAllocateDeviceMemory(rows, cols, 1)
m = rt.device_image(p["mask"]); g = rt.device_image(p["gray"])
for extrapolate in (False, True):
    d = rt.device_image(p["depth"]); hist = []; prev_x = None; wait = 0; out = []
    for k in range(40):
        x_before = d.clone()
        its, res = c.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_MULTIGRID, maxIterations=1, tolerance=1e-30)
        hist.append(res); out.append("%.1e" % res)
        if res <= 1e-4: break
        wait += 1
        if extrapolate and wait >= 3 and len(hist) >= 3:
            l1, l0 = hist[-1] / hist[-2], hist[-2] / hist[-3]
            if 0.3 < l1 < 0.995 and abs(l1 - l0) <= 0.05 * l1:
                a = l1 / (1 - l1)
                d += a * (d - x_before)              # x_before = iterate before this cycle = x_{k-1}
                torch.clamp_(d, 0, 255)
                out[-1] += "*"; wait = 0
    print("extrapolation" if extrapolate else "plain        ", rows, cols, seed, "cycles", len(hist), " ".join(out), flush=True)
