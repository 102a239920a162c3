"""Prototype: extrapolation inside the Gauss-Seidel polish of the SOR cycles.  If the first cycle's polish plateaus on one slowly
decaying family (residual ratio between 20-sweep blocks settled), x + l/(1-l) (x - x_20_sweeps_ago) may save the doubled cycle.
usage: sor_extrapolation_probe.py ROWS COLS seed"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import realtimedepthdiffusion_amd as rt
from realtimedepthdiffusion_amd.synth import make_problem
rows, cols, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
p = make_problem(rows, cols, seed=seed)
c = rt.Context(0); c.GPULoadWeights(0.4); c.GPUAllocateDeviceMemory(rows, cols, 1)
m = rt.device_image(p["mask"]); g = rt.device_image(p["gray"])
N = max(rows, cols)
w0 = min(1.99, max(1.0, 2.0 / (1.0 + math.sin(4.0 * math.pi / N))))
def sweeps(d, n, om, tol=0.0, chk=0):
    return c.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_RED_BLACK_GS, maxIterations=n, tolerance=tol, checkEvery=chk or n, relaxation=om)
for extrap in (False, True):
    d = rt.device_image(p["depth"]); total = 0; log = []
    for cycle in range(6):
        e = min(cycle, 6); gap = max(0.005, (2.0 - w0) / (1 << e)); n_hi = N << e
        sweeps(d, n_hi, 2.0 - gap); sweeps(d, n_hi // 4, max(1.0, 2.0 - 10 * gap)); total += n_hi + n_hi // 4
        hist = []; ok = False
        for k in range(5 if not extrap else 8):
            before = d.clone()
            _, res = sweeps(d, 20, 1.0, 1e-30, 20); total += 20; hist.append(res)
            if res <= 1e-4: ok = True; break
            if extrap and len(hist) >= 3:
                l1, l0 = hist[-1] / hist[-2], hist[-2] / hist[-3]
                if 0.3 < l1 < 0.995 and abs(l1 - l0) <= 0.08 * l1:
                    d += (l1 / (1 - l1)) * (d - before); torch.clamp_(d, 0, 255); hist = []
                    log.append("x")
        log.append("cycle %d: %s" % (cycle, " ".join("%.1e" % h for h in hist)))
        if ok: break
    print("extrapolation" if extrap else "plain        ", rows, cols, seed, "total sweeps", total, "|", " | ".join(log), flush=True)
