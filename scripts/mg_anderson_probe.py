"""Prototype: Anderson acceleration of the V-cycle map (depth m), in torch on the device.  Does it rescue the instances the
plain V-cycle stalls on?  usage: mg_anderson_probe.py ROWS COLS seed m [golden name]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import realtimedepthdiffusion_amd as rt
from realtimedepthdiffusion_amd.synth import make_problem
rows, cols, seed, m_depth = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
p = make_problem(rows, cols, seed=seed)
if len(sys.argv) > 5:
    g_ = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", sys.argv[5] + "_256.npz"))
    p = {"gray": g_["gray0"], "mask": g_["mask0"], "depth": g_["depth_before_c1_L0"]}; rows, cols = 256, 256
c = rt.Context(0); c.GPULoadWeights(0.4); c.GPUAllocateDeviceMemory(rows, cols, 1)
m = rt.device_image(p["mask"]); g = rt.device_image(p["gray"])
def G(x):
    y = x.clone(); c.solve_ex(y, m, g, rows, cols, 0, method=rt.METHOD_MULTIGRID, maxIterations=1, tolerance=0.0); return y
def res(x):
    y = x.clone(); _, r = c.solve_ex(y, m, g, rows, cols, 0, method=rt.METHOD_RED_BLACK_GS, maxIterations=0, tolerance=1e-30, checkEvery=1)
    return r
for use_aa in (False, True):
    x = rt.device_image(p["depth"]); dX, dF = [], []; xo = fo = None; out = []
    for k in range(60):
        gx = G(x); f = gx - x
        r = c.solve_ex(gx.clone(), m, g, rows, cols, 0, method=rt.METHOD_RED_BLACK_GS, maxIterations=1, tolerance=1e-30, checkEvery=1)[1]
        out.append("%.1e" % r)
        if r <= 1e-4: break
        if use_aa and xo is not None:
            dX.append((x - xo).flatten().double()); dF.append((f - fo).flatten().double())
            dX, dF = dX[-m_depth:], dF[-m_depth:]
        xo, fo = x.clone(), f.clone()
        if use_aa and dF:
            Fm = torch.stack(dF, 1); Xm = torch.stack(dX, 1)
            gam = torch.linalg.lstsq(Fm, f.flatten().double().unsqueeze(1)).solution
            x = (x.flatten().double() + f.flatten().double() - ((Xm + Fm) @ gam).squeeze(1)).float().reshape(x.shape).clamp(0, 255)
        else:
            x = gx
    print("anderson(%d)" % m_depth if use_aa else "plain      ", rows, cols, seed, "cycles", len(out), " ".join(out[::2]), flush=True)
