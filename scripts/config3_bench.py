"""BASELINE config 3: one 3840x2160 image, red-black (S)OR to a 1e-4 residual on one MI355X.
Cascade warm start (rtdd_estimate_depth) -> rtdd_solve_ex(RED_BLACK_GS, RTDD_RELAXATION_AUTO, tolerance 1e-4) on level 0.
usage: config3_bench.py [ROWS COLS [REPEATS]]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import realtimedepthdiffusion_amd as rt
from realtimedepthdiffusion_amd.synth import make_problem
rows, cols = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2160, 3840)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
p = make_problem(rows, cols, seed=1234)
bgr = np.repeat(p["gray"][..., None], 3, 2); ann = np.where(p["mask"] == 255, p["edited"][..., 0], 32).astype(np.uint8)
c = rt.Context(0); c.GPULoadWeights(0.4)
c.pyramid_create(rows, cols); c.pyramid_set_image(rt.device_image(bgr)); c.pyramid_set_annotation(rt.device_image(ann))
c2 = rt.Context(0); c2.GPULoadWeights(0.4); c2.GPUAllocateDeviceMemory(rows, cols, 1)
m = rt.device_image(p["mask"]); g = rt.device_image(p["gray"])
out = []
for r in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    c.estimate_depth(1000); c.synchronize()
    t1 = time.perf_counter()
    d = rt.device_image(c.pyramid_download(rt.IMG_DEPTH, 0)); torch.cuda.synchronize()
    t2 = time.perf_counter()
    its, res = c2.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_RED_BLACK_GS, maxIterations=200000, tolerance=1e-4, relaxation=rt.RELAXATION_AUTO)
    c2.synchronize(); t3 = time.perf_counter()
    out.append(dict(rows=rows, cols=cols, cascade_ms=(t1 - t0) * 1e3, sor_sweeps=its, residual=res, sor_ms=(t3 - t2) * 1e3,
                    gpx_sweeps_per_s=rows * cols * its / (t3 - t2) / 1e9))
    print(json.dumps(out[-1]), flush=True)
