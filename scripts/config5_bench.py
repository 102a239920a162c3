"""BASELINE config 5: one 7680x4320 image, multigrid V-cycles on one MI355X.
Cascade warm start (rtdd_estimate_depth) -> rtdd_solve_ex(RTDD_METHOD_MULTIGRID, tolerance 1e-4) on level 0; the same from
the cold start (depth 255 + labels); red-black SOR cycles beside it.
usage: config5_bench.py [ROWS COLS [REPEATS]]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import realtimedepthdiffusion_amd as rt
from realtimedepthdiffusion_amd.synth import make_problem
rows, cols = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4320, 7680)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
p = make_problem(rows, cols, seed=1234)
bgr = np.repeat(p["gray"][..., None], 3, 2); ann = np.where(p["mask"] == 255, p["edited"][..., 0], 32).astype(np.uint8)
c = rt.Context(0); c.GPULoadWeights(0.4)
c.pyramid_create(rows, cols); c.pyramid_set_image(rt.device_image(bgr)); c.pyramid_set_annotation(rt.device_image(ann))
c.estimate_depth(1000); c.synchronize()
warm = c.pyramid_download(rt.IMG_DEPTH, 0)
c2 = rt.Context(0); c2.GPULoadWeights(0.4); c2.GPUAllocateDeviceMemory(rows, cols, 1)
m = rt.device_image(p["mask"]); g = rt.device_image(p["gray"])
for start, img in (("cascade", warm), ("cold", p["depth"])):
    for r in range(reps):
        d = rt.device_image(img); torch.cuda.synchronize(); t0 = time.perf_counter()
        its, res = c2.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_MULTIGRID, maxIterations=200, tolerance=1e-4)
        c2.synchronize(); t1 = time.perf_counter()
        print(json.dumps(dict(method="multigrid", start=start, rows=rows, cols=cols, cycles=its, residual=res, ms=(t1 - t0) * 1e3, ms_per_cycle=(t1 - t0) * 1e3 / max(its, 1))), flush=True)
# fixed number of cycles, no residual checks: the pure cycle time
d = rt.device_image(warm); torch.cuda.synchronize(); t0 = time.perf_counter()
its, res = c2.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_MULTIGRID, maxIterations=10, tolerance=0.0)
c2.synchronize(); t1 = time.perf_counter()
print(json.dumps(dict(method="multigrid", note="10 cycles incl. hierarchy setup, no residual checks", ms=(t1 - t0) * 1e3)), flush=True)
d = rt.device_image(warm); torch.cuda.synchronize(); t0 = time.perf_counter()
its, res = c2.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_MULTIGRID, maxIterations=1, tolerance=0.0)
c2.synchronize(); t1 = time.perf_counter()
print(json.dumps(dict(method="multigrid", note="1 cycle incl. hierarchy setup", ms=(t1 - t0) * 1e3)), flush=True)
if os.environ.get("WITH_SOR", "1") == "1":
    d = rt.device_image(warm); torch.cuda.synchronize(); t0 = time.perf_counter()
    its, res = c2.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_RED_BLACK_GS, maxIterations=400000, tolerance=1e-4, relaxation=rt.RELAXATION_AUTO)
    c2.synchronize(); t1 = time.perf_counter()
    print(json.dumps(dict(method="sor cycles", start="cascade", sweeps=its, residual=res, ms=(t1 - t0) * 1e3)), flush=True)
