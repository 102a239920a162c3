"""Wall time of GPUSimulateDefocus by path (RTDD_OPT_DEFOCUS_PATH 1 = global table, 2 = per-tile tables in LDS), for a piecewise-smooth
and a per-pixel random depth map, at the sizes the tile kernel takes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import realtimedepthdiffusion_amd as rt
from realtimedepthdiffusion_amd.synth import make_problem

def timeit(f, n=200):
    for _ in range(5): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n

for rows, cols in ((1080, 1920), (853, 1280), (624, 672), (1440, 1754), (270, 480)):
    p = make_problem(rows, cols, seed=1)
    rng = np.random.default_rng(0)
    orig = rng.integers(0, 256, (rows, cols, 3), dtype=np.uint8)
    c = rt.Context(0)
    o = rt.device_image(orig)
    art = rt.device_image(np.zeros_like(orig))
    depths = {"smooth": p["gray"].astype(np.float32), "random": (p["depth"] * rng.uniform(0, 1, (rows, cols))).astype(np.float32)}
    line = f"{cols}x{rows}:"
    for name, dm in depths.items():
        d = rt.device_image(dm)
        for path in (1, 2):
            c.set_option(rt.OPT_DEFOCUS_PATH, path)
            t = timeit(lambda: c.GPUSimulateDefocus(o, d, art, rows, cols))
            line += f"  {name} path {path}: {t*1e6:6.1f} us"
    print(line)
    c.close()
