"""Defocus by path (RTDD_OPT_DEFOCUS_PATH) and size: smooth and random depth, microseconds per call.  usage: defocus_paths.py [paths, e.g. 0,1,2]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import realtimedepthdiffusion_amd as rt
from realtimedepthdiffusion_amd.synth import make_problem

paths = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")]
def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6

for rows, cols in ((624, 672), (853, 1280), (1080, 1920), (1440, 2560), (2160, 3840), (4320, 7680)):
    p = make_problem(rows, cols, seed=1)
    rng = np.random.default_rng(0)
    orig = rng.integers(0, 256, (rows, cols, 3), dtype=np.uint8)
    rnd = (p["depth"] * rng.uniform(0, 1, (rows, cols))).astype(np.float32)
    c = rt.Context(0)
    o, d_rnd, d_smooth = rt.device_image(orig), rt.device_image(rnd), rt.device_image(p["gray"].astype(np.float32))
    d_255 = rt.device_image(np.full((rows, cols), 255, np.float32))
    art = rt.device_image(np.zeros_like(orig))
    line = f"{cols}x{rows}:"
    for path in paths:
        try:
            c.set_option(rt.OPT_DEFOCUS_PATH, path)
        except rt.RtddError:
            continue
        t = [timeit(lambda d=d: c.GPUSimulateDefocus(o, d, art, rows, cols)) for d in (d_smooth, d_rnd, d_255)]
        line += f"  path {path}: smooth {t[0]:7.1f}  random {t[1]:7.1f}  depth-255 {t[2]:7.1f} us |"
    print(line, flush=True)
    c.close()
