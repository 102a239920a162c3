"""Time the per-pixel passes (edge-weight/staging pass, annotation kernels, depth effects) against their algorithmic bytes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import realtimedepthdiffusion_amd as rt
from realtimedepthdiffusion_amd.synth import make_problem

def timeit(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n

for rows, cols in ((1080, 1920), (2160, 3840), (4320, 7680)):
    p = make_problem(rows, cols, seed=1)
    rng = np.random.default_rng(0)
    orig = rng.integers(0, 256, (rows, cols, 3), dtype=np.uint8)
    depth = (p["depth"] * rng.uniform(0, 1, (rows, cols))).astype(np.float32)
    c = rt.Context(0); c.GPULoadWeights(0.4); c.GPUAllocateDeviceMemory(rows, cols, 1)
    o, g, d, m, e = (rt.device_image(x) for x in (orig, p["gray"], depth, p["mask"], p["edited"]))
    art = rt.device_image(np.zeros_like(orig))
    px = rows * cols
    res = []
    res.append(("desaturation", 11, timeit(lambda: c.GPUSimulateDesaturation(o, g, d, art, rows, cols))))
    res.append(("haze", 10, timeit(lambda: c.GPUSimulateHaze(o, d, art, rows, cols))))
    res.append(("defocus (SAT build + lookup)", 10, timeit(lambda: c.GPUSimulateDefocus(o, d, art, rows, cols))))
    res.append(("convert_to_float", 1.4, timeit(lambda: c.GPUConvertToFloat(e, d, m, rows, cols))))
    res.append(("solver with 0 sweeps (prepare+finish)", 9 + 8, timeit(lambda: c.GPUMatrixFreeSolver(d, m, g, rows, cols, 0.4, 0, 0, 0))))
    print(f"--- {cols}x{rows}")
    for name, bpp, t in res:
        print(f"  {name:40s} {t*1e6:9.1f} us   {px*bpp/t/1e9:8.1f} GB/s algorithmic ({bpp} B/px)   frac of 8 TB/s {px*bpp/t/8e12:.3f}")
    c.close()
