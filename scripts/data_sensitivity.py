import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import realtimedepthdiffusion_amd as rt
from realtimedepthdiffusion_amd.synth import make_problem
rows, cols, iters = 1080, 1920, 1000
p = make_problem(rows, cols, seed=1234)
rng = np.random.default_rng(0)
variants = {
  "synthetic problem (bench)": (p["depth"], p["mask"], p["gray"]),
  "random depth, same mask/gray": (rng.uniform(1, 255, (rows, cols)).astype(np.float32), p["mask"], p["gray"]),
  "synthetic depth, no Dirichlet": (p["depth"], np.full_like(p["mask"], 32), p["gray"]),
  "labels never 0 (min 64)": (np.maximum(p["depth"], 64), p["mask"], p["gray"]),
  "flat gray": (p["depth"], p["mask"], np.full_like(p["gray"], 100)),
}
c = rt.Context(0); c.GPULoadWeights(0.4); c.GPUAllocateDeviceMemory(rows, cols, 1)
for name, (d, m, g) in variants.items():
    md, gd = rt.device_image(m), rt.device_image(g)
    ds = [rt.device_image(d) for _ in range(6)]
    for i in range(2): c.GPUMatrixFreeSolver(ds[i], md, gd, rows, cols, 0.4, iters, 0, 0)
    c.synchronize(); t = time.perf_counter()
    for i in range(2, 6): c.GPUMatrixFreeSolver(ds[i], md, gd, rows, cols, 0.4, iters, 0, 0)
    c.synchronize(); el = (time.perf_counter() - t) / 4
    out = rt.to_host(ds[5])
    print(f"{name:34s} {el*1e3:.3f} ms/solve  {rows*cols*iters/el/1e9:.0f} Gpx-it/s   min|x|>0: {np.abs(out[out!=0]).min():.3e}  zeros: {(out==0).mean():.3f}")
