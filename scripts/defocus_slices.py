"""Defocus on the table path by slice budget (RTDD_OPT_DEFOCUS_SLICE_MB; 0 = one whole-image table) at 4K and 8K: a piecewise-smooth depth
map, a real depth map (the library's own estimate of the bundled Dog pair, tiled with mirroring to the size), a random depth per pixel
(the worst case: no coherence between neighbouring lookups) and depth 255 everywhere (the largest windows), microseconds per call."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import realtimedepthdiffusion_amd as rt
from realtimedepthdiffusion_amd.synth import make_problem

budgets = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,32,64,96,128").split(",")]


def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6


def dog_depth():
    g = np.load(os.path.join(ROOT, "tests", "golden", "Dog_full.npz"), allow_pickle=False)
    bgr, ann = g["bgr"], g["annotation"]
    rows, cols = ann.shape
    with rt.Context(0) as c:
        c.GPULoadWeights(0.4); c.pyramid_create(rows, cols)
        c.pyramid_set_image(rt.device_image(bgr)); c.pyramid_set_annotation(rt.device_image(ann))
        c.estimate_depth(1000); c.synchronize()
        return c.pyramid_download(rt.IMG_DEPTH, 0)


def tile(a, rows, cols):
    a2 = np.concatenate([a, a[:, ::-1]], 1); a4 = np.concatenate([a2, a2[::-1]], 0)
    return np.ascontiguousarray(np.tile(a4, (-(-rows // a4.shape[0]), -(-cols // a4.shape[1])))[:rows, :cols])


dog = dog_depth()
for rows, cols in ((2160, 3840), (4320, 7680)):
    p = make_problem(rows, cols, seed=1)
    rng = np.random.default_rng(0)
    orig = rng.integers(0, 256, (rows, cols, 3), dtype=np.uint8)
    rnd = (p["depth"] * rng.uniform(0, 1, (rows, cols))).astype(np.float32)
    c = rt.Context(0)
    o = rt.device_image(orig); art = rt.device_image(np.zeros_like(orig))
    ds = {"smooth": rt.device_image(p["gray"].astype(np.float32)), "dataset depth map": rt.device_image(tile(dog, rows, cols)), "random": rt.device_image(rnd),
          "depth 255": rt.device_image(np.full((rows, cols), 255, np.float32))}
    for mb in budgets:
        c.set_option(rt.OPT_DEFOCUS_SLICE_MB, mb)
        line = f"{cols}x{rows} slice budget {mb:3d} MB:"
        for name, d in ds.items():
            t = timeit(lambda d=d: c.GPUSimulateDefocus(o, d, art, rows, cols))
            line += f"  {name} {t:7.1f} us"
        print(line + f"  ({c.get_option(rt.OPT_DEFOCUS_LAST_SLICES)} slice(s))", flush=True)
    c.close()
