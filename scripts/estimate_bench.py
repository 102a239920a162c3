"""Wall time of one whole depth estimate (src/main.cpp:232-295 sequence) and of each pyramid level's solve."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import realtimedepthdiffusion_amd as rt
from realtimedepthdiffusion_amd.synth import make_problem

rows, cols = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
p = make_problem(rows, cols, seed=1234)
bgr = np.repeat(p["gray"][..., None], 3, 2)
ann = np.where(p["mask"] == 255, p["edited"][..., 0], 32).astype(np.uint8)
c = rt.Context(0); c.GPULoadWeights(0.4)
P = c.pyramid_create(rows, cols)
c.pyramid_set_image(rt.device_image(bgr)); c.pyramid_set_annotation(rt.device_image(ann))
for _ in range(3): c.estimate_depth(1000)
c.synchronize()
n = 20
t = time.perf_counter()
for _ in range(n): c.estimate_depth(1000)
c.synchronize()
el = (time.perf_counter() - t) / n
pxit = sum((rows >> l) * (cols >> l) * int(1000 / 2 ** (P - 1 - l)) for l in range(P))
print(f"{cols}x{rows}: P={P} estimate {el*1e3:.3f} ms  ({pxit/1e6:.1f} Mpx-it -> {pxit/el/1e9:.1f} Gpx-it/s)")
# per-level solve times
for l in range(P - 1, -1, -1):
    r, cc = rows >> l, cols >> l
    it = int(1000 / 2 ** (P - 1 - l))
    dptr = c.pyramid_image(rt.IMG_DEPTH, l); sptr = c.pyramid_image(rt.IMG_SCRIBBLE, l); gptr = c.pyramid_image(rt.IMG_GRAY, l)
    args = ((dptr[0], dptr[1]), (sptr[0], sptr[1]), (gptr[0], gptr[1]), dptr[2], dptr[3], 0.4, it, 1e-5, l)
    for _ in range(2): c.GPUMatrixFreeSolver(*args)
    c.synchronize(); t = time.perf_counter()
    for _ in range(10): c.GPUMatrixFreeSolver(*args)
    c.synchronize(); e = (time.perf_counter() - t) / 10
    print(f"  level {l}: {dptr[3]}x{dptr[2]} x {it} sweeps: {e*1e3:.3f} ms  ({dptr[2]*dptr[3]*it/e/1e9:.1f} Gpx-it/s, {e/it*1e6:.2f} us/sweep)")
