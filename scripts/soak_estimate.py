"""Soak of the cascade: N back-to-back 1080p depth estimates (cold pyramid each time) must give the same bits every time.
usage: soak_estimate.py [N]"""
import sys, os, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import realtimedepthdiffusion_amd as rt
from realtimedepthdiffusion_amd.synth import make_problem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rows, cols = 1080, 1920
p = make_problem(rows, cols, seed=7)
c = rt.Context(0); c.GPULoadWeights(0.4)
bgr = np.repeat(p["gray"][..., None], 3, 2)
ann = np.where(p["mask"] == 255, p["edited"][..., 0], 32).astype(np.uint8)
c.pyramid_create(rows, cols)
img, an = rt.device_image(bgr), rt.device_image(ann)
ref = None
t = time.time()
for i in range(n):
    c.pyramid_set_image(img); c.pyramid_set_annotation(an)          # resets the depth pyramid: a cold estimate
    c.estimate_depth(1000)
    c.synchronize()
    d = c.pyramid_download(rt.IMG_DEPTH, 0)                          # the f32 depth map of the finest level
    h = hashlib.sha1(d.tobytes()).hexdigest()
    if ref is None: ref = h
    assert h == ref, (i, h, ref)
    if i % 500 == 499: print(i + 1, "estimates ok, %.1f s" % (time.time() - t), flush=True)
print("soak ok:", n, "estimates,", ref)
