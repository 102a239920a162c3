"""Twelve pipelined live frames at 1080p (rtdd_live_submit, two in flight) for a kernel + memory-copy trace: what a frame costs beyond the estimate.
usage: prof_live.py [defocus|desaturation|haze]   (round 6: the frame with a sticky effect, rtdd_live_submit_ex)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import realtimedepthdiffusion_amd as rt
from realtimedepthdiffusion_amd.synth import make_problem
rows, cols = 1080, 1920
p = make_problem(rows, cols, seed=1234)
bgr = np.repeat(p["gray"][..., None], 3, 2); ann = np.where(p["mask"] == 255, p["edited"][..., 0], 32).astype(np.uint8)
c = rt.Context(0); c.set_stream(torch.cuda.Stream().cuda_stream); c.GPULoadWeights(0.4)
c.pyramid_create(rows, cols); c.pyramid_set_image(rt.device_image(bgr)); c.pyramid_set_annotation(rt.device_image(ann)); c.synchronize()
scr = rt.host_image((rows, cols)); ed = rt.host_image((rows, cols, 3)); outs = [rt.host_image((rows, cols)) for _ in range(2)]
scr.a[...] = c.pyramid_download(rt.IMG_SCRIBBLE, 0); ed.a[...] = c.pyramid_download(rt.IMG_EDITED, 0)
fx = {"defocus": rt.EFFECT_DEFOCUS, "desaturation": rt.EFFECT_DESATURATION, "haze": rt.EFFECT_HAZE}.get(sys.argv[1] if len(sys.argv) > 1 else "", rt.EFFECT_NONE)
arts = [rt.host_image((rows, cols, 3)) for _ in range(2)]
for f in range(12):
    if f >= 2: c.live_wait()
    c.live_submit_ex(scr.a, ed.a, outs[f % 2].a, fx, arts[f % 2].a if fx else None, 1000)
while c.live_pending(): c.live_wait()
c.synchronize(); c.close()
