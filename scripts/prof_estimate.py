"""Ten 1080p depth estimates (the 5-level cascade), for a kernel trace:
rocprofv3 --kernel-trace --output-format csv -d DIR -o est -- python3 scripts/prof_estimate.py ; then scripts/prof_estimate_summary.py DIR"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import realtimedepthdiffusion_amd as rt
from realtimedepthdiffusion_amd.synth import make_problem
dev = "cuda:0"
rows, cols = 1080, 1920
p = make_problem(rows, cols, seed=1)
c = rt.Context(0)
c.set_stream(torch.cuda.current_stream().cuda_stream)
c.GPULoadWeights(0.4)
bgr = np.repeat(p["gray"][..., None], 3, 2)
ann = np.where(p["mask"] == 255, p["edited"][..., 0], 32).astype(np.uint8)
c.pyramid_create(rows, cols)
c.pyramid_set_image(rt.device_image(bgr, dev)); c.pyramid_set_annotation(rt.device_image(ann, dev))
for _ in range(10):
    c.estimate_depth(1000)
    c.synchronize()
c.pyramid_destroy(); c.close()
