"""Calls the defocus effect N times at 1080p and 4K (profile with: rocprofv3 --kernel-trace --stats -d DIR -- python3 scripts/prof_defocus.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import realtimedepthdiffusion_amd as rt
from realtimedepthdiffusion_amd.synth import make_problem
dev = "cuda:0"
for rows, cols in ((1080, 1920), (2160, 3840)):
    p = make_problem(rows, cols, seed=1)
    rng = np.random.default_rng(0)
    orig = rng.integers(0, 256, (rows, cols, 3), dtype=np.uint8)
    depth = (p["depth"] * rng.uniform(0, 1, (rows, cols))).astype(np.float32)
    c = rt.Context(0)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    o, d = rt.device_image(orig, dev), rt.device_image(depth, dev)
    ds = rt.device_image(p["gray"].astype(np.float32), dev)       # a piecewise-smooth depth map (value noise + rectangles)
    art = rt.device_image(np.zeros_like(orig), dev)
    for _ in range(20):
        c.GPUSimulateDefocus(o, d, art, rows, cols)               # per-pixel random depth: every window size, no coherence
    for _ in range(20):
        c.GPUSimulateDefocus(o, ds, art, rows, cols)
    torch.cuda.synchronize()
    c.close()
