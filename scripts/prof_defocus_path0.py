"""20 defocus calls at 1080p and 4K (smooth depth) on path 0, for a kernel trace."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import realtimedepthdiffusion_amd as rt
from realtimedepthdiffusion_amd.synth import make_problem
path = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for rows, cols in ((1080, 1920), (2160, 3840)):
    p = make_problem(rows, cols, seed=1)
    orig = np.random.default_rng(0).integers(0, 256, (rows, cols, 3), dtype=np.uint8)
    c = rt.Context(0); c.set_option(rt.OPT_DEFOCUS_PATH, path)
    o, d = rt.device_image(orig), rt.device_image(p["gray"].astype(np.float32))
    art = rt.device_image(np.zeros_like(orig))
    for _ in range(20): c.GPUSimulateDefocus(o, d, art, rows, cols)
    c.synchronize(); c.close()
