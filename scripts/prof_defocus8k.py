"""The defocus effect at 8K, 10 calls each on a piecewise-smooth depth map and on a real one (the Dog estimate tiled), for a kernel trace:
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d DIR -o df -- python3 scripts/prof_defocus8k.py [strips 0|1|2] [slice MB]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import realtimedepthdiffusion_amd as rt
from realtimedepthdiffusion_amd.synth import make_problem
rows, cols = 4320, 7680
p = make_problem(rows, cols, seed=1)
orig = np.random.default_rng(0).integers(0, 256, (rows, cols, 3), dtype=np.uint8)
c = rt.Context(0); c.set_stream(torch.cuda.current_stream().cuda_stream)
if len(sys.argv) > 1:
    c.set_option(rt.OPT_DEFOCUS_STRIPS, int(sys.argv[1]))
if len(sys.argv) > 2:
    c.set_option(rt.OPT_DEFOCUS_SLICE_MB, int(sys.argv[2]))
o = rt.device_image(orig); ds = rt.device_image(p["gray"].astype(np.float32)); art = rt.device_image(np.zeros_like(orig))
for _ in range(10):
    c.GPUSimulateDefocus(o, ds, art, rows, cols)
torch.cuda.synchronize(); c.close()
