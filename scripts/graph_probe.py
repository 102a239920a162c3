"""Would a hipGraph of the whole estimate (about 100 launches at 1080p) be faster than the stream of launches?
Captured here with torch.cuda.graph around rtdd_estimate_depth (the library launches on the stream it is given)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import realtimedepthdiffusion_amd as rt
from realtimedepthdiffusion_amd.synth import make_problem
rows, cols = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
p = make_problem(rows, cols, seed=1234)
bgr = np.repeat(p["gray"][..., None], 3, 2); ann = np.where(p["mask"] == 255, p["edited"][..., 0], 32).astype(np.uint8)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    c = rt.Context(0); c.set_stream(s.cuda_stream); c.GPULoadWeights(0.4)
    c.pyramid_create(rows, cols); c.pyramid_set_image(rt.device_image(bgr)); c.pyramid_set_annotation(rt.device_image(ann))
    for _ in range(3): c.estimate_depth(1000)
    s.synchronize()
    t = time.perf_counter()
    for _ in range(50): c.estimate_depth(1000)
    s.synchronize(); eager = (time.perf_counter() - t) / 50
    ref = c.pyramid_download(rt.IMG_DEPTH, 0)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        c.estimate_depth(1000)
    for _ in range(3): g.replay()
    s.synchronize()
    t = time.perf_counter()
    for _ in range(50): g.replay()
    s.synchronize(); graph = (time.perf_counter() - t) / 50
    print(f"{cols}x{rows}: eager {eager*1e3:.3f} ms, graph replay {graph*1e3:.3f} ms")
