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
    # the two coarsest levels on their own (the launch loops of k_sweep_col: 36 and 21 launches): eager against a captured graph
    for lr, lc, it in ((67, 120, 1000), (135, 240, 500)):
        q = make_problem(lr, lc, seed=7)
        c2 = rt.Context(0); c2.set_stream(s.cuda_stream); c2.GPULoadWeights(0.4); c2.GPUAllocateDeviceMemory(lr, lc, 1)
        d, m, gr = rt.device_image(q["depth"]), rt.device_image(q["mask"]), rt.device_image(q["gray"])
        f = lambda: c2.GPUMatrixFreeSolver(d, m, gr, lr, lc, 0.4, it, 0.0, 0)
        for _ in range(3): f()
        s.synchronize(); t = time.perf_counter()
        for _ in range(100): f()
        s.synchronize(); eager = (time.perf_counter() - t) / 100
        info = c2.last_solve_info()
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, stream=s):
            f()
        for _ in range(3): g2.replay()
        s.synchronize(); t = time.perf_counter()
        for _ in range(100): g2.replay()
        s.synchronize(); graph = (time.perf_counter() - t) / 100
        print(f"{lc}x{lr} x {it} sweeps ({info.launches} launches of tile {info.tile}, {info.temporal_depth} sweeps each): eager {eager*1e6:.1f} us, graph replay {graph*1e6:.1f} us")
        c2.synchronize(); c2.close()
