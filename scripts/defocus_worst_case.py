import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import realtimedepthdiffusion_amd as rt
for rows, cols in ((270, 480), (540, 960), (1080, 1920)):
    orig = np.random.default_rng(0).integers(0, 256, (rows, cols, 3), dtype=np.uint8)
    depth = np.full((rows, cols), 1e6, np.float32)
    c = rt.Context(0); o = rt.device_image(orig); d = rt.device_image(depth); art = rt.device_image(np.zeros_like(orig))
    for call in range(3):                  # automatic path: the first call meets the windows in the tile kernel, the later ones take the table
        t = time.perf_counter(); c.GPUSimulateDefocus(o, d, art, rows, cols); c.synchronize()
        print(f"{cols}x{rows} every window the whole image, automatic path, call {call}: {(time.perf_counter()-t)*1e3:.2f} ms", flush=True)
    d2 = rt.device_image(np.full((rows, cols), 5000.0, np.float32))
    c2 = rt.Context(0)
    for call in range(2):
        t = time.perf_counter(); c2.GPUSimulateDefocus(o, d2, art, rows, cols); c2.synchronize()
        print(f"{cols}x{rows} depth 5000 everywhere, automatic path, call {call}: {(time.perf_counter()-t)*1e3:.2f} ms", flush=True)
    c2.close()
    c.close()
