"""The estimate legs of the bench line alone (bench.estimate_ms): device-resident ms, the reference's own timing region
(src/main.cpp:234-293: upload + cascade + download) one frame at a time, and the same frames two in flight.  usage: estimate_e2e.py [n]"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
import realtimedepthdiffusion_amd as rt
from realtimedepthdiffusion_amd.synth import make_problem

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for rows, cols in ((1080, 1920), (2160, 3840)):
    c = rt.Context(0); c.set_stream(torch.cuda.current_stream().cuda_stream); c.GPULoadWeights(0.4)
    r = bench.estimate_ms(rt, c, make_problem(rows, cols, seed=1234), rows, cols, "cuda:0", n=n)
    c.close()
    print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if not k.endswith("_is")}))
