"""ms per estimate on the twelve bundled photographs (bench.py estimate_dataset), alone: for A/B builds (RTDD_LIBRARY)."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import realtimedepthdiffusion_amd as rt
import bench
d = bench.estimate_dataset(rt, "cuda:0", n=20)
print(os.environ.get("RTDD_LIBRARY", "default"), {k: v["ms"] for k, v in d["pairs"].items()}, "mean", round(d["mean_ms"], 4))
