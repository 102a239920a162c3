"""Soak of the persistent hand-off (sc1 halo loads without an agent acquire, monotonic flags): N back-to-back persistent
1080p solves -- 3 of 5 Chebyshev-Jacobi x 1000 (125 exchanges of 252 workgroups each), 1 red-black SOR x 200, 1 V-cycle x 3 -- while a
second stream streams 1 GiB through the memory system every third solve (uneven load, warm caches), then M cold 1080p estimates.
EVERY Jacobi result must equal the CPU oracle's bits (computed once here), every other result its first run's, and no launch may
time out: since round 4 a time-out is healed silently, so RTDD_OPT_TIMEOUT_HEALS must still read 0 at the end of each phase.  Third
phase (round 4): K pairs of pipelined live frames (rtdd_live_submit, two in flight, uploads / downloads on their own streams) from a
cold start, each map against the oracle cascade's first / second estimate.  Fourth phase: H fresh contexts, each made to time out (a
hand-off flag withheld) in front of three queued solves and an effect, each healed to the oracle's bits.
Fifth phase (round 5): batches of six cold estimates in the same launches (rtdd_estimate_depth_batch), each against the oracle cascade.
usage: soak.py [N solves] [M estimates] [K live pairs] [H healed contexts] [batches]"""
import sys, os, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle                                             # the checker (this is a test script, not the product)
import realtimedepthdiffusion_amd as rt
from realtimedepthdiffusion_amd.synth import make_problem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
m_est = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
k_live = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
rows, cols = 1080, 1920
p = make_problem(rows, cols, seed=1234)
lut = oracle.load_weights(0.4)
want = oracle.solve(p["depth"].copy(), p["mask"], p["gray"], 1000, 0, 0, lut, 1, threads=oracle.max_threads())
ref = {"jacobi": hashlib.sha1(want.tobytes()).hexdigest()}
main, side = torch.cuda.Stream(), torch.cuda.Stream()
c = rt.Context(0); c.set_stream(main.cuda_stream); c.GPULoadWeights(0.4); c.GPUAllocateDeviceMemory(rows, cols, 1)
m = rt.device_image(p["mask"]); g = rt.device_image(p["gray"]); src = rt.device_image(p["depth"])
noise = torch.empty(256 << 20, dtype=torch.float32, device="cuda:0")
torch.cuda.synchronize()
t = time.time()
for i in range(n):
    kind = ("jacobi", "jacobi", "jacobi", "sor", "mg")[i % 5]
    with torch.cuda.stream(main):
        d = src.clone()
    main.synchronize()
    if i % 3 == 0:
        with torch.cuda.stream(side):
            noise.mul_(1.0001)                            # a bandwidth hog overlapping this solve
    if kind == "jacobi": c.GPUMatrixFreeSolver(d, m, g, rows, cols, 0.4, 1000, 1e-5, 0)
    elif kind == "sor": c.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_RED_BLACK_GS, maxIterations=200, relaxation=1.9)
    else: c.solve_ex(d, m, g, rows, cols, 0, method=rt.METHOD_MULTIGRID, maxIterations=3)
    c.synchronize()                                       # raises on RTDD_ERR_TIMEOUT
    h = hashlib.sha1(d.cpu().numpy().tobytes()).hexdigest()
    assert ref.setdefault(kind, h) == h, (i, kind, "differs from " + ("the ORACLE" if kind == "jacobi" else "its first run"))
    if i % 1000 == 999: print(i + 1, "solves ok, %.1f s" % (time.time() - t), flush=True)
assert c.get_option(rt.OPT_TIMEOUT_HEALS) == 0 and c.get_option(rt.OPT_PERSISTENT) == 1, "a persistent launch timed out (and was healed)"
print("soak ok:", n, "solves (every Jacobi result == the oracle's bits, no time-out healed),", {k: v[:12] for k, v in ref.items()}, flush=True)
c.close(); torch.cuda.synchronize()

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from cascade_ref import Cascade                           # tests/cascade_ref.py
bgr = np.repeat(p["gray"][..., None], 3, 2)
ann = np.where(p["mask"] == 255, p["edited"][..., 0], 32).astype(np.uint8)
cas = Cascade(oracle, bgr, ann, lut, 1, threads=oracle.max_threads()); cas.estimate(1000)
ref_e = hashlib.sha1(cas.depth[0].tobytes()).hexdigest()
c = rt.Context(0); c.set_stream(main.cuda_stream); c.GPULoadWeights(0.4); c.pyramid_create(rows, cols)
img, an = rt.device_image(bgr), rt.device_image(ann)
torch.cuda.synchronize()
t = time.time()
for i in range(m_est):
    c.pyramid_set_image(img); c.pyramid_set_annotation(an)          # resets the depth pyramid: a cold estimate
    if i % 3 == 0:
        with torch.cuda.stream(side):
            noise.mul_(1.0001)
    c.estimate_depth(1000)
    c.synchronize()
    h = hashlib.sha1(c.pyramid_download(rt.IMG_DEPTH, 0).tobytes()).hexdigest()
    assert h == ref_e, (i, "estimate differs from the oracle cascade")
    if i % 500 == 499: print(i + 1, "estimates ok, %.1f s" % (time.time() - t), flush=True)
assert c.get_option(rt.OPT_TIMEOUT_HEALS) == 0, "a persistent launch timed out (and was healed)"
print("soak ok:", m_est, "estimates, every one == the oracle cascade's bits", ref_e[:12], flush=True)

# ---- live frames, two in flight: cold start, frame 1 == the oracle's first estimate, frame 2 == its second (warm-started) one
u8_1 = hashlib.sha1(cas.depth_u8.tobytes()).hexdigest()
cas.estimate(1000)
u8_2 = hashlib.sha1(cas.depth_u8.tobytes()).hexdigest()
scr = rt.host_image((rows, cols)); ed = rt.host_image((rows, cols, 3)); outs = [rt.host_image((rows, cols)) for _ in range(2)]
c.pyramid_set_image(img); c.pyramid_set_annotation(an); c.synchronize()
scr.a[...] = c.pyramid_download(rt.IMG_SCRIBBLE, 0); ed.a[...] = c.pyramid_download(rt.IMG_EDITED, 0)
t = time.time()
for i in range(k_live):
    c.pyramid_set_image(img); c.pyramid_set_annotation(an)
    if i % 3 == 0:
        with torch.cuda.stream(side):
            noise.mul_(1.0001)
    c.live_submit(scr.a, ed.a, outs[0].a, 1000); c.live_submit(scr.a, ed.a, outs[1].a, 1000)
    c.live_wait(); h1 = hashlib.sha1(outs[0].a.tobytes()).hexdigest()
    c.live_wait(); h2 = hashlib.sha1(outs[1].a.tobytes()).hexdigest()
    assert (h1, h2) == (u8_1, u8_2), (i, "a live frame differs from the oracle cascade")
    if i % 500 == 499: print(i + 1, "live pairs ok, %.1f s" % (time.time() - t), flush=True)
assert c.get_option(rt.OPT_TIMEOUT_HEALS) == 0, "a persistent launch timed out (and was healed)"
print("soak ok:", k_live, "pairs of pipelined live frames, every map == the oracle cascade's", u8_1[:12], u8_2[:12])
# ---- (round 6) the same pairs with a sticky effect (rtdd_live_submit_ex): the artistic image of either frame == the oracle's effect on the
# oracle's first / second estimate
cas2 = Cascade(oracle, bgr, ann, lut, 1, threads=oracle.max_threads()); cas2.estimate(1000)
a1 = hashlib.sha1(oracle.defocus(bgr, cas2.depth[0], threads=oracle.max_threads()).tobytes()).hexdigest()
cas2.estimate(1000)
a2 = hashlib.sha1(oracle.defocus(bgr, cas2.depth[0], threads=oracle.max_threads()).tobytes()).hexdigest()
arts = [rt.host_image((rows, cols, 3)) for _ in range(2)]
k_fx = max(k_live // 4, 1)
t = time.time()
for i in range(k_fx):
    c.pyramid_set_image(img); c.pyramid_set_annotation(an)
    if i % 3 == 0:
        with torch.cuda.stream(side):
            noise.mul_(1.0001)
    c.live_submit_ex(scr.a, ed.a, outs[0].a, rt.EFFECT_DEFOCUS, arts[0].a, 1000); c.live_submit_ex(scr.a, ed.a, outs[1].a, rt.EFFECT_DEFOCUS, arts[1].a, 1000)
    c.live_wait(); h1 = (hashlib.sha1(outs[0].a.tobytes()).hexdigest(), hashlib.sha1(arts[0].a.tobytes()).hexdigest())
    c.live_wait(); h2 = (hashlib.sha1(outs[1].a.tobytes()).hexdigest(), hashlib.sha1(arts[1].a.tobytes()).hexdigest())
    assert (h1, h2) == ((u8_1, a1), (u8_2, a2)), (i, "a live frame with an effect differs from the oracle")
    if i % 200 == 199: print(i + 1, "live pairs with defocus ok, %.1f s" % (time.time() - t), flush=True)
assert c.get_option(rt.OPT_TIMEOUT_HEALS) == 0, "a persistent launch timed out (and was healed)"
print("soak ok:", k_fx, "pairs of pipelined live frames with a sticky defocus, every map and every artistic image == the oracle's")
c.close(); torch.cuda.synchronize()

# ---- the healing path itself, many times over (round 4): a fresh context per round, a hand-off flag withheld at a random tile, three
# queued 1080p solves (each on the one before's result) + an effect behind them; the synchronising call must return OK with the oracle's bits
k_heal = int(sys.argv[4]) if len(sys.argv) > 4 else 200
chain = p["depth"].copy()
for _ in range(3):
    chain = oracle.solve(chain, p["mask"], p["gray"], 120, 0, 0, lut, 1, threads=oracle.max_threads())
ref_chain = hashlib.sha1(chain.tobytes()).hexdigest()
rgb = np.random.default_rng(1).integers(0, 256, (rows, cols, 3), dtype=np.uint8)
ref_haze = hashlib.sha1(oracle.haze(rgb, chain, 1).tobytes()).hexdigest()
o = rt.device_image(rgb); art = rt.device_image(np.zeros_like(rgb))
rng = np.random.default_rng(7)
t = time.time()
for i in range(k_heal):
    c = rt.Context(0); c.set_stream(main.cuda_stream); c.GPULoadWeights(0.4); c.GPUAllocateDeviceMemory(rows, cols, 1)
    c.set_option(rt.OPT_DEBUG_POLL_LIMIT_US, 1500); c.set_option(rt.OPT_DEBUG_WITHHOLD_TILE, int(rng.integers(1, 253)))
    with torch.cuda.stream(main):
        d = src.clone()
    for _ in range(3):
        c.GPUMatrixFreeSolver(d, m, g, rows, cols, 0.4, 120, 1e-5, 0)
    c.GPUSimulateHaze(o, d, art, rows, cols)
    c.synchronize()
    assert c.get_option(rt.OPT_TIMEOUT_HEALS) == 1, (i, "no time-out?")
    assert hashlib.sha1(d.cpu().numpy().tobytes()).hexdigest() == ref_chain, (i, "healed solves differ from the oracle")
    assert hashlib.sha1(art.cpu().numpy().tobytes()).hexdigest() == ref_haze, (i, "the effect behind the healed solves differs")
    c.close()
    if i % 50 == 49: print(i + 1, "healed contexts ok, %.1f s" % (time.time() - t), flush=True)
print("soak ok:", k_heal, "contexts, each healed one timed-out persistent launch: three queued solves + haze == the oracle's bits")

# ---- fifth phase (round 5): batched estimates under the same load -- B cold 1080p estimates in the same launches (blockIdx.z = image,
# one flag set per image and tile in the persistent levels), every image's finest level == the oracle cascade's first estimate
k_batch = int(sys.argv[5]) if len(sys.argv) > 5 else 200
B = 6
c = rt.Context(0); c.set_stream(main.cuda_stream); c.GPULoadWeights(0.4); c.pyramid_create_batch(rows, cols, B)
t = time.time()
for i in range(k_batch):
    for b in range(B):
        c.pyramid_select(b); c.pyramid_set_image(img); c.pyramid_set_annotation(an)      # cold
    if i % 3 == 0:
        with torch.cuda.stream(side):
            noise.mul_(1.0001)
    c.estimate_depth_batch(1000)
    c.synchronize()
    for b in (i % B, (i + 3) % B):                        # two of the six per round (hashing 8 MB on the host is what this loop costs)
        c.pyramid_select(b)
        assert hashlib.sha1(c.pyramid_download(rt.IMG_DEPTH, 0).tobytes()).hexdigest() == ref_e, (i, b, "a batched estimate differs from the oracle cascade")
    if i % 100 == 99: print(i + 1, "batches ok, %.1f s" % (time.time() - t), flush=True)
assert c.get_option(rt.OPT_TIMEOUT_HEALS) == 0, "a persistent launch timed out (and was healed)"
print("soak ok:", k_batch, "batches of", B, "cold estimates, every checked image == the oracle cascade's bits")
c.close()
