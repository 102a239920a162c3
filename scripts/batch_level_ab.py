"""A/B of the sweep configuration of ONE pyramid level inside the 64-image 1080p batch (needs the librtdd_force.so variant:
scripts/build_variant.sh force "-DRTDD_FORCE_CFG_HOOK=1" sweep_blocked.hip).  Prints the whole batched estimate's ms per forced choice."""
import sys, os, time, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import numpy as np, torch
    import realtimedepthdiffusion_amd as rt
    from realtimedepthdiffusion_amd.synth import make_problem
    rows, cols, images = 1080, 1920, 64
    ctx = rt.Context(0); ctx.GPULoadWeights(0.4)
    ctx.pyramid_create_batch(rows, cols, images)
    p = make_problem(rows, cols, seed=1)
    img = rt.device_image(np.repeat(p["gray"][..., None], 3, 2)); ann = rt.device_image(np.where(p["mask"] == 255, p["edited"][..., 0], 32).astype(np.uint8))
    for b in range(images):
        ctx.pyramid_select(b); ctx.pyramid_set_image(img); ctx.pyramid_set_annotation(ann)
    ctx.estimate_depth_batch(1000); ctx.synchronize()
    t = time.perf_counter()
    for _ in range(5): ctx.estimate_depth_batch(1000)
    ctx.synchronize()
    print(f"{(time.perf_counter() - t) / 5 * 1e3:.3f}")
    sys.exit(0)
levels = {4: (120, 67), 3: (240, 135), 2: (480, 270), 1: (960, 540), 0: (1920, 1080)}
cases = [("default", "")]
if len(sys.argv) > 1 and sys.argv[1] == "--round2":
    for t, d, pe in ((4, 16, 0), (4, 24, 0), (3, 12, 0), (3, 16, 0), (3, 24, 0), (12, 12, 0), (13, 12, 0), (9, 12, 0), (9, 16, 0), (5, 12, 0)):
        cases.append((f"L3 tile {t} T {d} p {pe}", f"240,135,{t},{d},{pe},0"))
    for t, d, pe in ((4, 16, 0), (3, 12, 0), (3, 16, 0), (5, 12, 0), (7, 12, 0), (6, 16, 0), (13, 12, 0)):
        cases.append((f"L2 tile {t} T {d} p {pe}", f"480,270,{t},{d},{pe},0"))
    best = "120,67,4,8,0,0;480,270,6,12,0,0;1920,1080,4,8,1,1"
    cases.append(("L4 + L2 + L0 best", best))
    cases.append(("L4 + L2 + L0 best + L3 tile 4 T 12", best + ";240,135,4,12,0,0"))
else:
    for t, d, pe in ((4, 8, 0), (4, 8, 1), (3, 8, 0), (12, 8, 0), (13, 8, 0), (8, 8, 1), (8, 4, 1), (8, 12, 1), (9, 8, 1), (14, 28, 0), (5, 8, 1)):
        cases.append((f"L4 tile {t} T {d} p {pe}", f"120,67,{t},{d},{pe},0"))
    for t, d, pe in ((4, 4, 1), (4, 8, 1), (4, 8, 0), (4, 12, 0), (3, 8, 1), (3, 4, 1), (3, 12, 1), (6, 8, 0), (5, 8, 0), (8, 8, 1), (8, 4, 1), (9, 8, 0), (14, 24, 0)):
        cases.append((f"L3 tile {t} T {d} p {pe}", f"240,135,{t},{d},{pe},0"))
    for t, d, pe in ((6, 8, 0), (6, 12, 0), (5, 4, 0), (5, 8, 0), (4, 8, 0), (4, 12, 0), (7, 8, 0), (9, 8, 0), (12, 8, 0)):
        cases.append((f"L2 tile {t} T {d} p {pe}", f"480,270,{t},{d},{pe},0"))
    for t, d, pe in ((6, 8, 0), (6, 12, 0), (4, 8, 0), (4, 12, 0), (5, 8, 0)):
        cases.append((f"L1 tile {t} T {d} p {pe}", f"960,540,{t},{d},{pe},0"))
    for t, d, pe, pi in ((4, 8, 1, 1), (6, 8, 0, 0), (6, 12, 0, 0), (4, 8, 0, 0), (4, 12, 0, 0)):
        cases.append((f"L0 tile {t} T {d} p {pe} per-image {pi}", f"1920,1080,{t},{d},{pe},{pi}"))
for name, cfg in cases:
    env = dict(os.environ, RTDD_LIBRARY="realtimedepthdiffusion_amd/librtdd_force.so", RTDD_FORCE_CFG=cfg, RTDD_DEBUG_CONFIG="1" if name == "default" else "")
    if not env["RTDD_DEBUG_CONFIG"]: del env["RTDD_DEBUG_CONFIG"]
    r = subprocess.run([sys.executable, __file__, "--one"], env=env, capture_output=True, text=True, timeout=300)
    out = [l for l in r.stdout.split("\n") if l.strip()]
    print(f"{name:40s} {out[-1] if out else 'FAILED ' + r.stderr[-300:]}", flush=True)
    if name == "default":
        seen = set()
        for l in r.stderr.split("\n"):
            if l.startswith("[rtdd]") and l not in seen: seen.add(l); print("   ", l)
