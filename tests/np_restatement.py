"""Independent numpy restatement of the hot-path kernels (second witness for the C oracle).

Written from SURVEY.md Appendix A / the reference sources, in a different language and a
different (whole-array, shifted-neighbour) structure than oracle/rtdd_oracle.c, so that a
transcription slip in one is unlikely to be repeated in the other.  All arithmetic is
numpy float32 (one IEEE rounding per op); the contracted variant uses an EXACT float32 fma
built from float64 products + a round-to-odd sum, so both FP-contraction variants can be
compared bit-for-bit.

Cites: /root/reference/src/GPUSolver.cu:73-106 (mean), :136-224 (indices), :226-262 (sweep),
:282-312 (driver).
"""
import numpy as np

f32 = np.float32


def fma32(a, b, c):
    """Exact fmaf for float32 arrays: round_to_f32(a*b + c) with a single rounding."""
    a = np.asarray(a, np.float32).astype(np.float64)
    b = np.asarray(b, np.float32).astype(np.float64)
    c = np.asarray(c, np.float32).astype(np.float64)
    p = a * b                       # exact: 24+24 significand bits fit in 53
    s = p + c                       # rounded to nearest f64
    bb = s - p                      # TwoSum error term (exact)
    e = (p - (s - bb)) + (c - bb)
    # round-to-odd: if inexact and s is even, step towards the true value
    bits = s.view(np.int64) if isinstance(s, np.ndarray) else np.float64(s).view(np.int64)
    even = (bits & 1) == 0
    step_up = (e > 0) & even
    step_dn = (e < 0) & even
    s = np.where(step_up, np.nextafter(s, np.inf), np.where(step_dn, np.nextafter(s, -np.inf), s))
    return s.astype(np.float32)


def index_maps(gray, depth, level, max_level):
    """left/right/up/down index maps (int32, 256 = no neighbour)."""
    g = gray.astype(np.int32)
    rows, cols = g.shape
    out = {k: np.full((rows, cols), 256, np.int32) for k in ("left", "right", "up", "down")}

    def sad(a, b):
        return np.abs(a - b)

    gh = sad(g[:, 1:], g[:, :-1])    # between x-1 and x
    gv = sad(g[1:, :], g[:-1, :])    # between y-1 and y
    if level != max_level:
        d = np.where(depth >= 0, np.minimum(depth, 255), 0)      # saturating u8 cast (defined behaviour)
        d = np.where(np.isnan(depth), 0, d).astype(np.int32)     # trunc toward zero for d >= 0
        thr = 0 if level == 0 else 4
        gh = np.where(sad(d[:, 1:], d[:, :-1]) > thr, gh, 0)
        gv = np.where(sad(d[1:, :], d[:-1, :]) > thr, gv, 0)
    out["left"][:, 1:] = gh
    out["right"][:, :-1] = gh
    out["up"][1:, :] = gv
    out["down"][:-1, :] = gv
    return out


def pack_index(maps):
    return np.stack([maps["left"] * 1000 + maps["right"], maps["up"] * 1000 + maps["down"]], axis=-1).astype(np.int32)


def sweep(x, maps, mask, prev, omega, lut, contract, gamma=f32(0.99)):
    """Returns (out, new_prev).  Dirichlet pixels keep x / prev."""
    rows, cols = x.shape
    x = x.astype(np.float32)
    pad = np.zeros((rows + 2, cols + 2), np.float32)
    pad[1:-1, 1:-1] = x
    nb = {"left": pad[1:-1, :-2], "right": pad[1:-1, 2:], "up": pad[:-2, 1:-1], "down": pad[2:, 1:-1]}
    s = np.zeros((rows, cols), np.float32)
    cnt = np.zeros((rows, cols), np.float32)
    for k in ("left", "right", "up", "down"):
        idx = maps[k]
        valid = idx != 256
        w = lut[np.minimum(idx, 256)]
        if contract:
            s2 = fma32(w, nb[k], s)
        else:
            s2 = (s + (w * nb[k]).astype(np.float32)).astype(np.float32)
        s = np.where(valid, s2, s)
        cnt = np.where(valid, (cnt + w).astype(np.float32), cnt)
    with np.errstate(divide="ignore", invalid="ignore"):
        q = (s / cnt).astype(np.float32)
    r = np.where(q >= 0, q, f32(0))
    r = np.where(r > 255, f32(255), r).astype(np.float32)
    r = np.where(cnt == 0, f32(0), r).astype(np.float32)
    omega = f32(omega); gamma = f32(gamma)
    if contract:
        t = fma32(gamma, (r - x).astype(np.float32), x)
        o = fma32(omega, (t - prev).astype(np.float32), prev)
    else:
        t = ((gamma * (r - x)).astype(np.float32) + x).astype(np.float32)
        o = ((omega * (t - prev).astype(np.float32)).astype(np.float32) + prev).astype(np.float32)
    free = mask != 255
    return np.where(free, o, x).astype(np.float32), np.where(free, x, prev).astype(np.float32)


def omega_schedule(n):
    S = 10
    rho = f32(0.99)
    om = f32(0)
    out = []
    for it in range(n):
        if it < S:
            om = f32(1)
        elif it == S:
            om = f32(2.0 / (2.0 - float(f32(rho * rho))))
        else:
            om = f32(4.0 / (4.0 - float(f32(f32(rho * rho) * om))))
        out.append(om)
    return np.array(out, np.float32)


def solve(depth, mask, gray, iters, level, max_level, lut, contract):
    maps = index_maps(gray, depth, level, max_level)
    x = depth.astype(np.float32).copy()
    prev = np.zeros_like(x)
    for om in omega_schedule(iters):
        x, prev = sweep(x, maps, mask, prev, om, lut, contract)
    return x
