"""Prototype (float64, numpy): does alternating zebra LINE relaxation repair the V-cycle where point smoothing stalls?
The stall (boxmg_proto.py, DIAG=1): residual concentrated on one-pixel-wide rows whose horizontal coupling is ~0.5 and whose vertical
coupling is ~1e-7 -- a 1-D chain that standard 2x coarsening cannot represent when it lies on an odd row.
usage: zebra_probe.py ROWS COLS NU CYCLES [seed]   env: LINES=0|1 (level-0 smoother), COARSE_LINES=0|1, GOLDEN=name"""
import sys, os, time
import numpy as np
import boxmg_proto as bp
from boxmg_proto import shift, stencil8, fine_level, build_P, galerkin, gs4, apply_A
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from realtimedepthdiffusion_amd.synth import make_problem


def thomas_rows(lo, di, up, rhs):
    """Solve, for every row r, the tridiagonal system lo[r,i] x[i-1] + di[r,i] x[i] + up[r,i] x[i+1] = rhs[r,i] (vectorised over rows)."""
    n = di.shape[1]
    c = np.zeros_like(di); d = np.zeros_like(di)
    c[:, 0] = up[:, 0] / di[:, 0]; d[:, 0] = rhs[:, 0] / di[:, 0]
    for i in range(1, n):
        den = di[:, i] - lo[:, i] * c[:, i - 1]
        c[:, i] = up[:, i] / den
        d[:, i] = (rhs[:, i] - lo[:, i] * d[:, i - 1]) / den
    x = np.zeros_like(di)
    x[:, -1] = d[:, -1]
    for i in range(n - 2, -1, -1):
        x[:, i] = d[:, i] - c[:, i] * x[:, i + 1]
    return x


def line_sweep(W, D, x, b, axis, parity):
    """Solve all lines (rows if axis == 1, columns if axis == 0) of the given parity exactly, other lines fixed.  W: dict of the 8 couplings."""
    act = D > 0
    if axis == 0:       # columns: transpose everything
        Wt = {(dx, dy): w.T for (dy, dx), w in W.items()}
        return line_sweep(Wt, D.T, x.T, b.T, 1, parity).T
    rhs = b.copy()
    for (dy, dx), w in W.items():
        if dy != 0:
            rhs += w * shift(x, dy, dx)
    di = np.where(act, D, 1.0); lo = np.where(act, -W[(0, -1)], 0.0); up = np.where(act, -W[(0, 1)], 0.0); rhs = np.where(act, rhs, 0.0)
    sel = np.arange(parity, D.shape[0], 2)
    xn = x.copy()
    xn[sel] = np.where(act[sel], thomas_rows(lo[sel], di[sel], up[sel], rhs[sel]), x[sel])
    return xn


def zebra(L, x, b, n, reverse=False):
    W = stencil8(L)
    steps = [(1, 0), (1, 1), (0, 0), (0, 1)]
    if reverse: steps = steps[::-1]
    for _ in range(n):
        for axis, parity in steps:
            x = line_sweep(W, L[4], x, b, axis, parity)
    return x


def vcycle(levels, Ps, l, x, b, nu, lines0, linesc):
    L = levels[l]
    if l == len(levels) - 1:
        return gs4(L, x, b, 30)
    use = lines0 if l == 0 else linesc
    x = zebra(L, x, b, nu) if use else gs4(L, x, b, nu)
    r = np.where(L[4] > 0, b - apply_A(L, x), 0.0)
    shape_c = levels[l + 1][4].shape
    rc = (Ps[l].T @ r.ravel()).reshape(shape_c)
    ec = vcycle(levels, Ps, l + 1, np.zeros(shape_c), rc, nu, lines0, linesc)
    x = x + (Ps[l] @ ec.ravel()).reshape(x.shape)
    return zebra(L, x, b, nu, True) if use else gs4(L, x, b, nu, order=(3, 2, 1, 0))


if __name__ == "__main__":
    rows, cols, nu, cycles = (int(v) for v in sys.argv[1:5])
    seed = int(sys.argv[5]) if len(sys.argv) > 5 else 1234
    p = make_problem(rows, cols, seed=seed)
    if os.environ.get("GOLDEN"):
        g = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", os.environ["GOLDEN"] + "_256.npz"))
        p = {"gray": g["gray0"], "mask": g["mask0"], "depth": g["depth_before_c1_L0"]}
    os.environ.setdefault("THETA", "0")
    # TRUE operator at level 0 as well (no theta) -- the hierarchy uses the thresholded one
    L0t, (wr, wd, free) = fine_level(p["gray"], p["mask"])
    levels = [L0t]; Ps = []
    while min(levels[-1][4].shape) > 8 and len(levels) < 12:
        P = build_P(levels[-1]); Lc, Psp = galerkin(levels[-1], P); levels.append(Lc); Ps.append(Psp)
    x0 = p["depth"].astype(np.float64)
    wl, wu = shift(wr, 0, -1), shift(wd, -1, 0); dsum = wl + wr + wu + wd
    def res(x):
        j = (wl * shift(x, 0, -1) + wr * shift(x, 0, 1) + wu * shift(x, -1, 0) + wd * shift(x, 1, 0)) / np.where(dsum > 0, dsum, 1)
        return np.abs(np.where(free, j - x, 0)).max()
    xd = np.where(free, 0.0, x0)
    b = np.where(free, wl * shift(xd, 0, -1) + wr * shift(xd, 0, 1) + wu * shift(xd, -1, 0) + wd * shift(xd, 1, 0), 0.0)
    # level-0 operator for the smoother: the true one (all links between free points)
    E = np.where(free & shift(free, 0, 1), wr, 0.0); S = np.where(free & shift(free, 1, 0), wd, 0.0)
    levels[0] = (E, S, np.zeros_like(E), np.zeros_like(E), np.where(free, dsum, 0.0))
    x = np.where(free, x0, 0.0)
    lines0 = os.environ.get("LINES", "1") == "1"; linesc = os.environ.get("COARSE_LINES", "0") == "1"
    print("levels", [l[4].shape for l in levels], "lines0", lines0, "coarse lines", linesc, "initial", res(np.where(free, x, x0)))
    prev = None; t = time.time()
    for c in range(cycles):
        x = vcycle(levels, Ps, 0, x, b, nu, lines0, linesc)
        r = res(np.where(free, x, x0))
        print(f"cycle {c + 1}: residual {r:.3e}" + (f"  factor {r / prev:.3f}" if prev else "") + f"  ({time.time() - t:.1f}s)", flush=True)
        prev = r
        if r < 1e-6: break
