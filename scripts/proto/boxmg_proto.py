"""numpy/scipy prototype (float64, convergence study only): multigrid with OPERATOR-DEPENDENT interpolation
(Dendy's black-box multigrid: standard 2x coarsening, interpolation weights taken from the stencil, Galerkin 9-point
coarse operators) for the depth-diffusion system.  usage: boxmg_proto.py ROWS COLS NU [CYCLES] [seed]
Every level is a 9-point stencil on a grid: weights E,S,SE,SW >= 0 (couplings to the right/down/down-right/down-left
neighbour; the other four come from the neighbours by symmetry) and a diagonal D; D == 0 marks an inactive point (e = 0)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.sparse as sp
from realtimedepthdiffusion_amd.synth import make_problem


def shift(a, dy, dx):
    """b[y, x] = a[y + dy, x + dx], zero outside."""
    R, C = a.shape
    b = np.zeros_like(a)
    ys = slice(max(0, -dy), min(R, R - dy)); xs = slice(max(0, -dx), min(C, C - dx))
    yd = slice(max(0, -dy) + dy, min(R, R - dy) + dy); xd = slice(max(0, -dx) + dx, min(C, C - dx) + dx)
    b[ys, xs] = a[yd, xd]
    return b


def stencil8(L):
    """All eight couplings of every point: dict dir -> array.  dirs as (dy, dx)."""
    E, S, SE, SW, D = L
    return {(0, 1): E, (1, 0): S, (1, 1): SE, (1, -1): SW,
            (0, -1): shift(E, 0, -1), (-1, 0): shift(S, -1, 0), (-1, -1): shift(SE, -1, -1), (-1, 1): shift(SW, -1, 1)}


def fine_level(gray, mask, beta=0.4):
    g = gray.astype(np.int64); lut = np.exp(-beta * np.arange(256)).astype(np.float32).astype(np.float64)
    R, C = g.shape
    wr = np.zeros((R, C)); wd = np.zeros((R, C))
    wr[:, :-1] = lut[np.abs(g[:, 1:] - g[:, :-1])]
    wd[:-1, :] = lut[np.abs(g[1:, :] - g[:-1, :])]
    free = mask != 255
    D = np.where(free, wr + wd + shift(wr, 0, -1) + shift(wd, -1, 0), 0.0)          # includes links to Dirichlet pixels
    theta = float(os.environ.get("THETA", "0"))                                      # links weaker than theta stay in D only (anchors)
    rel = float(os.environ.get("REL", "0"))                                          # experiment: links weaker than rel x the strongest link of EITHER endpoint stay in D only
    mx = np.maximum(np.maximum(wr, wd), np.maximum(shift(wr, 0, -1), shift(wd, -1, 0)))
    okr = wr >= rel * np.minimum(mx, shift(mx, 0, 1)) if os.environ.get("RELMODE", "min") == "min" else wr >= rel * np.maximum(mx, shift(mx, 0, 1))
    okd = wd >= rel * np.minimum(mx, shift(mx, 1, 0)) if os.environ.get("RELMODE", "min") == "min" else wd >= rel * np.maximum(mx, shift(mx, 1, 0))
    E = np.where(free & shift(free, 0, 1) & (wr >= theta) & okr, wr, 0.0)             # links to inactive points are dropped
    S = np.where(free & shift(free, 1, 0) & (wd >= theta) & okd, wd, 0.0)
    DT = np.dtype(os.environ.get("DT", "float64"))
    return tuple(a.astype(DT) for a in (E, S, np.zeros_like(E), np.zeros_like(E), D)), (wr.astype(DT), wd.astype(DT), free)


def build_P(L):
    """Interpolation weights of every fine point towards the four corners (c00, c01, c10, c11) of its coarse cell."""
    E, S, SE, SW, D = L
    W = stencil8(L)
    R, C = D.shape
    yy, xx = np.mgrid[0:R, 0:C]
    act = D > 0
    P = np.zeros((4, R, C), D.dtype)
    ee = (yy % 2 == 0) & (xx % 2 == 0); eo = (yy % 2 == 0) & (xx % 2 == 1); oe = (yy % 2 == 1) & (xx % 2 == 0); oo = (yy % 2 == 1) & (xx % 2 == 1)
    P[0][ee & act] = 1.0
    # horizontal edge points (even row, odd column): collapse the stencil vertically
    den = D - W[(-1, 0)] - W[(1, 0)]
    ok = eo & act & (den > 0)
    sden = np.where(den > 0, den, 1.0)
    P[0] = np.where(ok, (W[(0, -1)] + W[(-1, -1)] + W[(1, -1)]) / sden, P[0])
    P[1] = np.where(ok, (W[(0, 1)] + W[(-1, 1)] + W[(1, 1)]) / sden, P[1])
    # vertical edge points (odd row, even column)
    den = D - W[(0, -1)] - W[(0, 1)]
    ok = oe & act & (den > 0)
    sden = np.where(den > 0, den, 1.0)
    P[0] = np.where(ok, (W[(-1, 0)] + W[(-1, -1)] + W[(-1, 1)]) / sden, P[0])
    P[2] = np.where(ok, (W[(1, 0)] + W[(1, -1)] + W[(1, 1)]) / sden, P[2])
    # cell centres: the stencil equation with the edge neighbours replaced by their interpolants
    sD = np.where(D > 0, D, 1.0)
    n0, n1 = shift(P[0], -1, 0), shift(P[1], -1, 0)          # north neighbour (horizontal edge point): to c00, c01
    s0, s1 = shift(P[0], 1, 0), shift(P[1], 1, 0)            # south neighbour: its c00/c01 are OUR c10/c11
    w0, w2 = shift(P[0], 0, -1), shift(P[2], 0, -1)          # west neighbour (vertical edge point): to c00, c10
    e0, e2 = shift(P[0], 0, 1), shift(P[2], 0, 1)            # east neighbour: its c00/c10 are OUR c01/c11
    ok = oo & act
    P[0] = np.where(ok, (W[(-1, -1)] + W[(-1, 0)] * n0 + W[(0, -1)] * w0) / sD, P[0])
    P[1] = np.where(ok, (W[(-1, 1)] + W[(-1, 0)] * n1 + W[(0, 1)] * e0) / sD, P[1])
    P[2] = np.where(ok, (W[(1, -1)] + W[(1, 0)] * s0 + W[(0, -1)] * w2) / sD, P[2])
    P[3] = np.where(ok, (W[(1, 1)] + W[(1, 0)] * s1 + W[(0, 1)] * e2) / sD, P[3])
    return P


def sparse_A(L):
    E, S, SE, SW, D = L
    R, C = D.shape; n = R * C
    idx = np.arange(n).reshape(R, C)
    rows, cols, vals = [idx.ravel()], [idx.ravel()], [D.ravel()]
    for (dy, dx), w in (((0, 1), E), ((1, 0), S), ((1, 1), SE), ((1, -1), SW)):
        m = w != 0
        a = idx[m]; b = a + dy * C + dx
        rows += [a, b]; cols += [b, a]; vals += [-w[m], -w[m]]
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))


def sparse_P(P, shape_f, shape_c):
    R, C = shape_f; R2, C2 = shape_c
    yy, xx = np.mgrid[0:R, 0:C]
    rows, cols, vals = [], [], []
    for k, (di, dj) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
        ci, cj = yy // 2 + di, xx // 2 + dj
        m = (P[k] != 0) & (ci < R2) & (cj < C2)
        rows.append((yy * C + xx)[m]); cols.append((ci * C2 + cj)[m]); vals.append(P[k][m])
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(R * C, R2 * C2))


def galerkin(L, P):
    R, C = L[4].shape; R2, C2 = (R + 1) // 2, (C + 1) // 2
    Ps = sparse_P(P, (R, C), (R2, C2))
    Ac = (Ps.T @ sparse_A(L) @ Ps).tocsr()
    idx = np.arange(R2 * C2).reshape(R2, C2)
    def diag_of(dy, dx):
        out = np.zeros((R2, C2), L[4].dtype)
        ys = slice(max(0, -dy), min(R2, R2 - dy)); xs = slice(max(0, -dx), min(C2, C2 - dx))
        a = idx[ys, xs].ravel(); b = a + dy * C2 + dx
        out[ys, xs] = np.asarray(Ac[a, b]).reshape(out[ys, xs].shape)
        return out
    D = diag_of(0, 0)
    Lc = (-diag_of(0, 1), -diag_of(1, 0), -diag_of(1, 1), -diag_of(1, -1), D)
    # check: nothing outside the 9-point stencil
    nnz9 = sum((np.abs(a) > 0).sum() for a in Lc[:4]) * 2 + (D != 0).sum()
    assert Ac.count_nonzero() <= nnz9 + 0, (Ac.count_nonzero(), nnz9)
    return Lc, Ps


def gs4(L, x, b, n, order=(0, 1, 2, 3)):
    """4-colour Gauss-Seidel: colour = (y & 1) * 2 + (x & 1)."""
    E, S, SE, SW, D = L
    W = stencil8(L)
    R, C = D.shape
    yy, xx = np.mgrid[0:R, 0:C]
    col = (yy & 1) * 2 + (xx & 1)
    sD = np.where(D > 0, D, 1.0)
    for _ in range(n):
        for c in order:
            s = b.copy()
            for (dy, dx), w in W.items():
                s += w * shift(x, dy, dx)
            x = np.where((col == c) & (D > 0), s / sD, x)
    return x


def apply_A(L, x):
    W = stencil8(L)
    s = L[4] * x
    for (dy, dx), w in W.items():
        s -= w * shift(x, dy, dx)
    return s


FINE = None
OMEGA = float(os.environ.get('OMEGA', '1.0'))


def vcycle(levels, Ps, l, x, b, nu):
    L = levels[l]
    if l == len(levels) - 1:
        return gs4(L, x, b, int(os.environ.get('NCOARSE', '30')))
    x = gs4(L, x, b, nu) if (l > 0 or FINE is None) else rbgs_true(x, nu)
    r = np.where(L[4] > 0, b - apply_A(L, x), 0.0).astype(x.dtype)
    if l == 0 and FINE is not None:                                 # true fine operator, difference form (exact differences)
        wr, wd, free, x0 = FINE
        full = np.where(free, x, x0)
        r = np.zeros_like(x)
        for (dy, dx), w in (((0, -1), shift(wr, 0, -1)), ((0, 1), wr), ((-1, 0), shift(wd, -1, 0)), ((1, 0), wd)):
            r = r + w * (shift(full, dy, dx) - full)
        r = np.where(free, r, 0).astype(x.dtype)
    shape_c = levels[l + 1][4].shape
    rc = (Ps[l].T @ r.ravel()).reshape(shape_c)
    ec = vcycle(levels, Ps, l + 1, np.zeros(shape_c, x.dtype), rc, nu)
    x = x + (Ps[l] @ ec.ravel()).reshape(x.shape)
    return gs4(L, x, b, nu, order=(3, 2, 1, 0)) if (l > 0 or FINE is None) else rbgs_true(x, nu)


def rbgs_true(x, n):
    """Level 0: red-black Gauss-Seidel on the TRUE operator (all four LUT weights, Dirichlet values in place)."""
    wr, wd, free, x0 = FINE
    wl, wu = shift(wr, 0, -1), shift(wd, -1, 0)
    d = wl + wr + wu + wd
    sd = np.where(d > 0, d, 1)
    yy, xx = np.mgrid[0:x.shape[0], 0:x.shape[1]]
    full = np.where(free, x, x0)
    for _ in range(n):
        for c in (0, 1):
            s = wl * shift(full, 0, -1) + wr * shift(full, 0, 1) + wu * shift(full, -1, 0) + wd * shift(full, 1, 0)
            full = np.where(free & (((yy + xx) & 1) == c), np.clip(full + OMEGA * (np.clip(s / sd, 0, 255) - full), 0, 255), full).astype(x.dtype)
    return np.where(free, full, 0).astype(x.dtype)


if __name__ == "__main__":
    rows, cols, nu = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    cycles = int(sys.argv[4]) if len(sys.argv) > 4 else 15
    seed = int(sys.argv[5]) if len(sys.argv) > 5 else 1234
    p = make_problem(rows, cols, seed=seed)
    if os.environ.get("GOLDEN"):
        g = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", os.environ["GOLDEN"] + "_256.npz"))
        p = {"gray": g["gray0"], "mask": g["mask0"], "depth": g["depth_before_c1_L0"]}
    L0, (wr, wd, free) = fine_level(p["gray"], p["mask"])
    levels = [L0]; Ps = []
    t = time.time()
    while min(levels[-1][4].shape) > 8 and len(levels) < 12:
        P = build_P(levels[-1])
        Lc, Psp = galerkin(levels[-1], P)
        levels.append(Lc); Ps.append(Psp)
    print("levels", [l[4].shape for l in levels], "setup %.1fs" % (time.time() - t))
    for i, l in enumerate(levels[1:], 1):
        print("  level", i, "min offdiag weight", min(a.min() for a in l[:4]), "max", max(a.max() for a in l[:4]), "active", (l[4] > 0).mean())
    x0 = p["depth"].astype(L0[4].dtype)
    wl, wu = shift(wr, 0, -1), shift(wd, -1, 0)
    dsum = wl + wr + wu + wd
    def res(x):
        j = (wl * shift(x, 0, -1) + wr * shift(x, 0, 1) + wu * shift(x, -1, 0) + wd * shift(x, 1, 0)) / np.where(dsum > 0, dsum, 1)
        return np.abs(np.where(free, j - x, 0)).max()
    # error equation on the free points: b = W_fd x_d, unknown = x on free points
    xd = np.where(free, 0.0, x0)
    b = np.where(free, wl * shift(xd, 0, -1) + wr * shift(xd, 0, 1) + wu * shift(xd, -1, 0) + wd * shift(xd, 1, 0), 0.0)
    x = np.where(free, x0, 0.0)
    print("initial residual", res(np.where(free, x, x0)))
    if os.environ.get("TRUEFINE", "1") == "1":
        FINE = (wr, wd, free, x0)
    prev = None
    if os.environ.get("PCG"):
        A0 = lambda v: np.where(free, apply_A(L0, v), 0.0)
        M = lambda r: vcycle(levels, Ps, 0, np.zeros_like(r), r, nu)
        r = b - A0(x); z = M(r); pd = z.copy(); rz = (r * z).sum()
        for c in range(cycles):
            Ap = A0(pd); a = rz / (pd * Ap).sum()
            x = x + a * pd; r = r - a * Ap
            z = M(r); rz2 = (r * z).sum(); pd = z + (rz2 / rz) * pd; rz = rz2
            rr = res(np.where(free, x, x0))
            print(f"pcg {c + 1}: residual {rr:.3e}" + (f"  factor {rr / prev:.3f}" if prev else "") + f"  ({time.time() - t:.1f}s)", flush=True)
            prev = rr
            if rr < 1e-6: break
        sys.exit(0)
    for c in range(cycles):
        x = vcycle(levels, Ps, 0, x, b, nu)
        r = res(np.where(free, x, x0))
        print(f"cycle {c + 1}: residual {r:.3e}" + (f"  factor {r / prev:.3f}" if prev else "") + f"  ({time.time() - t:.1f}s)", flush=True)
        prev = r
        if r < 1e-6: break
    if os.environ.get("DIAG"):
        full = np.where(free, x, x0)
        j = (wl * shift(full, 0, -1) + wr * shift(full, 0, 1) + wu * shift(full, -1, 0) + wd * shift(full, 1, 0)) / np.where(dsum > 0, dsum, 1)
        rr = np.abs(np.where(free, j - full, 0))
        for _ in range(3):
            y, xx_ = np.unravel_index(np.argmax(rr), rr.shape)
            print("max residual at", y, xx_, rr[y, xx_])
            ys = slice(max(0, y - 4), y + 5); xs = slice(max(0, xx_ - 4), xx_ + 5)
            print(p["gray"][ys, xs]); print((p["mask"][ys, xs] == 255).astype(int)); print(np.round(rr[ys, xs] / rr.max(), 2))
            rr[ys, xs] = 0
