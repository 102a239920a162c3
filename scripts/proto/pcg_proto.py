"""numpy prototype: conjugate gradients on the depth-diffusion system, preconditioned by one symmetric
aggregation-multigrid V-cycle (damped-Jacobi smoothing).  Convergence study only (float64)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mg_proto import fine_level, weights4, neighbours, coarsen, restrict, prolong
from realtimedepthdiffusion_amd.synth import make_problem

def apply_A(x, wr, wd, diag, free):          # A restricted to free nodes (x = 0 elsewhere)
    wl, wr_, wu, wd_ = weights4(wr, wd); x = np.where(free, x, 0.0)
    xl, xr, xu, xd = neighbours(x)
    return np.where(free, diag * x - (wl * xl + wr_ * xr + wu * xu + wd_ * xd), 0.0)

def jacobi(x, b, lv, n, om=0.8):
    wr, wd, diag, free = lv
    dinv = np.where(free & (diag > 0), 1.0 / np.where(diag > 0, diag, 1), 0.0)
    for _ in range(n):
        x = x + om * dinv * (b - apply_A(x, wr, wd, diag, free))
    return x

def vcycle(lv, l, b, nu, alpha):
    wr, wd, diag, free = lv[l]
    if l == len(lv) - 1:
        return jacobi(np.zeros_like(b), b, lv[l], 30)
    x = jacobi(np.zeros_like(b), b, lv[l], nu)
    r = b - apply_A(x, wr, wd, diag, free)
    ec = vcycle(lv, l + 1, restrict(r, lv[l + 1][0].shape), nu, alpha)
    x = x + np.where(free, alpha * prolong(ec, x.shape), 0.0)
    return jacobi(x, b, lv[l], nu)

if __name__ == "__main__":
    rows, cols = int(sys.argv[1]), int(sys.argv[2]); nu = int(sys.argv[3]); nlev = int(sys.argv[4]); alpha = float(sys.argv[5])
    p = make_problem(rows, cols, seed=1234)
    wr, wd, free = fine_level(p["gray"], p["mask"])
    wl, _, wu, _ = weights4(wr, wd); diag = wl + wr + wu + wd
    lv = [(wr, wd, diag, free)]
    for _ in range(nlev - 1): lv.append(coarsen(*lv[-1]))
    x0 = p["depth"].astype(np.float64)
    # rhs from Dirichlet values: A_ff x_f = W_fd x_d
    xd = np.where(free, 0.0, x0); xl, xr, xu, xdn = neighbours(xd)
    b = np.where(free, wl * xl + wr * xr + wu * xu + wd * xdn, 0.0)
    def res(x):
        full = np.where(free, x, x0); xl, xr, xu, xd_ = neighbours(full)
        j = (wl * xl + wr * xr + wu * xu + wd * xd_) / np.where(diag > 0, diag, 1)
        return np.abs(np.where(free, j - full, 0)).max()
    x = np.where(free, x0, 0.0)
    r = b - apply_A(x, *lv[0]); z = vcycle(lv, 0, r, nu, alpha); pdir = z.copy(); rz = (r * z).sum()
    print("levels", [l[0].shape for l in lv], "initial residual", res(x))
    t = time.time()
    for it in range(200):
        Ap = apply_A(pdir, *lv[0]); a = rz / (pdir * Ap).sum()
        x += a * pdir; r -= a * Ap
        z = vcycle(lv, 0, r, nu, alpha); rz2 = (r * z).sum(); pdir = z + (rz2 / rz) * pdir; rz = rz2
        rr = res(x)
        if it % 5 == 4 or rr < 1e-4: print(f"iter {it + 1}: max|J(x)-x| {rr:.3e}  ({time.time() - t:.1f}s)", flush=True)
        if rr < 1e-4: break
