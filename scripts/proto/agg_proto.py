"""numpy / scipy prototype (float64, convergence study only; nothing here is built into the library): would ALGEBRAIC coarsening
converge the depth-diffusion system on photographs, where the structured (2x, operator-dependent interpolation) V-cycle stalls?
Pairwise aggregation along the strongest link, twice per level (aggregates of <= 4 pixels that follow fur and foliage instead
of the pixel grid), piecewise-constant interpolation, Galerkin coarse operators, symmetric Gauss-Seidel smoothing; used as a V-cycle,
as a K-cycle (two Krylov-accelerated coarse corrections per level, Notay's AGMG) and as the preconditioner of conjugate gradients.
usage: agg_proto.py NAME(Dog|Arara|WomanParasol|synthetic) [ROWS COLS]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def system(gray, mask, x0, beta=0.4):
    """A x = b on the free pixels (rows scaled as the solver has them: sum_j w_ij (x_i - x_j) = 0, Dirichlet values on the right)."""
    g = gray.astype(np.int64); lut = np.exp(-beta * np.arange(256)).astype(np.float32).astype(np.float64)
    R, C = g.shape; n = R * C
    idx = np.arange(n).reshape(R, C)
    wr = lut[np.abs(g[:, 1:] - g[:, :-1])]; wd = lut[np.abs(g[1:, :] - g[:-1, :])]
    i = np.concatenate([idx[:, :-1].ravel(), idx[:-1, :].ravel()]); j = np.concatenate([idx[:, 1:].ravel(), idx[1:, :].ravel()])
    w = np.concatenate([wr.ravel(), wd.ravel()])
    W = sp.coo_matrix((np.concatenate([w, w]), (np.concatenate([i, j]), np.concatenate([j, i]))), shape=(n, n)).tocsr()
    L = sp.diags(np.asarray(W.sum(1)).ravel()) - W
    free = (mask != 255).ravel()
    f = np.flatnonzero(free); d = np.flatnonzero(~free)
    A = L[f][:, f].tocsr(); b = -L[f][:, d] @ x0.ravel()[d].astype(np.float64)
    return A, b, f, W, free


def pairwise(A, theta=0.25):
    """One pass of pairwise aggregation: greedy matching along the strongest negative coupling (>= theta x the row's strongest)."""
    n = A.shape[0]
    S = -A.copy(); S.setdiag(0); S.eliminate_zeros(); S.data = np.maximum(S.data, 0)
    mx = np.maximum(S.max(1).toarray().ravel(), 1e-300)
    agg = -np.ones(n, np.int64)
    indptr, indices, data = S.indptr, S.indices, S.data
    # visit vertices by decreasing strongest link (a cheap stand-in for Notay's ordering)
    order = np.argsort(-mx, kind="stable")
    na = 0
    for i in order:
        if agg[i] >= 0: continue
        best, bw = -1, 0.0
        for k in range(indptr[i], indptr[i + 1]):
            j = indices[k]
            if agg[j] < 0 and data[k] >= theta * mx[i] and data[k] > bw: best, bw = j, data[k]
        agg[i] = na
        if best >= 0: agg[best] = na
        na += 1
    P = sp.csr_matrix((np.ones(n), (np.arange(n), agg)), shape=(n, na))
    return P


def hierarchy(A, passes=2, min_n=400, max_levels=30):
    levels = [A]; Ps = []
    while levels[-1].shape[0] > min_n and len(levels) < max_levels:
        Ak = levels[-1]; P = None
        for _ in range(passes):
            Pk = pairwise(Ak)
            Ak = (Pk.T @ Ak @ Pk).tocsr()
            P = Pk if P is None else (P @ Pk).tocsr()
        if Ak.shape[0] > 0.8 * levels[-1].shape[0]: break
        levels.append(Ak); Ps.append(P)
    return levels, Ps


class MG:
    def __init__(self, levels, Ps, cycle="V", nu=1):
        self.A = levels; self.P = Ps; self.cycle = cycle; self.nu = nu
        self.Lo = [sp.tril(a, format="csr") for a in levels]; self.Up = [sp.triu(a, format="csr") for a in levels]
        self.D = [a.diagonal() for a in levels]
        self.coarse = spla.splu(levels[-1].tocsc())
        self.work = 0.0

    def smooth(self, l, x, b, forward):
        A = self.A[l]
        for _ in range(self.nu):
            r = b - A @ x
            x = x + spla.spsolve_triangular(self.Lo[l] if forward else self.Up[l], r, lower=forward)
            self.work += A.nnz
        return x

    def solve(self, l, b):
        if l == len(self.A) - 1:
            return self.coarse.solve(b)
        x = self.smooth(l, np.zeros_like(b), b, True)
        r = b - self.A[l] @ x
        rc = self.P[l].T @ r
        if self.cycle == "K" and l + 1 < len(self.A) - 1:
            # two steps of flexible CG on the coarse problem, preconditioned by the cycle below (Notay)
            Ac = self.A[l + 1]
            c1 = self.solve(l + 1, rc); v1 = Ac @ c1; rho1 = c1 @ v1; a1 = c1 @ rc
            r2 = rc - (a1 / rho1) * v1
            if np.linalg.norm(r2) <= 0.25 * np.linalg.norm(rc):
                ec = (a1 / rho1) * c1
            else:
                c2 = self.solve(l + 1, r2); v2 = Ac @ c2; g = c2 @ v1; be = c2 @ v2; a2 = c2 @ r2
                rho2 = be - g * g / rho1
                ec = (a1 / rho1 - g * a2 / (rho1 * rho2)) * c1 + (a2 / rho2) * c2
        else:
            ec = self.solve(l + 1, rc)
        x = x + self.P[l] @ ec
        return self.smooth(l, x, b, False)


def resid(A, W, free, fidx, x0, xf):
    """the solver's own stopping quantity: max |Jacobi update| over the free pixels (grey levels)"""
    full = x0.ravel().astype(np.float64).copy(); full[fidx] = xf
    j = (W @ full) / np.maximum(np.asarray(W.sum(1)).ravel(), 1e-300)
    return np.abs((j - full)[free]).max()


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "Dog"
    if name == "synthetic":
        from realtimedepthdiffusion_amd.synth import make_problem
        p = make_problem(int(sys.argv[2]), int(sys.argv[3]), seed=1234)
        gray, mask, x0 = p["gray"], p["mask"], p["depth"]
    else:
        g = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", name + "_256.npz"))
        gray, mask, x0 = g["gray0"], g["mask0"], g["depth_before_c1_L0"]
    t = time.time()
    A, b, fidx, W, free = system(gray, mask, x0)
    levels, Ps = hierarchy(A)
    print(f"{name}: {A.shape[0]} unknowns, levels {[a.shape[0] for a in levels]}, operator complexity {sum(a.nnz for a in levels) / A.nnz:.2f}, setup {time.time() - t:.1f}s")
    xs = x0.ravel()[fidx].astype(np.float64)
    print(f"initial residual {resid(A, W, free, fidx, x0, xs):.3e}")
    for cyc in ("V", "K"):
        mg = MG(levels, Ps, cyc)
        # (a) stationary cycles
        x = xs.copy(); prev = None
        for c in range(40):
            x = x + mg.solve(0, b - A @ x)
            r = resid(A, W, free, fidx, x0, x)
            if c % 5 == 4 or r < 1e-4: print(f"  {cyc}-cycle {c + 1:3d}: residual {r:.3e}" + (f" factor {(r / prev) ** 0.2:.3f}" if prev and c % 5 == 4 else ""))
            if c % 5 == 4: prev = r
            if r < 1e-4: break
        # (b) conjugate gradients preconditioned by one cycle (flexible form)
        x = xs.copy(); r = b - A @ x; z = mg.solve(0, r); pd = z.copy(); rz = r @ z
        for c in range(100):
            Ap = A @ pd; a = rz / (pd @ Ap); x = x + a * pd; rn = r - a * Ap
            z = mg.solve(0, rn); rz2 = z @ (rn - r) ; beta = rz2 / rz; rz = z @ rn; r = rn; pd = z + beta * pd
            rr = resid(A, W, free, fidx, x0, x)
            if c % 5 == 4 or rr < 1e-4: print(f"  CG + {cyc}-cycle {c + 1:3d}: residual {rr:.3e}")
            if rr < 1e-4: break
