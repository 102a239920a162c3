// multigrid.hip -- multigrid V-cycle for the depth-diffusion system (EXTENSION: the reference has no such solver;
// BASELINE config 5).  Black-box multigrid in Dendy's sense: standard 2x coarsening (coarse point (I,J) = fine point
// (2I,2J)), OPERATOR-DEPENDENT interpolation read off the stencil, Galerkin coarse operators (9-point from level 1 on).
//
// Level 0 is the image: unknown x on the free pixels, smoothed by the register-blocked red-black Gauss-Seidel kernel
// (rbgs_blocked.hip) on the TRUE operator (LUT weights, Dirichlet values in place); its residual is formed from exact
// differences, r_p = sum_q w_pq (x_q - x_p), so it carries no cancellation noise.  The hierarchy below it is built from
// a copy of the level-0 operator in which links weaker than kTheta are kept on the diagonal only (an anchor to e = 0):
// that bounds the condition of every coarse equation in f32 and changes the preconditioner, not the solution.
// Levels >= 1 hold a symmetric 9-point stencil as four couplings per point (E, S, SE, SW: to the right / lower /
// lower-right / lower-left neighbour; the other four are the neighbours' own) and the diagonal D; D == 0 marks an
// inactive point (e = 0 for ever).  P holds, per fine point, its interpolation weights towards the four coarse points
// at the corners of its coarse cell.  Coarse smoother: four-colour Gauss-Seidel (colour = (y&1)*2 + (x&1)), one launch per
// smoothing step: LDS tiles with a halo for large levels, the whole level in LDS for small ones.
//
// Every point is computed with a fixed evaluation order and no atomics (the smoothers recompute halo points rather than
// exchange them): oracle/rtdd_mg_oracle.c restates the same arithmetic and the parity tests compare bit for bit.
#include <new>

#include "rtdd_internal.hpp"

namespace rtdd {

constexpr float kTheta = 1e-4f;       // hierarchy only: weaker links become anchors
#ifndef RTDD_MG_NU
#define RTDD_MG_NU 2
#endif
constexpr int kNu = RTDD_MG_NU;       // pre- and post-smoothing sweeps on every level (the oracle restates 2; other values: experiments only)
constexpr int kCoarsestSweeps = 30;
constexpr int kSmallLevel = 16384;    // fallback single-workgroup smoother out of global memory (not used by the cycle as configured)

struct MgLevel {
    int rows = 0, cols = 0, pitch = 0;
    float *buf = nullptr;             // 12 planes: E S SE SW D P0 P1 P2 P3 e b r
    size_t plane = 0;
    float *A(int i) const { return buf + (size_t)i * plane; }
    float *E() const { return A(0); }
    float *S() const { return A(1); }
    float *SE() const { return A(2); }
    float *SW() const { return A(3); }
    float *D() const { return A(4); }
    float *Pw(int k) const { return A(5 + k); }
    int ei = 9, ri = 11;              // e and r trade places after every tiled smoothing step (it writes e into r's plane)
    float *e() const { return A(ei); }
    float *b() const { return A(10); }
    float *r() const { return A(ri); }
};

struct MgState {
    int rows = 0, cols = 0;
    std::vector<MgLevel> lv;
    void release() {
        for (auto &l : lv) if (l.buf) (void)hipFree(l.buf);
        lv.clear(); rows = cols = 0;
    }
};

struct Stencil {                      // device view of one level
    const float *E, *S, *SE, *SW, *D;
    int rows, cols, pitch;
};

namespace {

__device__ __forceinline__ float at(const float *a, int pitch, int rows, int cols, int y, int x) {
    return (y >= 0 && y < rows && x >= 0 && x < cols) ? a[(size_t)y * pitch + x] : 0.0f;
}

// coupling of point (y,x) towards (y+dy, x+dx), (dy,dx) != (0,0)
__device__ __forceinline__ float coupling(const Stencil &s, int y, int x, int dy, int dx) {
    if (dy == 0) return dx > 0 ? at(s.E, s.pitch, s.rows, s.cols, y, x) : at(s.E, s.pitch, s.rows, s.cols, y, x - 1);
    if (dy > 0) {
        if (dx == 0) return at(s.S, s.pitch, s.rows, s.cols, y, x);
        return dx > 0 ? at(s.SE, s.pitch, s.rows, s.cols, y, x) : at(s.SW, s.pitch, s.rows, s.cols, y, x);
    }
    if (dx == 0) return at(s.S, s.pitch, s.rows, s.cols, y - 1, x);
    return dx < 0 ? at(s.SE, s.pitch, s.rows, s.cols, y - 1, x - 1) : at(s.SW, s.pitch, s.rows, s.cols, y - 1, x + 1);
}

// level-0 hierarchy operator from the packed metadata
__global__ __launch_bounds__(256) void k_mg_stencil0(const uint32_t *__restrict__ M, const float *__restrict__ lut_g, int ip, int rows, int cols,
                                                     float *E, float *S, float *SE, float *SW, float *D, int pitch, float theta) {
    __shared__ float lut[257];
    for (int i = threadIdx.x; i < 257; i += 256) lut[i] = lut_g[i];
    __syncthreads();
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + wave_id();
    if (x >= cols || y >= rows) return;
    const size_t p = (size_t)y * ip + x;
    const uint32_t m = M[p];
    const bool fr = !(m & kMetaDirichlet);
    const float wr = x + 1 < cols ? lut[m & 255] : 0.0f;
    const float wd = y + 1 < rows ? lut[(m >> 8) & 255] : 0.0f;
    const float wl = x > 0 ? lut[M[p - 1] & 255] : 0.0f;
    const float wu = y > 0 ? lut[(M[p - ip] >> 8) & 255] : 0.0f;
    float d = 0.0f;
    d += wl; d += wr; d += wu; d += wd;
    const bool rfree = x + 1 < cols && !(M[p + 1] & kMetaDirichlet);
    const bool dfree = y + 1 < rows && !(M[p + ip] & kMetaDirichlet);
    const size_t q = (size_t)y * pitch + x;
    E[q] = (fr && rfree && wr >= theta) ? wr : 0.0f;
    S[q] = (fr && dfree && wd >= theta) ? wd : 0.0f;
    SE[q] = 0.0f; SW[q] = 0.0f;
    D[q] = fr ? d : 0.0f;
}

// interpolation weights of coarse-coincident and edge points; cell centres in the second pass
template <int PASS>
__global__ __launch_bounds__(256) void k_mg_build_p(Stencil s, float *P0, float *P1, float *P2, float *P3) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + wave_id();
    if (x >= s.cols || y >= s.rows) return;
    const size_t q = (size_t)y * s.pitch + x;
    const float d = s.D[q];
    const bool act = d > 0.0f;
    const bool oy = y & 1, ox = x & 1;
    if (PASS == 0) {
        float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f;
        if (!oy && !ox) p0 = act ? 1.0f : 0.0f;
        else if (!oy && ox) {                 // on a coarse row, between two coarse points: collapse the stencil vertically
            const float den = (d - coupling(s, y, x, -1, 0)) - coupling(s, y, x, 1, 0);
            if (act && den > 0.0f) {
                p0 = ((coupling(s, y, x, 0, -1) + coupling(s, y, x, -1, -1)) + coupling(s, y, x, 1, -1)) / den;
                p1 = ((coupling(s, y, x, 0, 1) + coupling(s, y, x, -1, 1)) + coupling(s, y, x, 1, 1)) / den;
            }
        } else if (oy && !ox) {               // on a coarse column: collapse horizontally
            const float den = (d - coupling(s, y, x, 0, -1)) - coupling(s, y, x, 0, 1);
            if (act && den > 0.0f) {
                p0 = ((coupling(s, y, x, -1, 0) + coupling(s, y, x, -1, -1)) + coupling(s, y, x, -1, 1)) / den;
                p2 = ((coupling(s, y, x, 1, 0) + coupling(s, y, x, 1, -1)) + coupling(s, y, x, 1, 1)) / den;
            }
        }
        P0[q] = p0; P1[q] = p1; P2[q] = p2; P3[q] = 0.0f;
    } else {
        if (!(oy && ox) || !act) return;      // cell centre: its own equation with the edge neighbours replaced by their interpolants
        const float wn = coupling(s, y, x, -1, 0), ws = coupling(s, y, x, 1, 0), ww = coupling(s, y, x, 0, -1), we = coupling(s, y, x, 0, 1);
        const float n0 = at(P0, s.pitch, s.rows, s.cols, y - 1, x), n1 = at(P1, s.pitch, s.rows, s.cols, y - 1, x);
        const float s0 = at(P0, s.pitch, s.rows, s.cols, y + 1, x), s1 = at(P1, s.pitch, s.rows, s.cols, y + 1, x);
        const float w0 = at(P0, s.pitch, s.rows, s.cols, y, x - 1), w2 = at(P2, s.pitch, s.rows, s.cols, y, x - 1);
        const float e0 = at(P0, s.pitch, s.rows, s.cols, y, x + 1), e2 = at(P2, s.pitch, s.rows, s.cols, y, x + 1);
        P0[q] = ((coupling(s, y, x, -1, -1) + wn * n0) + ww * w0) / d;
        P1[q] = ((coupling(s, y, x, -1, 1) + wn * n1) + we * e0) / d;
        P2[q] = ((coupling(s, y, x, 1, -1) + ws * s0) + ww * w2) / d;
        P3[q] = ((coupling(s, y, x, 1, 1) + ws * s1) + we * e2) / d;
    }
}

struct Interp { const float *P0, *P1, *P2, *P3; };

// weight of fine point (y,x) towards coarse point (I,J)
__device__ __forceinline__ float pweight(const Interp &ip, int pitch, int rows, int cols, int y, int x, int I, int J) {
    if (y < 0 || y >= rows || x < 0 || x >= cols) return 0.0f;
    const int di = I - (y >> 1), dj = J - (x >> 1);
    if (di < 0 || di > 1 || dj < 0 || dj > 1) return 0.0f;
    const float *P = di ? (dj ? ip.P3 : ip.P2) : (dj ? ip.P1 : ip.P0);
    return P[(size_t)y * pitch + x];
}

// Galerkin coarse operator A_c = P^T A P, one coarse point c = (I,J) per thread:
//   A_c[c, c'] = sum_p sum_q P[p -> c] A[p, q] P[q -> c'],  p over the 3x3 support of c's basis function, q over p's stencil.
// Each (p, q) pair feeds the up to four coarse points q interpolates from; with the loops unrolled every offset is a
// compile-time constant.  For one target c' the terms arrive in (py, px, qy, qx) order -- the order of the restatement's
// plain loops (terms with a zero weight add +-0 and change nothing).  FIVE: the fine operator is the image's 5-point one
// (its diagonal couplings are all zero), so the four diagonal q are skipped -- again only +-0 terms.
template <bool FIVE>
__global__ __launch_bounds__(256) void k_mg_galerkin(Stencil f, Interp ip, float *E, float *S, float *SE, float *SW, float *D, int crows, int ccols, int cpitch) {
    const int J = blockIdx.x * 64 + (threadIdx.x & 63), I = blockIdx.y * 4 + wave_id();
    if (J >= ccols || I >= crows) return;
    float out[3][3] = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};      // [dI + 1][dJ + 1]
#pragma unroll
    for (int py = -1; py <= 1; py++)
#pragma unroll
        for (int px = -1; px <= 1; px++) {
            const int y = 2 * I + py, x = 2 * J + px;
            const float wp = pweight(ip, f.pitch, f.rows, f.cols, y, x, I, J);
            if (wp == 0.0f) continue;
#pragma unroll
            for (int qy = -1; qy <= 1; qy++)
#pragma unroll
                for (int qx = -1; qx <= 1; qx++) {
                    if (FIVE && qy != 0 && qx != 0) continue;
                    const int v = y + qy, u = x + qx;
                    if (v < 0 || v >= f.rows || u < 0 || u >= f.cols) continue;
                    const float a = (qy == 0 && qx == 0) ? f.D[(size_t)y * f.pitch + x] : -coupling(f, y, x, qy, qx);
                    const size_t qo = (size_t)v * f.pitch + u;
                    // q = 2c + (py+qy, px+qx): its coarse cell starts at c + floor((py+qy)/2), floor((px+qx)/2)
                    const int bi = (py + qy) >> 1, bj = (px + qx) >> 1;
                    const float w4[4] = {ip.P0[qo], ip.P1[qo], ip.P2[qo], ip.P3[qo]};
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int di = bi + (k >> 1), dj = bj + (k & 1);
                        if (di < -1 || di > 1 || dj < -1 || dj > 1) continue;
                        out[di + 1][dj + 1] += wp * (a * w4[k]);
                    }
                }
        }
    const size_t q = (size_t)I * cpitch + J;
    const bool act = out[1][1] > 0.0f;
    // targets outside the coarse grid received only zero weights: their sums are +0 and the stored couplings -0
    D[q] = act ? out[1][1] : 0.0f;
    E[q] = act ? -out[1][2] : 0.0f; S[q] = act ? -out[2][1] : 0.0f; SE[q] = act ? -out[2][2] : 0.0f; SW[q] = act ? -out[2][0] : 0.0f;
}

// a coupling towards an inactive point would read e = 0 for ever: harmless; but a coupling FROM an inactive point was
// zeroed above while its mirror is read through the neighbour -- so zero the couplings that END at an inactive point too
__global__ __launch_bounds__(256) void k_mg_prune(float *E, float *S, float *SE, float *SW, const float *D, int rows, int cols, int pitch) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + wave_id();
    if (x >= cols || y >= rows) return;
    const size_t q = (size_t)y * pitch + x;
    if (!(at(D, pitch, rows, cols, y, x + 1) > 0.0f)) E[q] = 0.0f;
    if (!(at(D, pitch, rows, cols, y + 1, x) > 0.0f)) S[q] = 0.0f;
    if (!(at(D, pitch, rows, cols, y + 1, x + 1) > 0.0f)) SE[q] = 0.0f;
    if (!(at(D, pitch, rows, cols, y + 1, x - 1) > 0.0f)) SW[q] = 0.0f;
}

// level-0 residual from exact differences, on the true operator; 4 pixels per thread (16-byte loads)
__global__ __launch_bounds__(256) void k_mg_residual0(const float *__restrict__ X, const uint32_t *__restrict__ M, const float *__restrict__ lut_g,
                                                      int ip, int rows, int cols, float *R, int pitch) {
    __shared__ float lut[257];
    for (int i = threadIdx.x; i < 257; i += 256) lut[i] = lut_g[i];
    __syncthreads();
    const int x0 = 4 * (blockIdx.x * 64 + (threadIdx.x & 63)), y = blockIdx.y * 4 + wave_id();
    if (x0 >= cols || y >= rows) return;
    const size_t p = (size_t)y * ip + x0;                          // the solver planes' guard rows/columns make these addresses valid
    const float4 c4 = *(const float4 *)(X + p), u4 = *(const float4 *)(X + p - ip), d4 = *(const float4 *)(X + p + ip);
    const uint4 m4 = *(const uint4 *)(M + p), mu4 = *(const uint4 *)(M + p - ip);
    const float xc[4] = {c4.x, c4.y, c4.z, c4.w}, xu[4] = {u4.x, u4.y, u4.z, u4.w}, xd[4] = {d4.x, d4.y, d4.z, d4.w};
    const uint32_t mm[4] = {m4.x, m4.y, m4.z, m4.w}, mu[4] = {mu4.x, mu4.y, mu4.z, mu4.w};
    const float xl0 = X[p - 1], xr4 = X[p + 4];
    const uint32_t ml0 = M[p - 1];
    float r[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int x = x0 + i;
        float v = 0.0f;
        if (x < cols && !(mm[i] & kMetaDirichlet)) {
            if (x > 0) v += lut[(i == 0 ? ml0 : mm[i - 1]) & 255] * ((i == 0 ? xl0 : xc[i - 1]) - xc[i]);
            if (x + 1 < cols) v += lut[mm[i] & 255] * ((i == 3 ? xr4 : xc[i + 1]) - xc[i]);
            if (y > 0) v += lut[(mu[i] >> 8) & 255] * (xu[i] - xc[i]);
            if (y + 1 < rows) v += lut[(mm[i] >> 8) & 255] * (xd[i] - xc[i]);
        }
        r[i] = v;
    }
    float *o = R + (size_t)y * pitch + x0;                           // pitch is a multiple of 64: whole float4s stay inside the row
    if (x0 + 3 < pitch) *(float4 *)o = make_float4(r[0], r[1], r[2], r[3]);
}

__device__ __forceinline__ float gs_sum(const Stencil &s, const float *e, const float *b, int y, int x) {
    float v = b[(size_t)y * s.pitch + x];
    v += coupling(s, y, x, 0, -1) * at(e, s.pitch, s.rows, s.cols, y, x - 1);
    v += coupling(s, y, x, 0, 1) * at(e, s.pitch, s.rows, s.cols, y, x + 1);
    v += coupling(s, y, x, -1, 0) * at(e, s.pitch, s.rows, s.cols, y - 1, x);
    v += coupling(s, y, x, 1, 0) * at(e, s.pitch, s.rows, s.cols, y + 1, x);
    v += coupling(s, y, x, -1, -1) * at(e, s.pitch, s.rows, s.cols, y - 1, x - 1);
    v += coupling(s, y, x, -1, 1) * at(e, s.pitch, s.rows, s.cols, y - 1, x + 1);
    v += coupling(s, y, x, 1, -1) * at(e, s.pitch, s.rows, s.cols, y + 1, x - 1);
    v += coupling(s, y, x, 1, 1) * at(e, s.pitch, s.rows, s.cols, y + 1, x + 1);
    return v;
}

__global__ __launch_bounds__(256) void k_mg_residual(Stencil s, const float *e, const float *b, float *R) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + wave_id();
    if (x >= s.cols || y >= s.rows) return;
    const size_t q = (size_t)y * s.pitch + x;
    const float d = s.D[q];
    R[q] = d > 0.0f ? gs_sum(s, e, b, y, x) - d * e[q] : 0.0f;
}

// one colour of the four-colour Gauss-Seidel sweep; the launch covers that colour's quarter grid
__global__ __launch_bounds__(256) void k_mg_gs(Stencil s, float *e, const float *b, int colour) {
    const int x = 2 * (blockIdx.x * 64 + (threadIdx.x & 63)) + (colour & 1), y = 2 * (blockIdx.y * 4 + wave_id()) + (colour >> 1);
    if (x >= s.cols || y >= s.rows) return;
    const size_t q = (size_t)y * s.pitch + x;
    const float d = s.D[q];
    if (d > 0.0f) e[q] = gs_sum(s, e, b, y, x) / d;
}

// small levels: all sweeps of a smoothing step in one launch of one workgroup
__global__ __launch_bounds__(1024) void k_mg_gs_small(Stencil s, float *e, const float *b, int nsweeps, int reverse) {
    const int hr = (s.rows + 1) >> 1, hc = (s.cols + 1) >> 1;
    for (int sw = 0; sw < nsweeps; sw++)
        for (int c = 0; c < 4; c++) {
            const int colour = reverse ? 3 - c : c;
            for (int i = threadIdx.x; i < hr * hc; i += 1024) {
                const int y = 2 * (i / hc) + (colour >> 1), x = 2 * (i % hc) + (colour & 1);
                if (x < s.cols && y < s.rows) {
                    const size_t q = (size_t)y * s.pitch + x;
                    const float d = s.D[q];
                    if (d > 0.0f) e[q] = gs_sum(s, e, b, y, x) / d;
                }
            }
            __syncthreads();
        }
}

// levels of at most PPT * 1024 points whose iterate (plus a ring of zeros) fits in 62 KB of LDS: the same smoothing step
// with the iterate in LDS and every thread's points' couplings, diagonal and right-hand side in registers, so that a colour
// step costs an LDS round trip and a barrier instead of a chain of global loads (8 160 points: 42 -> 6 us; the 30
// coarsest sweeps: 123 -> 12 us).  Same sums in the same order as gs_sum: a missing neighbour is (coupling 0, value 0).
template <int PPT>
__global__ __launch_bounds__(1024) void k_mg_gs_lds(Stencil s, float *e, const float *b, int nsweeps, int reverse) {
    extern __shared__ float le[];                       // (rows + 2) x (cols + 2)
    const int R = s.rows, C = s.cols, LP = C + 2, tid = threadIdx.x;
    for (int i = tid; i < (R + 2) * LP; i += 1024) le[i] = 0.0f;
    __syncthreads();
    for (int i = tid; i < R * C; i += 1024) le[(i / C + 1) * LP + i % C + 1] = e[(size_t)(i / C) * s.pitch + i % C];
    float cw[PPT][8], dg[PPT], rb[PPT];
    int at_[PPT], col[PPT];
#pragma unroll
    for (int k = 0; k < PPT; k++) {
        const int i = tid + k * 1024;
        const bool in = i < R * C;
        const int y = in ? i / C : 0, x = in ? i % C : 0;
        at_[k] = (y + 1) * LP + x + 1;
        col[k] = in ? (y & 1) * 2 + (x & 1) : -1;
        dg[k] = in ? s.D[(size_t)y * s.pitch + x] : 0.0f;
        rb[k] = in ? b[(size_t)y * s.pitch + x] : 0.0f;
        cw[k][0] = coupling(s, y, x, 0, -1); cw[k][1] = coupling(s, y, x, 0, 1); cw[k][2] = coupling(s, y, x, -1, 0); cw[k][3] = coupling(s, y, x, 1, 0);
        cw[k][4] = coupling(s, y, x, -1, -1); cw[k][5] = coupling(s, y, x, -1, 1); cw[k][6] = coupling(s, y, x, 1, -1); cw[k][7] = coupling(s, y, x, 1, 1);
        if (!(dg[k] > 0.0f)) col[k] = -1;               // inactive points are never updated
    }
    __syncthreads();
    for (int sw = 0; sw < nsweeps; sw++)
        for (int c = 0; c < 4; c++) {
            const int colour = reverse ? 3 - c : c;
#pragma unroll
            for (int k = 0; k < PPT; k++)
                if (col[k] == colour) {
                    const int a = at_[k];
                    float v = rb[k];
                    v += cw[k][0] * le[a - 1];
                    v += cw[k][1] * le[a + 1];
                    v += cw[k][2] * le[a - LP];
                    v += cw[k][3] * le[a + LP];
                    v += cw[k][4] * le[a - LP - 1];
                    v += cw[k][5] * le[a - LP + 1];
                    v += cw[k][6] * le[a + LP - 1];
                    v += cw[k][7] * le[a + LP + 1];
                    le[a] = v / dg[k];
                }
            __syncthreads();
        }
#pragma unroll
    for (int k = 0; k < PPT; k++) {
        const int i = tid + k * 1024;
        if (i < R * C) e[(size_t)(i / C) * s.pitch + i % C] = le[at_[k]];
    }
}

// larger levels: the same smoothing step tile by tile.  A workgroup takes a 64 x 32 tile plus a halo of 4 points per sweep
// into LDS and runs ALL the colour steps of the smoothing step on it; what a missing outer neighbour spoils moves inwards one
// point per colour step and stops short of the tile, so the tile's own points end up exactly as a level-wide sweep leaves
// them.  One launch instead of 4 per sweep.  Tiles read each other's points, so the result goes to a second plane.
//
// Data movement (round 2): a thread owns two groups of 4 consecutive points and fetches E, S, SE, SW, D, b and e of a group as
// one 16-byte load each -- 7 coalesced loads per group instead of 11 bounds-checked scalar gathers per POINT.  The other four
// couplings of a point are its neighbours' own (W = the left point's E, N = the upper point's S, NW = the upper-left point's SE,
// NE = the upper-right point's SW): the threads trade them through two LDS staging planes (E and S, then SE and SW).  A
// neighbour outside the extended tile reads as coupling 0, which only touches the outermost ring -- spoilt from the first step
// anyway.  512 threads and ~100 registers: two workgroups per CU, so one loads while the other relaxes.
constexpr int kTileX = 64, kTileY = 32, kTileThreads = 512;
__global__ __launch_bounds__(kTileThreads, 4) void k_mg_gs_tile(Stencil s, const float *e, float *out, const float *b, int nsweeps, int reverse) {
    constexpr int GPT = 2;                               // groups of 4 points per thread
    typedef float f4 __attribute__((ext_vector_type(4)));
    extern __shared__ float lds[];
    const int H = 4 * nsweeps, EW = kTileX + 2 * H, EH = kTileY + 2 * H, LP = EW + 2, GW = EW / 4, tid = threadIdx.x;
    const int plane = (EH + 2) * LP;
    float *le = lds, *st0 = lds + plane, *st1 = lds + 2 * plane;
    const int ox = blockIdx.x * kTileX - H, oy = blockIdx.y * kTileY - H;
    for (int i = tid; i < 3 * plane; i += kTileThreads) lds[i] = 0.0f;
    __syncthreads();
    float cw[GPT][4][8], dg[GPT][4], rb[GPT][4];         // dg == 0 marks a point that is never updated (inactive, or outside the level)
    int at_[GPT];                                        // LDS index of the group's first point
#pragma unroll
    for (int k = 0; k < GPT; k++) {
        const int gi = tid + k * kTileThreads;
        const int ly = gi / GW, lx = 4 * (gi % GW), y = oy + ly, x0 = ox + lx;
        const bool row = gi < GW * EH && y >= 0 && y < s.rows && x0 >= 0 && x0 < s.cols;     // x0 and the pitch are multiples of 4: the 16-byte loads stay inside the row
        at_[k] = gi < GW * EH ? (ly + 1) * LP + lx + 1 : LP + 1;
        const f4 z = {0.0f, 0.0f, 0.0f, 0.0f};
        f4 vd = z, vb = z, ve = z, pe = z, ps = z, pse = z, psw = z;
        if (row) {
            const size_t q = (size_t)y * s.pitch + x0;
            pe = *(const f4 *)(s.E + q); ps = *(const f4 *)(s.S + q); pse = *(const f4 *)(s.SE + q); psw = *(const f4 *)(s.SW + q);
            vd = *(const f4 *)(s.D + q); vb = *(const f4 *)(b + q); ve = *(const f4 *)(e + q);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const bool in = row && x0 + j < s.cols;
            dg[k][j] = in ? vd[j] : 0.0f;
            rb[k][j] = in ? vb[j] : 0.0f;
            if (in) le[at_[k] + j] = ve[j];
            cw[k][j][1] = in ? pe[j] : 0.0f; cw[k][j][3] = in ? ps[j] : 0.0f;                 // the point's own couplings: E, S, ...
            cw[k][j][7] = in ? pse[j] : 0.0f; cw[k][j][6] = in ? psw[j] : 0.0f;               // ... SE, SW
        }
        if (gi < GW * EH) {
#pragma unroll
            for (int j = 0; j < 4; j++) { st0[at_[k] + j] = cw[k][j][1]; st1[at_[k] + j] = cw[k][j][3]; }
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < GPT; k++)
#pragma unroll
        for (int j = 0; j < 4; j++) { cw[k][j][0] = st0[at_[k] + j - 1]; cw[k][j][2] = st1[at_[k] + j - LP]; }        // W = left point's E, N = upper point's S
    __syncthreads();
#pragma unroll
    for (int k = 0; k < GPT; k++)
        if (tid + k * kTileThreads < GW * EH) {
#pragma unroll
            for (int j = 0; j < 4; j++) { st0[at_[k] + j] = cw[k][j][7]; st1[at_[k] + j] = cw[k][j][6]; }
        }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < GPT; k++)
#pragma unroll
        for (int j = 0; j < 4; j++) { cw[k][j][4] = st0[at_[k] + j - LP - 1]; cw[k][j][5] = st1[at_[k] + j - LP + 1]; }   // NW = upper-left point's SE, NE = upper-right point's SW
    // (le was complete before the first barrier above; no thread writes it before the loop below)
    for (int sw = 0; sw < nsweeps; sw++)
        for (int c = 0; c < 4; c++) {
            const int colour = reverse ? 3 - c : c;
#pragma unroll
            for (int k = 0; k < GPT; k++) {
                const int gi = tid + k * kTileThreads;
                const int ly = gi / GW, y = oy + ly;
                if ((y & 1) != (colour >> 1)) continue;                       // this row has no point of the colour
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    // x0 is a multiple of 4, so point j of a group has x parity j & 1
                    if ((j & 1) != (colour & 1) || !(dg[k][j] > 0.0f)) continue;
                    const int a = at_[k] + j;
                    float v = rb[k][j];
                    v += cw[k][j][0] * le[a - 1];
                    v += cw[k][j][1] * le[a + 1];
                    v += cw[k][j][2] * le[a - LP];
                    v += cw[k][j][3] * le[a + LP];
                    v += cw[k][j][4] * le[a - LP - 1];
                    v += cw[k][j][5] * le[a - LP + 1];
                    v += cw[k][j][6] * le[a + LP - 1];
                    v += cw[k][j][7] * le[a + LP + 1];
                    le[a] = v / dg[k][j];
                }
            }
            __syncthreads();
        }
#pragma unroll
    for (int k = 0; k < GPT; k++) {
        const int gi = tid + k * kTileThreads;
        const int ly = gi / GW, lx = 4 * (gi % GW), y = oy + ly, x0 = ox + lx;
        if (gi < GW * EH && ly >= H && ly < H + kTileY && lx >= H && lx < H + kTileX && y < s.rows) {
            float *o = out + (size_t)y * s.pitch + x0;
            if (x0 + 3 < s.cols) { const f4 v = {le[at_[k]], le[at_[k] + 1], le[at_[k] + 2], le[at_[k] + 3]}; *(f4 *)o = v; }
            else for (int j = 0; j < 4; j++) if (x0 + j < s.cols) o[j] = le[at_[k] + j];
        }
    }
}

// b_c = P^T r, and the coarse correction starts from zero.  TWO coarse points (I, J0), (I, J0 + 1), J0 even, per thread: their
// 3 x 5 fine points start at the 16-byte aligned column 2 J0, so every fine row is one 16-byte load plus the scalar to its left
// for r and for each weight plane it needs -- 15 mostly 16-byte loads for two points instead of 36 strided scalar ones.
// Which plane holds the weight of fine point (y, x) towards (I, J) (pweight): rows 2I and 2I+1 lie in coarse row I (planes P0/P1),
// row 2I-1 in coarse row I-1 (P2/P3); columns 2J and 2J+1 in coarse column J (even planes), column 2J-1 in column J-1 (odd planes).
// Terms are added in the order of the plain loops (py, then px), a zero weight adds nothing -- as the restatement does.
__global__ __launch_bounds__(256) void k_mg_restrict(const float *__restrict__ R, Interp ip, int frows, int fcols, int fpitch, float *bc, float *ec, int crows, int ccols, int cpitch) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int J0 = 2 * (blockIdx.x * 64 + (threadIdx.x & 63)), I = blockIdx.y * 4 + wave_id();
    if (J0 >= ccols || I >= crows) return;
    const int x0 = 2 * J0;                                         // < fcols, a multiple of 4: the 16-byte loads stay inside the row
    float acc0 = 0.0f, acc1 = 0.0f;
#pragma unroll
    for (int py = -1; py <= 1; py++) {
        const int y = 2 * I + py;
        if (y < 0 || y >= frows) continue;
        const size_t q = (size_t)y * fpitch + x0;
        const f4 r4 = *(const f4 *)(R + q);
        const f4 we = *(const f4 *)((py < 0 ? ip.P2 : ip.P0) + q);      // towards the coarse column of an even fine column
        const f4 wo = *(const f4 *)((py < 0 ? ip.P3 : ip.P1) + q);      // ... of the odd fine column to its right: only [1] (column x0+1 -> J0+1) is used
        const bool left = x0 > 0;
        const float rl = left ? R[q - 1] : 0.0f, wl = left ? (py < 0 ? ip.P3 : ip.P1)[q - 1] : 0.0f;      // column x0-1 -> J0
        // coarse point J0: fine columns x0-1, x0, x0+1
        if (wl != 0.0f) acc0 += wl * rl;
        if (we[0] != 0.0f) acc0 += we[0] * r4[0];
        if (x0 + 1 < fcols && we[1] != 0.0f) acc0 += we[1] * r4[1];
        // coarse point J0+1: fine columns x0+1, x0+2, x0+3
        if (x0 + 1 < fcols && wo[1] != 0.0f) acc1 += wo[1] * r4[1];
        if (x0 + 2 < fcols && we[2] != 0.0f) acc1 += we[2] * r4[2];
        if (x0 + 3 < fcols && we[3] != 0.0f) acc1 += we[3] * r4[3];
    }
    const size_t o = (size_t)I * cpitch + J0;
    bc[o] = acc0; ec[o] = 0.0f;
    if (J0 + 1 < ccols) { bc[o + 1] = acc1; ec[o + 1] = 0.0f; }
}

// target += P e_c; four fine points per thread (16-byte loads of the four weight planes and of the target row)
__global__ __launch_bounds__(256) void k_mg_prolong(const float *__restrict__ ec, int crows, int ccols, int cpitch, Interp ip, int frows, int fcols, int fpitch, float *T, int tpitch) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int x0 = 4 * (blockIdx.x * 64 + (threadIdx.x & 63)), y = blockIdx.y * 4 + wave_id();
    if (x0 >= fcols || y >= frows) return;
    const size_t q = (size_t)y * fpitch + x0;
    // A fine point on an EVEN row lies on a coarse row: its weights towards the coarse row below (planes P2, P3) are +0 by
    // construction (k_mg_build_p) -- those two loads are skipped and the same arithmetic runs on literal zeros.
    const f4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    const bool odd_row = (y & 1) != 0;                              // (wave-uniform: a wave holds one row)
    const f4 p0 = *(const f4 *)(ip.P0 + q), p1 = *(const f4 *)(ip.P1 + q);
    const f4 p2 = odd_row ? *(const f4 *)(ip.P2 + q) : zero4, p3 = odd_row ? *(const f4 *)(ip.P3 + q) : zero4;
    const int I = y >> 1, J = x0 >> 1;
    float c0[3], c1[3];                                            // e_c at (I, J..J+2) and (I+1, J..J+2)
#pragma unroll
    for (int k = 0; k < 3; k++) { c0[k] = at(ec, cpitch, crows, ccols, I, J + k); c1[k] = at(ec, cpitch, crows, ccols, I + 1, J + k); }
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int k = j >> 1;                                      // fine column x0 + j lies in coarse column J + k
        float t = p0[j] * c0[k];
        t += p1[j] * c0[k + 1];
        t += p2[j] * c1[k];
        t += p3[j] * c1[k + 1];
        v[j] = t;
    }
    float *o = T + (size_t)y * tpitch + x0;
    if (x0 + 3 < fcols) { f4 t4 = *(f4 *)o; t4[0] += v[0]; t4[1] += v[1]; t4[2] += v[2]; t4[3] += v[3]; *(f4 *)o = t4; }
    else for (int j = 0; j < 4; j++) if (x0 + j < fcols) o[j] += v[j];
}

// x <- clamp(x + alpha (x - x_prev)): removes an error component that shrinks by lambda = alpha / (1 + alpha) per cycle
template <bool CONTRACT>
__global__ __launch_bounds__(256) void k_mg_extrapolate(float *X, const float *Xp, int ip, int rows, int cols, float alpha) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + wave_id();
    if (x >= cols || y >= rows) return;
    const size_t p = (size_t)y * ip + x;
    const float v = X[p], d = v - Xp[p];
    const float w = CONTRACT ? __builtin_fmaf(alpha, d, v) : v + alpha * d;
    X[p] = __builtin_amdgcn_fmed3f(w, 0.0f, 255.0f);
}

inline dim3 grid_for(int rows, int cols) { return dim3((cols + 63) / 64, (rows + 3) / 4); }
inline Stencil view(const MgLevel &l) { return Stencil{l.E(), l.S(), l.SE(), l.SW(), l.D(), l.rows, l.cols, l.pitch}; }
inline Interp interp(const MgLevel &l) { return Interp{l.Pw(0), l.Pw(1), l.Pw(2), l.Pw(3)}; }

}  // namespace

void mg_release(rtdd_ctx *ctx) {
    if (ctx->mg) { ctx->mg->release(); delete ctx->mg; ctx->mg = nullptr; }
}

static int mg_allocate(rtdd_ctx *ctx, int rows, int cols) {
    if (ctx->mg && ctx->mg->rows == rows && ctx->mg->cols == cols) return RTDD_OK;
    mg_release(ctx);
    ctx->mg = new (std::nothrow) MgState();
    if (!ctx->mg) return fail(ctx, RTDD_ERR_NOMEM, "multigrid state");
    int r = rows, c = cols;
    for (int l = 0; l < 16; l++) {
        MgLevel L;
        L.rows = r; L.cols = c; L.pitch = (c + 63) / 64 * 64;
        L.plane = (size_t)L.pitch * r;
        const hipError_t e = hipMalloc((void **)&L.buf, 12 * L.plane * sizeof(float));      // 1.6 GB for level 0 at 8K
        if (e != hipSuccess) {           // leave NO half-built hierarchy behind: the next solve of this size must not find one
            mg_release(ctx);
            return fail(ctx, e == hipErrorOutOfMemory ? RTDD_ERR_NOMEM : RTDD_ERR_HIP, "hipMalloc(multigrid level)", e);
        }
        ctx->mg->lv.push_back(L);
        if ((size_t)r * c <= 256 || (r == 1 && c == 1)) break;
        r = (r + 1) / 2; c = (c + 1) / 2;
    }
    ctx->mg->rows = rows; ctx->mg->cols = cols;      // only a complete hierarchy is ever matched by the early-out above
    return RTDD_OK;
}

// the hierarchy for the current level-0 metadata (prepare has run): operators and interpolation of every level
static int mg_setup(rtdd_ctx *ctx, const Level &L0, size_t ip, int rows, int cols, int *launches) {
    int rc = mg_allocate(ctx, rows, cols);
    if (rc != RTDD_OK) return rc;
    auto &lv = ctx->mg->lv;
    hipLaunchKernelGGL(k_mg_stencil0, grid_for(rows, cols), dim3(256), 0, ctx->stream, L0.M(ip), ctx->lut_dev, (int)ip, rows, cols,
                       lv[0].E(), lv[0].S(), lv[0].SE(), lv[0].SW(), lv[0].D(), lv[0].pitch, kTheta);
    (*launches)++;
    for (size_t l = 0; l + 1 < lv.size(); l++) {
        const MgLevel &f = lv[l], &c = lv[l + 1];
        hipLaunchKernelGGL(k_mg_build_p<0>, grid_for(f.rows, f.cols), dim3(256), 0, ctx->stream, view(f), f.Pw(0), f.Pw(1), f.Pw(2), f.Pw(3));
        hipLaunchKernelGGL(k_mg_build_p<1>, grid_for(f.rows, f.cols), dim3(256), 0, ctx->stream, view(f), f.Pw(0), f.Pw(1), f.Pw(2), f.Pw(3));
        if (l == 0) hipLaunchKernelGGL(k_mg_galerkin<true>, grid_for(c.rows, c.cols), dim3(256), 0, ctx->stream, view(f), interp(f), c.E(), c.S(), c.SE(), c.SW(), c.D(), c.rows, c.cols, c.pitch);
        else hipLaunchKernelGGL(k_mg_galerkin<false>, grid_for(c.rows, c.cols), dim3(256), 0, ctx->stream, view(f), interp(f), c.E(), c.S(), c.SE(), c.SW(), c.D(), c.rows, c.cols, c.pitch);
        hipLaunchKernelGGL(k_mg_prune, grid_for(c.rows, c.cols), dim3(256), 0, ctx->stream, c.E(), c.S(), c.SE(), c.SW(), c.D(), c.rows, c.cols, c.pitch);
        *launches += 4;
    }
    RTDD_LAUNCH_CHECK(ctx, "multigrid setup");
    return RTDD_OK;
}

static void mg_smooth(rtdd_ctx *ctx, MgLevel &l, int nsweeps, bool reverse, int *launches) {
    const size_t npts = (size_t)l.rows * l.cols, lds = (size_t)(l.rows + 2) * (l.cols + 2) * sizeof(float);
    if (npts <= 8192 && lds <= 62 * 1024) {
        const Stencil v = view(l);
        const int rev = reverse ? 1 : 0;
        if (npts <= 1024) hipLaunchKernelGGL(k_mg_gs_lds<1>, dim3(1), dim3(1024), lds, ctx->stream, v, l.e(), l.b(), nsweeps, rev);
        else if (npts <= 2048) hipLaunchKernelGGL(k_mg_gs_lds<2>, dim3(1), dim3(1024), lds, ctx->stream, v, l.e(), l.b(), nsweeps, rev);
        else if (npts <= 4096) hipLaunchKernelGGL(k_mg_gs_lds<4>, dim3(1), dim3(1024), lds, ctx->stream, v, l.e(), l.b(), nsweeps, rev);
        else hipLaunchKernelGGL(k_mg_gs_lds<8>, dim3(1), dim3(1024), lds, ctx->stream, v, l.e(), l.b(), nsweeps, rev);
        (*launches)++;
        return;
    }
    if (nsweeps <= 2) {                                   // (64 + 16) x (32 + 16) extended points = 960 groups of 4: two per thread
        const int H = 4 * nsweeps;
        const size_t lds = 3 * (size_t)(kTileX + 2 * H + 2) * (kTileY + 2 * H + 2) * sizeof(float);       // the iterate + two staging planes
        const dim3 g((l.cols + kTileX - 1) / kTileX, (l.rows + kTileY - 1) / kTileY);
        hipLaunchKernelGGL(k_mg_gs_tile, g, dim3(kTileThreads), lds, ctx->stream, view(l), l.e(), l.r(), l.b(), nsweeps, reverse ? 1 : 0);
        const int t = l.ei; l.ei = l.ri; l.ri = t;        // the result is in what was r's plane
        (*launches)++;
        return;
    }
    // not reached by the cycle as configured (a coarsest level is at most 256 points): many sweeps on a level too large for LDS
    if (npts <= (size_t)kSmallLevel) {
        hipLaunchKernelGGL(k_mg_gs_small, dim3(1), dim3(1024), 0, ctx->stream, view(l), l.e(), l.b(), nsweeps, reverse ? 1 : 0);
        (*launches)++;
        return;
    }
    const dim3 g(((l.cols + 1) / 2 + 63) / 64, ((l.rows + 1) / 2 + 3) / 4);
    for (int s = 0; s < nsweeps; s++)
        for (int c = 0; c < 4; c++) {
            hipLaunchKernelGGL(k_mg_gs, g, dim3(256), 0, ctx->stream, view(l), l.e(), l.b(), reverse ? 3 - c : c);
            (*launches)++;
        }
}

// one V(kNu,kNu) cycle on the level-0 iterate in plane *plane
// The plane holding the iterate on entry is left untouched (the extrapolation needs x_{k-1}): the result lands in a third one.
static int mg_vcycle(rtdd_ctx *ctx, const Level &L0, size_t ip, int rows, int cols, int *plane, int *launches) {
    const int entry = *plane;
    auto &lv = ctx->mg->lv;
    const int last = (int)lv.size() - 1;
    int ln = 0, rc;
    if (last == 0) return launch_rbgs_blocked(ctx, L0, ip, rows, cols, 2 * kNu, 1.0f, plane, launches, -1);
    if ((rc = launch_rbgs_blocked(ctx, L0, ip, rows, cols, kNu, 1.0f, plane, &ln)) != RTDD_OK) return rc;
    *launches += ln;
    hipLaunchKernelGGL(k_mg_residual0, dim3((cols + 255) / 256, (rows + 3) / 4), dim3(256), 0, ctx->stream, L0.P(*plane, ip), L0.M(ip), ctx->lut_dev, (int)ip, rows, cols, lv[0].r(), lv[0].pitch);
    (*launches)++;
    for (int l = 0; l < last; l++) {                    // down
        MgLevel &f = lv[l], &c = lv[l + 1];
        hipLaunchKernelGGL(k_mg_restrict, grid_for(c.rows, (c.cols + 1) / 2), dim3(256), 0, ctx->stream, f.r(), interp(f), f.rows, f.cols, f.pitch, c.b(), c.e(), c.rows, c.cols, c.pitch);
        (*launches)++;
        if (l + 1 == last) { mg_smooth(ctx, c, kCoarsestSweeps, false, launches); break; }
        mg_smooth(ctx, c, kNu, false, launches);
        hipLaunchKernelGGL(k_mg_residual, grid_for(c.rows, c.cols), dim3(256), 0, ctx->stream, view(c), c.e(), c.b(), c.r());
        (*launches)++;
    }
    for (int l = last - 1; l >= 1; l--) {                // up
        MgLevel &f = lv[l], &c = lv[l + 1];
        hipLaunchKernelGGL(k_mg_prolong, grid_for(f.rows, (f.cols + 3) / 4), dim3(256), 0, ctx->stream, c.e(), c.rows, c.cols, c.pitch, interp(f), f.rows, f.cols, f.pitch, f.e(), f.pitch);
        (*launches)++;
        mg_smooth(ctx, f, kNu, true, launches);
    }
    hipLaunchKernelGGL(k_mg_prolong, grid_for(rows, (cols + 3) / 4), dim3(256), 0, ctx->stream, lv[1].e(), lv[1].rows, lv[1].cols, lv[1].pitch, interp(lv[0]), rows, cols, lv[0].pitch,
                       L0.P(*plane, ip), (int)ip);
    (*launches)++;
    RTDD_LAUNCH_CHECK(ctx, "multigrid cycle");
    ln = 0;
    if ((rc = launch_rbgs_blocked(ctx, L0, ip, rows, cols, kNu, 1.0f, plane, &ln, entry)) != RTDD_OK) return rc;
    *launches += ln;
    return RTDD_OK;
}

// alternative_seconds > 0 (RTDD_METHOD_AUTO): leave as soon as the cycles still needed at the rate of the last two,
// priced at cycle_seconds each, cost more than the alternative (SOR cycles from here).  Both prices come from the caller's
// constants (rtdd_set_option RTDD_OPT_AUTO_*): a deterministic rule on the f32 residuals -- no clocks -- so a solve is reproducible.
int launch_multigrid(rtdd_ctx *ctx, const Level &L0, size_t ip, int rows, int cols, int max_cycles, float tolerance, int check_every, double alternative_seconds,
                     double cycle_seconds, int *plane, int *cycles_done, float *residual, int *launches) {
    int rc = mg_setup(ctx, L0, ip, rows, cols, launches);
    if (rc != RTDD_OK) return rc;
    *cycles_done = 0;
    float before = INFINITY, before2 = INFINITY;         // residuals one and two checks ago
    int since = 0;                                       // cycles since the start or the last extrapolation
    while (*cycles_done < max_cycles) {
        const int prev_plane = *plane;                   // x_{k-1} stays here during the cycle
        if ((rc = mg_vcycle(ctx, L0, ip, rows, cols, plane, launches)) != RTDD_OK) return rc;
        (*cycles_done)++;
        if (tolerance > 0.0f && (*cycles_done % check_every == 0 || *cycles_done == max_cycles)) {
            if ((rc = launch_residual(ctx, L0, ip, *plane, rows, cols, residual)) != RTDD_OK) return rc;
            if (*residual <= tolerance) break;
            if (alternative_seconds > 0.0 && before2 < INFINITY) {
                const double rate = sqrt((double)*residual / (double)before2);         // per cycle, over the last two
                if (!(rate < 1.0)) break;
                const double needed = ceil(log((double)*residual / (double)tolerance) / -log(rate));
                if (needed * cycle_seconds > alternative_seconds) break;
            }
            // Vector extrapolation.  Once the residual shrinks by the same factor lambda two cycles in a row, what is left is
            // one slowly decaying family, e_k = lambda e_{k-1}, and x_k + lambda/(1-lambda) (x_k - x_{k-1}) removes it
            // (8K: 19 -> 11 cycles; scripts/mg_extrapolation_probe.py).  Decided on the f32 residuals alone: reproducible.
            since++;
            if (check_every == 1 && since >= 3 && before2 < INFINITY && ctx->mg->lv.size() > 1 && *plane != prev_plane && *cycles_done < max_cycles) {
                const double l1 = (double)*residual / (double)before, l0 = (double)before / (double)before2;
                if (l1 > 0.3 && l1 < 0.995 && fabs(l1 - l0) <= 0.05 * l1) {
                    const float alpha = (float)(l1 / (1.0 - l1));
                    if (ctx->opt.fp_contract) hipLaunchKernelGGL(k_mg_extrapolate<true>, grid_for(rows, cols), dim3(256), 0, ctx->stream, L0.P(*plane, ip), L0.P(prev_plane, ip), (int)ip, rows, cols, alpha);
                    else hipLaunchKernelGGL(k_mg_extrapolate<false>, grid_for(rows, cols), dim3(256), 0, ctx->stream, L0.P(*plane, ip), L0.P(prev_plane, ip), (int)ip, rows, cols, alpha);
                    (*launches)++;
                    since = 0;
                }
            }
            before2 = before; before = *residual;
        }
    }
    return RTDD_OK;
}

// diagnostics for the parity tests: copy one plane of one level to the host (dense rows x cols)
int mg_download(rtdd_ctx *ctx, int level, int which, float *host, int *rows, int *cols) {
    if (!ctx->mg || level < 0 || level >= (int)ctx->mg->lv.size() || which < 0 || which >= 12) return RTDD_ERR_INVALID;
    const MgLevel &l = ctx->mg->lv[level];
    *rows = l.rows; *cols = l.cols;
    if (!host) return RTDD_OK;
    RTDD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    RTDD_HIP(ctx, hipMemcpy2D(host, (size_t)l.cols * sizeof(float), l.A(which), (size_t)l.pitch * sizeof(float), (size_t)l.cols * sizeof(float), l.rows, hipMemcpyDeviceToHost));
    return RTDD_OK;
}

}  // namespace rtdd
