// solver_kernels.hip -- gfx950 kernels of the diffusion solve.
//
// What the reference does per level (/root/reference/src/GPUSolver.cu:290-312): memset prev,
// two pitched->dense copies, the edge-weight index pass, maxIterations one-sweep launches on
// 16x16 blocks moving 25 B per pixel-sweep, one dense->pitched copy.  What this file does:
//
//   k_prepare   one pass: depth -> x_0 plane, x_{-1} plane (0 on free pixels), and ONE packed
//               u32 of metadata per pixel (right/down weight index + Dirichlet bit).
//   k_sweep1    one Chebyshev-Jacobi sweep per launch.  A wave owns a 256-pixel-wide column
//               strip and walks R rows with a 3-row register window: every x_k row is fetched
//               once per wave as one 1-KiB dwordx4 instruction, x_{k-1} is read and x_{k+1}
//               written IN PLACE in the same plane (each pixel touches only its own slot), so a
//               sweep moves 16 B per pixel (4 x_k + 4 x_{k-1} + 4 x_{k+1} + 4 meta).  The 257-
//               entry weight LUT sits in LDS (divergent gathers are cheap there; the reference's
//               __constant__ reads would serialise on AMD's scalar cache).
//   k_finish    result plane -> caller's pitched buffer.
//
// Arithmetic is op-for-op that of oracle/rtdd_oracle.c (same order, same roundings, explicit
// fmaf only in the contracted variant; the file is compiled with -ffp-contract=off), which is
// what makes GPU-vs-oracle parity bit-exact rather than merely within 1e-4.
#include "rtdd_internal.hpp"
#include "persist_sync.hpp"

namespace rtdd {

__device__ __forceinline__ int sat_u8_dev(float v) {
    // defined behaviour for the reference's out-of-range float->uchar casts: saturate, then truncate
    if (!(v >= 0.0f)) return 0;
    if (v >= 255.0f) return 255;
    return (int)v;
}

__device__ __forceinline__ int iabs(int a) { return a < 0 ? -a : a; }

// ------------------------------------------------------------------------------------------------
// k_prepare: edge-weight indices (src/GPUSolver.cu:183-222) + staging (:290-292), fused.
// gated = (level != maxLevel); thr = (level == 0) ? 0 : 4   (:201-202)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prepare(const float *__restrict__ depth, size_t depthPitch,
                                                 const uint8_t *__restrict__ scribble, size_t scribblePitch,
                                                 const uint8_t *__restrict__ gray, size_t grayPitch,
                                                 float *__restrict__ X0, float *__restrict__ X1,
                                                 uint32_t *__restrict__ M, int ip, int rows, int cols, int gated, int thr,
                                                 size_t zDepth, size_t zScribble, size_t zGray, size_t zPlane) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + wave_id();
    if (x >= cols || y >= rows) return;
    RTDD_Z(depth, zDepth); RTDD_Z(scribble, zScribble); RTDD_Z(gray, zGray); RTDD_Z(X0, zPlane); RTDD_Z(X1, zPlane); RTDD_Z(M, zPlane);
    const float *drow = (const float *)((const char *)depth + (size_t)y * depthPitch);
    const uint8_t *grow = gray + (size_t)y * grayPitch;
    const float d = drow[x];
    const int g = grow[x];
    const bool dirichlet = scribble[(size_t)y * scribblePitch + x] == 255;
    int right = 0, down = 0;
    if (x + 1 < cols) {
        right = iabs(g - (int)grow[x + 1]);
        if (gated && !(iabs(sat_u8_dev(d) - sat_u8_dev(drow[x + 1])) > thr)) right = 0;
    }
    if (y + 1 < rows) {
        down = iabs(g - (int)grow[x + grayPitch]);
        if (gated) {
            const float dd = ((const float *)((const char *)depth + (size_t)(y + 1) * depthPitch))[x];
            if (!(iabs(sat_u8_dev(d) - sat_u8_dev(dd)) > thr)) down = 0;
        }
    }
    const size_t p = (size_t)y * ip + x;
    X0[p] = d;
    X1[p] = dirichlet ? d : 0.0f;     // x_{-1} = 0 on free pixels (cudaMemset, :290); Dirichlet value in both planes (:291-292)
    M[p] = (uint32_t)right | ((uint32_t)down << 8) | (dirichlet ? kMetaDirichlet : 0u);
}

// The same pass, four pixels per thread, for callers whose rows are 16-byte (depth) / 4-byte (gray, scribble) aligned -- what
// cudaMallocPitch-style allocations give: one 16-byte load per depth row, one 4-byte load per u8 row, 16-byte stores of the three
// planes.  A group that would run past the end of a row falls back to the scalar form above, pixel by pixel.
__global__ __launch_bounds__(256) void k_prepare4(const float *__restrict__ depth, size_t depthPitch,
                                                  const uint8_t *__restrict__ scribble, size_t scribblePitch,
                                                  const uint8_t *__restrict__ gray, size_t grayPitch,
                                                  float *__restrict__ X0, float *__restrict__ X1,
                                                  uint32_t *__restrict__ M, int ip, int rows, int cols, int gated, int thr,
                                                  size_t zDepth, size_t zScribble, size_t zGray, size_t zPlane) {
    const int x0 = 4 * (blockIdx.x * 64 + (threadIdx.x & 63));
    const int y = blockIdx.y * 4 + wave_id();
    if (x0 >= cols || y >= rows) return;
    RTDD_Z(depth, zDepth); RTDD_Z(scribble, zScribble); RTDD_Z(gray, zGray); RTDD_Z(X0, zPlane); RTDD_Z(X1, zPlane); RTDD_Z(M, zPlane);
    const float *drow = (const float *)((const char *)depth + (size_t)y * depthPitch);
    const uint8_t *grow = gray + (size_t)y * grayPitch, *srow = scribble + (size_t)y * scribblePitch;
    const bool down_ok = y + 1 < rows;
    float d[5], dd[4];
    int g[5], gd[4];
    uint32_t sc;
    if (x0 + 3 < cols) {
        const float4 d4 = *(const float4 *)(drow + x0);
        d[0] = d4.x; d[1] = d4.y; d[2] = d4.z; d[3] = d4.w;
        const uint32_t g4 = *(const uint32_t *)(grow + x0);
        g[0] = g4 & 255; g[1] = (g4 >> 8) & 255; g[2] = (g4 >> 16) & 255; g[3] = g4 >> 24;
        sc = *(const uint32_t *)(srow + x0);
        if (down_ok) {
            const uint32_t h4 = *(const uint32_t *)(grow + grayPitch + x0);
            gd[0] = h4 & 255; gd[1] = (h4 >> 8) & 255; gd[2] = (h4 >> 16) & 255; gd[3] = h4 >> 24;
            if (gated) { const float4 e4 = *(const float4 *)((const float *)((const char *)depth + (size_t)(y + 1) * depthPitch) + x0); dd[0] = e4.x; dd[1] = e4.y; dd[2] = e4.z; dd[3] = e4.w; }
        }
    } else {                                       // ragged end of the row
        sc = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const bool in = x0 + i < cols;
            d[i] = in ? drow[x0 + i] : 0.0f; g[i] = in ? grow[x0 + i] : 0;
            sc |= (uint32_t)(in ? srow[x0 + i] : 0) << (8 * i);
            gd[i] = in && down_ok ? grow[x0 + i + grayPitch] : 0;
            dd[i] = in && down_ok && gated ? ((const float *)((const char *)depth + (size_t)(y + 1) * depthPitch))[x0 + i] : 0.0f;
        }
    }
    const bool right_ok = x0 + 4 < cols;
    d[4] = right_ok ? drow[x0 + 4] : 0.0f; g[4] = right_ok ? grow[x0 + 4] : 0;
    float x0v[4], x1v[4];
    uint32_t mv[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const bool dirichlet = ((sc >> (8 * i)) & 255) == 255;
        int right = 0, down = 0;
        if (x0 + i + 1 < cols) {
            right = iabs(g[i] - g[i + 1]);
            if (gated && !(iabs(sat_u8_dev(d[i]) - sat_u8_dev(d[i + 1])) > thr)) right = 0;
        }
        if (down_ok) {
            down = iabs(g[i] - gd[i]);
            if (gated && !(iabs(sat_u8_dev(d[i]) - sat_u8_dev(dd[i])) > thr)) down = 0;
        }
        x0v[i] = d[i];
        x1v[i] = dirichlet ? d[i] : 0.0f;
        mv[i] = (uint32_t)right | ((uint32_t)down << 8) | (dirichlet ? kMetaDirichlet : 0u);
    }
    const size_t p = (size_t)y * ip + x0;          // the planes' rows are 256-byte aligned and padded: whole 16-byte stores always fit; columns >= cols are never read as pixels
    *(float4 *)(X0 + p) = make_float4(x0v[0], x0v[1], x0v[2], x0v[3]);
    *(float4 *)(X1 + p) = make_float4(x1v[0], x1v[1], x1v[2], x1v[3]);
    *(uint4 *)(M + p) = make_uint4(mv[0], mv[1], mv[2], mv[3]);
}

// The reference's own index format, for parity tests of the weight pass (src/GPUSolver.cu:136-224).
__global__ __launch_bounds__(256) void k_index_to_weight(const uint8_t *__restrict__ gray, size_t grayPitch,
                                                         const float *__restrict__ depth, size_t depthPitch,
                                                         int32_t *__restrict__ index2, int rows, int cols, int gated, int thr) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + wave_id();
    if (x >= cols || y >= rows) return;
    const int g = gray[(size_t)y * grayPitch + x];
    const float *drow = (const float *)((const char *)depth + (size_t)y * depthPitch);
    const int d = gated ? sat_u8_dev(drow[x]) : 0;
    int idx[4] = {256, 256, 256, 256};   // left right up down
    const int dx[4] = {-1, 1, 0, 0}, dy[4] = {0, 0, -1, 1};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int nx = x + dx[k], ny = y + dy[k];
        if (nx < 0 || ny < 0 || nx >= cols || ny >= rows) continue;
        int v = iabs(g - (int)gray[(size_t)ny * grayPitch + nx]);
        if (gated) {
            const float nd = ((const float *)((const char *)depth + (size_t)ny * depthPitch))[nx];
            if (!(iabs(d - sat_u8_dev(nd)) > thr)) v = 0;
        }
        idx[k] = v;
    }
    const size_t p = (size_t)y * cols + x;
    index2[2 * p + 0] = idx[0] * 1000 + idx[1];
    index2[2 * p + 1] = idx[2] * 1000 + idx[3];
}

// ------------------------------------------------------------------------------------------------
// One pixel of one sweep: solveDiffusion (src/GPUSolver.cu:73-106) + the Chebyshev update (:257-260)
// ------------------------------------------------------------------------------------------------
template <bool CONTRACT>
__device__ __forceinline__ float mean4(float xl, float xr, float xu, float xd, float wl, float wr, float wu, float wd,
                                       bool vl, bool vr, bool vu, bool vd) {
    float sum = 0.0f, cnt = 0.0f, s2;
    s2 = CONTRACT ? __builtin_fmaf(wl, xl, sum) : sum + wl * xl;  sum = vl ? s2 : sum;  cnt = vl ? cnt + wl : cnt;
    s2 = CONTRACT ? __builtin_fmaf(wr, xr, sum) : sum + wr * xr;  sum = vr ? s2 : sum;  cnt = vr ? cnt + wr : cnt;
    s2 = CONTRACT ? __builtin_fmaf(wu, xu, sum) : sum + wu * xu;  sum = vu ? s2 : sum;  cnt = vu ? cnt + wu : cnt;
    s2 = CONTRACT ? __builtin_fmaf(wd, xd, sum) : sum + wd * xd;  sum = vd ? s2 : sum;  cnt = vd ? cnt + wd : cnt;
    float r = sum / cnt;                      // IEEE correctly-rounded divide (-fhip-fp32-correctly-rounded-divide-sqrt)
    if (!(r >= 0.0f)) r = 0.0f;               // min(max(.,0),255) with fmax/fmin NaN semantics (:104)
    if (r > 255.0f) r = 255.0f;
    if (cnt == 0.0f) r = 0.0f;                // :103
    return r;
}

template <bool CONTRACT>
__device__ __forceinline__ float chebyshev(float r, float x, float prev, float omega, float gamma) {
    if (CONTRACT) return __builtin_fmaf(omega, __builtin_fmaf(gamma, r - x, x) - prev, prev);
    return (omega * (gamma * (r - x) + x - prev)) + prev;
}

// ------------------------------------------------------------------------------------------------
// k_sweep1: one sweep per launch.  block = 256 threads = 4 waves stacked vertically; wave w of
// block (bx,by) owns columns [256*bx, 256*bx+256) and rows [(4*by+w)*R, +R).  Lane l holds the
// 4 pixels 256*bx+4l .. +3 of each row.  X = x_k (read-only), Y = x_{k-1} in / x_{k+1} out.
// ------------------------------------------------------------------------------------------------
template <bool CONTRACT>
__global__ __launch_bounds__(256) void k_sweep1(const float *__restrict__ X, float *__restrict__ Y,
                                                const uint32_t *__restrict__ M, const float *__restrict__ lut_g,
                                                int ip, int rows, int cols, int R, float omega, float gamma) {
    __shared__ float lut[257];
    for (int i = threadIdx.x; i < 257; i += 256) lut[i] = lut_g[i];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = wave_id();
    const int x0 = blockIdx.x * 256 + lane * 4;
    const int ybeg = (blockIdx.y * 4 + wave) * R;
    const int yend = min(ybeg + R, rows);
    if (ybeg >= rows) return;

    const bool lane_in = x0 < cols;
    // per-lane validity of the horizontal neighbours of its 4 pixels
    const bool vl0 = x0 > 0;
    bool vr[4];
#pragma unroll
    for (int i = 0; i < 4; i++) vr[i] = x0 + i + 1 < cols;

    const float *xrow = X + (size_t)ybeg * ip + x0;
    float4 up = *(const float4 *)(xrow - ip);
    float4 cur = *(const float4 *)(xrow);
    float wu[4];
    {
        const uint4 mu = *(const uint4 *)(M + (size_t)(ybeg - 1) * ip + x0);   // guard row when ybeg == 0 (value unused)
        wu[0] = lut[(mu.x >> 8) & 255]; wu[1] = lut[(mu.y >> 8) & 255];
        wu[2] = lut[(mu.z >> 8) & 255]; wu[3] = lut[(mu.w >> 8) & 255];
    }

    for (int y = ybeg; y < yend; y++) {
        const size_t rowoff = (size_t)y * ip + x0;
        const float4 dn = *(const float4 *)(X + rowoff + ip);
        const uint4 m = *(const uint4 *)(M + rowoff);
        const float4 pv = *(const float4 *)(Y + rowoff);

        // horizontal halo: neighbours' edge values by wave shuffle, strip edges by scalar load
        float xl = __shfl_up(cur.w, 1);
        float xr = __shfl_down(cur.x, 1);
        uint32_t ml = __shfl_up(m.w, 1);
        if (lane == 0) { xl = X[rowoff - 1]; ml = M[rowoff - 1]; }
        if (lane == 63) xr = X[rowoff + 4];

        const float wr[4] = {lut[m.x & 255], lut[m.y & 255], lut[m.z & 255], lut[m.w & 255]};
        const float wd[4] = {lut[(m.x >> 8) & 255], lut[(m.y >> 8) & 255], lut[(m.z >> 8) & 255], lut[(m.w >> 8) & 255]};
        const float wl0 = lut[ml & 255];
        const bool vu = y > 0, vd = y + 1 < rows;

        const float xc[4] = {cur.x, cur.y, cur.z, cur.w};
        const float xu[4] = {up.x, up.y, up.z, up.w};
        const float xd[4] = {dn.x, dn.y, dn.z, dn.w};
        const float pp[4] = {pv.x, pv.y, pv.z, pv.w};
        const uint32_t mm[4] = {m.x, m.y, m.z, m.w};
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float r = mean4<CONTRACT>(i == 0 ? xl : xc[i - 1], i == 3 ? xr : xc[i + 1], xu[i], xd[i],
                                            i == 0 ? wl0 : wr[i - 1], wr[i], wu[i], wd[i],
                                            i == 0 ? vl0 : true, vr[i], vu, vd);
            const float v = chebyshev<CONTRACT>(r, xc[i], pp[i], omega, gamma);
            o[i] = (mm[i] & kMetaDirichlet) ? xc[i] : v;      // Dirichlet pixels keep their value (:248)
        }
        if (lane_in) *(float4 *)(Y + rowoff) = make_float4(o[0], o[1], o[2], o[3]);
        up = cur; cur = dn;
#pragma unroll
        for (int i = 0; i < 4; i++) wu[i] = wd[i];
    }
}

// ------------------------------------------------------------------------------------------------
// k_finish: dense result plane -> caller's pitched buffer (copyToPitchedData, src/GPUSolver.cu:122-134)
// ------------------------------------------------------------------------------------------------
// `u8` (optional): the same values as GpuMat::convertTo(CV_8UC1) makes them -- saturate(round-half-even), src/main.cpp:290 -- so that the
// finest level of an estimate needs no k_depth_to_u8 launch of its own.
__device__ __forceinline__ uint8_t round_u8(float v) {
    const float r = __builtin_rintf(v);
    return !(r >= 0.0f) ? 0 : (r >= 255.0f ? 255 : (uint8_t)(int)r);
}

__global__ __launch_bounds__(256) void k_finish(const float *__restrict__ X, int ip, float *__restrict__ depth, size_t depthPitch,
                                                int rows, int cols, uint8_t *__restrict__ u8, size_t u8Pitch, int *sync_words, int seq,
                                                size_t zPlane, size_t zDepth, size_t zU8, uint8_t *__restrict__ u8b, size_t u8bPitch) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + wave_id();
    if (solve_is_dead(sync_words, seq, (blockIdx.x | blockIdx.y | blockIdx.z | threadIdx.x) == 0)) return;
    if (x >= cols || y >= rows) return;
    RTDD_Z(X, zPlane); RTDD_Z(depth, zDepth); if (u8) RTDD_Z(u8, zU8);
    const float v = X[(size_t)y * ip + x];
    ((float *)((char *)depth + (size_t)y * depthPitch))[x] = v;
    if (u8) u8[(size_t)y * u8Pitch + x] = round_u8(v);
    if (u8b) u8b[(size_t)y * u8bPitch + x] = round_u8(v);          // (a live frame's staging slot: the same map once more, no copy kernel)
}

// four pixels per thread when the caller's rows are 16-byte aligned (a group past the end of the row: pixel by pixel)
__global__ __launch_bounds__(256) void k_finish4(const float *__restrict__ X, int ip, float *__restrict__ depth, size_t depthPitch,
                                                 int rows, int cols, uint8_t *__restrict__ u8, size_t u8Pitch, int *sync_words, int seq,
                                                 size_t zPlane, size_t zDepth, size_t zU8, uint8_t *__restrict__ u8b, size_t u8bPitch) {
    const int x0 = 4 * (blockIdx.x * 64 + (threadIdx.x & 63));
    const int y = blockIdx.y * 4 + wave_id();
    if (solve_is_dead(sync_words, seq, (blockIdx.x | blockIdx.y | blockIdx.z | threadIdx.x) == 0)) return;
    if (x0 >= cols || y >= rows) return;
    RTDD_Z(X, zPlane); RTDD_Z(depth, zDepth); if (u8) RTDD_Z(u8, zU8);
    const float4 v = *(const float4 *)(X + (size_t)y * ip + x0);
    float *o = (float *)((char *)depth + (size_t)y * depthPitch) + x0;
    const float t[4] = {v.x, v.y, v.z, v.w};
    if (x0 + 3 < cols) *(float4 *)o = v;
    else { for (int i = 0; i < 4; i++) if (x0 + i < cols) o[i] = t[i]; }
    uint8_t *targets[2] = {u8, u8b};              // (u8b: a live frame's staging slot -- the same map once more, no copy kernel)
    const size_t pitches[2] = {u8Pitch, u8bPitch};
#pragma unroll
    for (int k = 0; k < 2; k++) {
        if (!targets[k]) continue;
        uint8_t *q = targets[k] + (size_t)y * pitches[k] + x0;
        if (x0 + 3 < cols && pitches[k] % 4 == 0 && (uintptr_t)targets[k] % 4 == 0)
            *(uint32_t *)q = (uint32_t)round_u8(t[0]) | ((uint32_t)round_u8(t[1]) << 8) | ((uint32_t)round_u8(t[2]) << 16) | ((uint32_t)round_u8(t[3]) << 24);
        else { for (int i = 0; i < 4; i++) if (x0 + i < cols) q[i] = round_u8(t[i]); }
    }
}

// ------------------------------------------------------------------------------------------------
// Extensions (no reference behaviour): residual max|J(x)-x| and red-black Gauss-Seidel.
// ------------------------------------------------------------------------------------------------
template <bool CONTRACT>
__device__ __forceinline__ float mean_at(const float *__restrict__ X, const uint32_t *__restrict__ M, const float *lut,
                                         int ip, int rows, int cols, int x, int y, uint32_t m) {
    const size_t p = (size_t)y * ip + x;
    const bool vl = x > 0, vr = x + 1 < cols, vu = y > 0, vd = y + 1 < rows;
    const uint32_t ml = M[p - 1], mu = M[p - ip];            // guard cells when invalid (values unused)
    return mean4<CONTRACT>(X[p - 1], X[p + 1], X[p - ip], X[p + ip],
                           lut[ml & 255], lut[m & 255], lut[(mu >> 8) & 255], lut[(m >> 8) & 255], vl, vr, vu, vd);
}

// 4 pixels per thread (16-byte loads; the scalar one-pixel-per-thread version ran at 1.8 TB/s), block = 64 lanes x 4 rows
template <bool CONTRACT>
__global__ __launch_bounds__(256) void k_residual(const float *__restrict__ X, const uint32_t *__restrict__ M,
                                                  const float *__restrict__ lut_g, int ip, int rows, int cols,
                                                  unsigned int *__restrict__ out_bits) {
    __shared__ float lut[257];
    __shared__ float wmax[4];
    for (int i = threadIdx.x; i < 257; i += 256) lut[i] = lut_g[i];
    __syncthreads();
    const int x0 = 4 * (blockIdx.x * 64 + (threadIdx.x & 63));
    const int y = blockIdx.y * 4 + wave_id();
    float d = 0.0f;
    if (x0 < cols && y < rows) {
        const size_t p = (size_t)y * ip + x0;                      // guard rows/columns make every address below valid
        const float4 c4 = *(const float4 *)(X + p), u4 = *(const float4 *)(X + p - ip), d4 = *(const float4 *)(X + p + ip);
        const uint4 m4 = *(const uint4 *)(M + p), mu4 = *(const uint4 *)(M + p - ip);
        const float xc[4] = {c4.x, c4.y, c4.z, c4.w}, xu[4] = {u4.x, u4.y, u4.z, u4.w}, xd[4] = {d4.x, d4.y, d4.z, d4.w};
        const uint32_t mm[4] = {m4.x, m4.y, m4.z, m4.w}, mu[4] = {mu4.x, mu4.y, mu4.z, mu4.w};
        const float xl0 = X[p - 1], xr4 = X[p + 4];
        const uint32_t ml0 = M[p - 1];
        const bool vu = y > 0, vd = y + 1 < rows;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int x = x0 + i;
            if (x < cols && !(mm[i] & kMetaDirichlet)) {
                const float r = mean4<CONTRACT>(i == 0 ? xl0 : xc[i - 1], i == 3 ? xr4 : xc[i + 1], xu[i], xd[i],
                                                lut[(i == 0 ? ml0 : mm[i - 1]) & 255], lut[mm[i] & 255], lut[(mu[i] >> 8) & 255], lut[(mm[i] >> 8) & 255],
                                                x > 0, x + 1 < cols, vu, vd);
                float e = fabsf(r - xc[i]);
                if (!(e >= 0.0f)) e = __builtin_inff();            // NaN must not hide
                d = fmaxf(d, e);
            }
        }
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) d = fmaxf(d, __shfl_xor(d, s));   // wave64 butterfly
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        d = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        // non-negative floats order like their bit patterns.  The target only grows, so a block whose maximum is not above
        // what is already there has nothing to add: without this test 130 000 blocks of an 8K image serialise on one address
        const unsigned int bits = __float_as_uint(d);
        if (bits > __hip_atomic_load(out_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out_bits, bits);
    }
}

template <bool CONTRACT>
__global__ __launch_bounds__(256) void k_rbgs_half(float *__restrict__ X, const uint32_t *__restrict__ M,
                                                   const float *__restrict__ lut_g, int ip, int rows, int cols, int colour, float omega) {
    __shared__ float lut[257];
    for (int i = threadIdx.x; i < 257; i += 256) lut[i] = lut_g[i];
    __syncthreads();
    const int y = blockIdx.y * 4 + wave_id();
    const int x = 2 * (blockIdx.x * 64 + (threadIdx.x & 63)) + ((y + colour) & 1);
    if (x >= cols || y >= rows) return;
    const uint32_t m = M[(size_t)y * ip + x];
    if (m & kMetaDirichlet) return;
    float v = mean_at<CONTRACT>(X, M, lut, ip, rows, cols, x, y, m);
    if (omega != 1.0f) {                                       // SOR extension: x <- clamp(x + omega (gs - x))
        const float x0 = X[(size_t)y * ip + x];
        v = CONTRACT ? __builtin_fmaf(omega, v - x0, x0) : x0 + omega * (v - x0);
        v = fminf(fmaxf(v, 0.0f), 255.0f);
    }
    X[(size_t)y * ip + x] = v;
}

// ================================================================================================
// host-side launchers
// ================================================================================================
static inline dim3 grid64x4(int rows, int cols, int images = 1) { return dim3((cols + 63) / 64, (rows + 3) / 4, images); }

int launch_prepare(rtdd_ctx *ctx, const Level &L, size_t ip, const float *depth, size_t depthPitch,
                   const uint8_t *scribble, size_t scribblePitch, const uint8_t *gray, size_t grayPitch,
                   int rows, int cols, int level, const Batch &B) {
    const int gated = level != ctx->maxLevel;
    const int thr = level == 0 ? 0 : 4;
    // (B.n = 1: one image, strides unused; L is already image B.first's view)
    const size_t zP = L.elems * sizeof(float);
    const bool aligned = ((uintptr_t)depth % 16 == 0) && depthPitch % 16 == 0 && ((uintptr_t)gray % 4 == 0) && grayPitch % 4 == 0 &&
                         ((uintptr_t)scribble % 4 == 0) && scribblePitch % 4 == 0 && B.depth % 16 == 0 && B.gray % 4 == 0 && B.scribble % 4 == 0;
    if (aligned)
        hipLaunchKernelGGL(k_prepare4, grid64x4(rows, (cols + 3) / 4, B.n), dim3(256), 0, ctx->stream, depth, depthPitch, scribble, scribblePitch,
                           gray, grayPitch, L.P(0, ip), L.P(1, ip), L.M(ip), (int)ip, rows, cols, gated, thr, B.depth, B.scribble, B.gray, zP);
    else
        hipLaunchKernelGGL(k_prepare, grid64x4(rows, cols, B.n), dim3(256), 0, ctx->stream, depth, depthPitch, scribble, scribblePitch,
                           gray, grayPitch, L.P(0, ip), L.P(1, ip), L.M(ip), (int)ip, rows, cols, gated, thr, B.depth, B.scribble, B.gray, zP);
    RTDD_LAUNCH_CHECK(ctx, "k_prepare");
    return RTDD_OK;
}

int launch_index_to_weight(rtdd_ctx *ctx, const uint8_t *gray, size_t grayPitch, const float *depth, size_t depthPitch,
                           int32_t *index2, int level, int rows, int cols) {
    const int gated = level != ctx->maxLevel;
    const int thr = level == 0 ? 0 : 4;
    hipLaunchKernelGGL(k_index_to_weight, grid64x4(rows, cols), dim3(256), 0, ctx->stream, gray, grayPitch, depth, depthPitch,
                       index2, rows, cols, gated, thr);
    RTDD_LAUNCH_CHECK(ctx, "k_index_to_weight");
    return RTDD_OK;
}

static int pick_rows_per_wave(const rtdd_ctx *ctx, int rows, int cols) {
    if (ctx->opt.rows_per_wave > 0) return ctx->opt.rows_per_wave;
    // aim for >= ~16 waves per CU so HBM latency is covered by occupancy, but keep R >= 2 so a
    // wave's three-row window amortises its two halo rows
    const long strips = (cols + 255) / 256;
    const long want_waves = (long)ctx->num_cus * 16;
    long R = (strips * rows + want_waves - 1) / want_waves;
    if (R < 2) R = 2;
    if (R > 16) R = 16;
    return (int)R;
}

int launch_sweeps(rtdd_ctx *ctx, const Level &L, size_t ip, int rows, int cols, const float *omegas, int n,
                  int *pk, int *pm, int *launches) {
    const float gamma = 0.99;                 // src/GPUSolver.cu:285 (double literal narrowed to float)
    const int R = pick_rows_per_wave(ctx, rows, cols);
    ctx->last_info.kernel = 1; ctx->last_info.tile = 0; ctx->last_info.temporal_depth = 1; ctx->last_info.persistent = 0;
    const dim3 grid((cols + 255) / 256, (rows + 4 * R - 1) / (4 * R));
    int a = *pk, b = *pm;                     // plane a holds x_k, plane b holds x_{k-1} and receives x_{k+1}
    for (int it = 0; it < n; it++) {
        if (ctx->opt.fp_contract)
            hipLaunchKernelGGL(k_sweep1<true>, grid, dim3(256), 0, ctx->stream, L.P(a, ip), L.P(b, ip), L.M(ip), ctx->lut_dev,
                               (int)ip, rows, cols, R, omegas[it], gamma);
        else
            hipLaunchKernelGGL(k_sweep1<false>, grid, dim3(256), 0, ctx->stream, L.P(a, ip), L.P(b, ip), L.M(ip), ctx->lut_dev,
                               (int)ip, rows, cols, R, omegas[it], gamma);
        const int t = a; a = b; b = t;
    }
    RTDD_LAUNCH_CHECK(ctx, "k_sweep1");
    *pk = a; *pm = b;
    *launches = n;
    return RTDD_OK;
}

int launch_finish(rtdd_ctx *ctx, const Level &L, size_t ip, int src_plane, float *depth, size_t depthPitch, int rows, int cols,
                  const SolveTargets &t, int seq) {
    const Batch &B = t.batch;
    const size_t zP = L.elems * sizeof(float);
    if ((uintptr_t)depth % 16 == 0 && depthPitch % 16 == 0 && B.depth % 16 == 0 && B.u8 % 4 == 0)
        hipLaunchKernelGGL(k_finish4, grid64x4(rows, (cols + 3) / 4, B.n), dim3(256), 0, ctx->stream, L.P(src_plane, ip), (int)ip, depth, depthPitch,
                           rows, cols, t.u8, t.u8_pitch, ctx->sync_words, seq, zP, B.depth, B.u8, t.u8b, t.u8b_pitch);
    else
        hipLaunchKernelGGL(k_finish, grid64x4(rows, cols, B.n), dim3(256), 0, ctx->stream, L.P(src_plane, ip), (int)ip, depth, depthPitch,
                           rows, cols, t.u8, t.u8_pitch, ctx->sync_words, seq, zP, B.depth, B.u8, t.u8b, t.u8b_pitch);
    note_publisher(ctx, seq);                     // (the guard may have recorded a failed solve: the next synchronising call looks)
    RTDD_LAUNCH_CHECK(ctx, "k_finish");
    return RTDD_OK;
}

int launch_residual(rtdd_ctx *ctx, const Level &L, size_t ip, int plane, int rows, int cols, float *host_out) {
    RTDD_HIP(ctx, hipMemsetAsync(ctx->residual_dev, 0, sizeof(float), ctx->stream));
    if (ctx->opt.fp_contract)
        hipLaunchKernelGGL(k_residual<true>, dim3((cols + 255) / 256, (rows + 3) / 4), dim3(256), 0, ctx->stream, L.P(plane, ip), L.M(ip), ctx->lut_dev,
                           (int)ip, rows, cols, (unsigned int *)ctx->residual_dev);
    else
        hipLaunchKernelGGL(k_residual<false>, dim3((cols + 255) / 256, (rows + 3) / 4), dim3(256), 0, ctx->stream, L.P(plane, ip), L.M(ip), ctx->lut_dev,
                           (int)ip, rows, cols, (unsigned int *)ctx->residual_dev);
    RTDD_LAUNCH_CHECK(ctx, "k_residual");
    RTDD_HIP(ctx, hipMemcpyAsync(host_out, ctx->residual_dev, sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    RTDD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return check_persistent_status(ctx, true);  // the sweeps this residual judges may have been a persistent launch that gave up
}

int launch_rbgs(rtdd_ctx *ctx, const Level &L, size_t ip, int plane, int rows, int cols, int nsweeps, float omega) {
    ctx->last_info.kernel = 3; ctx->last_info.tile = 0; ctx->last_info.temporal_depth = 1; ctx->last_info.persistent = 0;
    const dim3 grid(((cols + 1) / 2 + 63) / 64, (rows + 3) / 4);
    for (int s = 0; s < nsweeps; s++)
        for (int colour = 0; colour < 2; colour++) {
            if (ctx->opt.fp_contract)
                hipLaunchKernelGGL(k_rbgs_half<true>, grid, dim3(256), 0, ctx->stream, L.P(plane, ip), L.M(ip), ctx->lut_dev, (int)ip, rows, cols, colour, omega);
            else
                hipLaunchKernelGGL(k_rbgs_half<false>, grid, dim3(256), 0, ctx->stream, L.P(plane, ip), L.M(ip), ctx->lut_dev, (int)ip, rows, cols, colour, omega);
        }
    RTDD_LAUNCH_CHECK(ctx, "k_rbgs_half");
    return RTDD_OK;
}

}  // namespace rtdd
