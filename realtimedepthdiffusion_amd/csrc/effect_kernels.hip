// effect_kernels.hip -- depth-driven artistic passes (desaturation, haze, defocus).
//
// Desaturation and haze are pure streaming (11 / 10 B per pixel).  A thread handles FOUR pixels so
// the interleaved u8x3 rows move as dwordx3 (12 B/lane, contiguous across the wave) and depth as
// dwordx4, instead of the reference's one-thread-per-pixel byte accesses
// (/root/reference/src/GPUDepthEffect.cu:18-25, 83-91); rows that are not 4-byte aligned take a
// byte-wise path with identical arithmetic.
//
// Defocus in the reference is a per-pixel O(k^2) gather (up to 48 400 taps at 8K,
// src/GPUDepthEffect.cu:47-60); here it is an exact O(1) lookup in a u32 summed-area table: window
// sums are < 2^24 so the reference's f32 accumulation is exact, and mod-2^32 subtraction of wrapped
// prefixes is exact too, so results are bit-identical while the cost no longer depends on the blur
// radius.  The table is built in three passes chosen for parallelism on 256 CUs:
//   k_sat_rows   one workgroup per image row: each wave scans a quarter of the row 64 pixels at a time
//                (lane = pixel, shuffle scan), 12-byte {B,G,R} prefixes stored contiguously;
//   k_sat_bands  column prefixes INSIDE bands of 32 rows (rows/32 x cols/256 workgroups instead of a
//                1080-step serial walk), band totals on the side;
//   k_sat_base   exclusive scan of the band totals (tiny);
// and k_defocus adds the band base while fetching its four corners (one dwordx3 each).
#include "rtdd_internal.hpp"

namespace rtdd {

__device__ __forceinline__ uint32_t store_u8(float v) {
    // defined behaviour for the reference's out-of-range float->uchar cast: saturate, then truncate
    if (!(v >= 0.0f)) return 0;
    if (v >= 255.0f) return 255;
    return (uint32_t)(int)v;
}

// simulateDesaturation (K8) -- src/GPUDepthEffect.cu:8-27
template <bool CONTRACT>
__device__ __forceinline__ uint32_t desat_px(float d, float g, float o) {
    const float f = (float)((double)d / 255.0);                    // :22 (double divide, narrowed)
    const float t = (1 - f) * o;
    return store_u8(CONTRACT ? __builtin_fmaf(f, g, t) : f * g + t);
}

// simulateHaze (K10) -- src/GPUDepthEffect.cu:74-93.  exp is evaluated in f64 and rounded once to f32: the
// correctly rounded expf in all but ~2^-29 of cases, which is also what the host libm the oracle uses delivers.
__device__ __forceinline__ float haze_t(float d) {
    const float arg = (float)((double)(-2.0f * d) / 255.0);        // :88
    return (float)exp((double)arg);
}
template <bool CONTRACT>
__device__ __forceinline__ uint32_t haze_px(float t, float w, float o) {
    return store_u8(CONTRACT ? __builtin_fmaf(t, o, w) : t * o + w);
}

// MODE 0 = desaturation, 1 = haze.  VEC: 4 pixels per thread with dword accesses (needs 4-byte aligned rows).
template <int MODE, bool CONTRACT, bool VEC>
__global__ __launch_bounds__(256) void k_blend(const uint8_t *__restrict__ orig, size_t op, const uint8_t *__restrict__ gray, size_t gp,
                                               const float *__restrict__ depth, size_t dp, uint8_t *__restrict__ art, size_t ap,
                                               int rows, int cols) {
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (y >= rows) return;
    const float *drow = (const float *)((const char *)depth + (size_t)y * dp);
    const uint8_t *orow = orig + (size_t)y * op;
    uint8_t *arow = art + (size_t)y * ap;
    if (VEC) {
        const int x = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
        if (x + 3 < cols) {
            const float4 d4 = *(const float4 *)(drow + x);
            const uint32_t *o3 = (const uint32_t *)(orow + 3 * x);
            const uint32_t w0 = o3[0], w1 = o3[1], w2 = o3[2];
            uint32_t g4 = 0;
            if (MODE == 0) g4 = *(const uint32_t *)(gray + (size_t)y * gp + x);
            const float dv[4] = {d4.x, d4.y, d4.z, d4.w};
            uint32_t ob[12], rb[12];
#pragma unroll
            for (int i = 0; i < 4; i++) { ob[i] = (w0 >> (8 * i)) & 255; ob[4 + i] = (w1 >> (8 * i)) & 255; ob[8 + i] = (w2 >> (8 * i)) & 255; }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (MODE == 0) {
                    const float g = (float)((g4 >> (8 * i)) & 255);
#pragma unroll
                    for (int c = 0; c < 3; c++) rb[3 * i + c] = desat_px<CONTRACT>(dv[i], g, (float)ob[3 * i + c]);
                } else {
                    const float t = haze_t(dv[i]);
                    const float w = (1 - t) * 255;
#pragma unroll
                    for (int c = 0; c < 3; c++) rb[3 * i + c] = haze_px<CONTRACT>(t, w, (float)ob[3 * i + c]);
                }
            }
            uint32_t *a3 = (uint32_t *)(arow + 3 * x);
            a3[0] = rb[0] | (rb[1] << 8) | (rb[2] << 16) | (rb[3] << 24);
            a3[1] = rb[4] | (rb[5] << 8) | (rb[6] << 16) | (rb[7] << 24);
            a3[2] = rb[8] | (rb[9] << 8) | (rb[10] << 16) | (rb[11] << 24);
            return;
        }
        // ragged tail of a vectorised row: fall through to the scalar body for the remaining pixels
        for (int xx = x; xx < cols; xx++) {
            const float d = drow[xx];
            if (MODE == 0) { const float g = (float)gray[(size_t)y * gp + xx];
                for (int c = 0; c < 3; c++) arow[3 * xx + c] = (uint8_t)desat_px<CONTRACT>(d, g, (float)orow[3 * xx + c]); }
            else { const float t = haze_t(d); const float w = (1 - t) * 255;
                for (int c = 0; c < 3; c++) arow[3 * xx + c] = (uint8_t)haze_px<CONTRACT>(t, w, (float)orow[3 * xx + c]); }
        }
    } else {
        const int x = blockIdx.x * 64 + (threadIdx.x & 63);
        if (x >= cols) return;
        const float d = drow[x];
        if (MODE == 0) { const float g = (float)gray[(size_t)y * gp + x];
#pragma unroll
            for (int c = 0; c < 3; c++) arow[3 * x + c] = (uint8_t)desat_px<CONTRACT>(d, g, (float)orow[3 * x + c]); }
        else { const float t = haze_t(d); const float w = (1 - t) * 255;
#pragma unroll
            for (int c = 0; c < 3; c++) arow[3 * x + c] = (uint8_t)haze_px<CONTRACT>(t, w, (float)orow[3 * x + c]); }
    }
}

// ---- defocus: summed-area table ------------------------------------------------------------------
// L has (rows+1) x (cols+1) entries of 3 x u32 {B, G, R} (12 B, moved as dwordx3): L[0][*] = L[*][0] = 0 and, for r >= 1,
// L[r][c] = sum over rows of r's band up to r-1, columns < c (mod 2^32).  The full prefix is
// S(r,c) = L[r][c] + base[(r-1)/kBand][c].
constexpr int kBand = 32;

struct u3 { uint32_t x, y, z; };                                    // 12-byte table entry
__device__ __forceinline__ u3 ld3(const u3 *p) { return *p; }
__device__ __forceinline__ void st3(u3 *p, uint32_t a, uint32_t b, uint32_t c) { u3 v; v.x = a; v.y = b; v.z = c; *p = v; }

// inclusive prefix sum over the 64 lanes of a wave: the 7-step DPP scan (3 row shifts of the input, row_shr:4 / :8 under bank
// masks, row_bcast:15 / :31 under row masks) -- VALU only, no LDS crossbar
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
#define RTDD_DPP(src, ctrl, rows, banks) (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(src), ctrl, rows, banks, true)
    uint32_t t = v + RTDD_DPP(v, 0x111, 0xF, 0xF);      // row_shr:1
    t += RTDD_DPP(v, 0x112, 0xF, 0xF);                  // row_shr:2
    t += RTDD_DPP(v, 0x113, 0xF, 0xF);                  // row_shr:3   -> v[i-3..i] within a row of 16
    t += RTDD_DPP(t, 0x114, 0xF, 0xE);                  // row_shr:4, banks 1-3
    t += RTDD_DPP(t, 0x118, 0xF, 0xC);                  // row_shr:8, banks 2-3  -> prefix within each row of 16
    t += RTDD_DPP(t, 0x142, 0xA, 0xF);                  // row_bcast:15 into rows 1 and 3
    t += RTDD_DPP(t, 0x143, 0xC, 0xF);                  // row_bcast:31 into rows 2 and 3
#undef RTDD_DPP
    return t;
}

// pass 1: row-wise inclusive prefix of image row y into L[y+1][1..].  One workgroup per row, one wave per 512-pixel segment (4 waves
// at 1080p, 8 at 4K, 16 at 8K; wider rows loop).  A wave loads its whole segment at once -- 8 groups of 64 pixels, lane = pixel,
// so the 3-byte loads and the 12-byte stores of a wave instruction are contiguous -- runs the 24 independent DPP scans, chains
// the group totals in scalar registers, trades segment totals through LDS (one barrier) and stores.  One pass over the row, every
// load in flight together.
constexpr int kSatGroups = 8;                                       // 64-pixel groups per wave and pass
__global__ __launch_bounds__(1024) void k_sat_rows(const uint8_t *__restrict__ orig, size_t op, u3 *__restrict__ L, int rows, int cols) {
    __shared__ uint32_t seg[2][16][3];
    const int y = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (int)blockDim.x >> 6;
    const uint8_t *o = orig + (size_t)y * op;
    u3 *lrow = L + (size_t)(y + 1) * (cols + 1);
    if (threadIdx.x == 0) st3(lrow, 0, 0, 0);
    uint32_t carry[3] = {0, 0, 0};                                  // everything left of this pass (wave-uniform)
    int buf = 0;
    for (int x0 = 0; x0 < cols; x0 += nw * 64 * kSatGroups, buf ^= 1) {
        const int xs = x0 + w * 64 * kSatGroups + lane;
        uint32_t v[kSatGroups][3];
#pragma unroll
        for (int g = 0; g < kSatGroups; g++) {
            const int x = xs + 64 * g;
#pragma unroll
            for (int c = 0; c < 3; c++) v[g][c] = x < cols ? o[3 * (size_t)x + c] : 0u;
        }
        uint32_t run[3] = {0, 0, 0};                                // groups of this wave so far (wave-uniform)
#pragma unroll
        for (int g = 0; g < kSatGroups; g++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                v[g][c] = wave_incl_scan(v[g][c]) + run[c];
                run[c] = (uint32_t)__builtin_amdgcn_readlane((int)v[g][c], 63);
            }
        if (lane == 0) { seg[buf][w][0] = run[0]; seg[buf][w][1] = run[1]; seg[buf][w][2] = run[2]; }
        __syncthreads();                                            // (seg is double buffered: one barrier per pass)
        uint32_t add[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            uint32_t before = 0, all = 0;
            for (int k = 0; k < nw; k++) { const uint32_t t = seg[buf][k][c]; all += t; if (k < w) before += t; }
            add[c] = carry[c] + before;
            carry[c] += all;
        }
#pragma unroll
        for (int g = 0; g < kSatGroups; g++) {
            const int x = xs + 64 * g;
            if (x < cols) st3(lrow + x + 1, v[g][0] + add[0], v[g][1] + add[1], v[g][2] + add[2]);
        }
    }
    if (y == 0) for (int i = threadIdx.x; i <= cols; i += (int)blockDim.x) st3(L + i, 0, 0, 0);
}

// pass 2: column prefix inside each band of kBand rows, in place; band totals -> tot[band][c]
__global__ __launch_bounds__(256) void k_sat_bands(u3 *__restrict__ L, u3 *__restrict__ tot, int rows, int width) {
    const int c = blockIdx.x * 256 + threadIdx.x, band = blockIdx.y;
    if (c >= width) return;
    const int ra = band * kBand + 1, rb = min(ra + kBand, rows + 1);
    uint32_t a0 = 0, a1 = 0, a2 = 0;
    u3 *p = L + (size_t)ra * width + c;
    int r = ra;
    for (; r + 4 <= rb; r += 4) {                                   // 4 independent loads in flight
        const u3 a = ld3(p), b = ld3(p + (size_t)width), cc = ld3(p + (size_t)2 * width), d = ld3(p + (size_t)3 * width);
        a0 += a.x; a1 += a.y; a2 += a.z; st3(p, a0, a1, a2);
        a0 += b.x; a1 += b.y; a2 += b.z; st3(p + (size_t)width, a0, a1, a2);
        a0 += cc.x; a1 += cc.y; a2 += cc.z; st3(p + (size_t)2 * width, a0, a1, a2);
        a0 += d.x; a1 += d.y; a2 += d.z; st3(p + (size_t)3 * width, a0, a1, a2);
        p += (size_t)4 * width;
    }
    for (; r < rb; r++) { const u3 a = ld3(p); a0 += a.x; a1 += a.y; a2 += a.z; st3(p, a0, a1, a2); p += width; }
    st3(tot + (size_t)band * width + c, a0, a1, a2);
}

// pass 3: base[b][c] = sum of the totals of bands < b (in place on tot); 8 loads in flight per step
__global__ __launch_bounds__(256) void k_sat_base(u3 *__restrict__ tot, int nbands, int width) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= width) return;
    uint32_t a0 = 0, a1 = 0, a2 = 0;
    for (int b0 = 0; b0 < nbands; b0 += 8) {
        u3 t[8];
#pragma unroll
        for (int j = 0; j < 8; j++) if (b0 + j < nbands) t[j] = ld3(tot + (size_t)(b0 + j) * width + c);
#pragma unroll
        for (int j = 0; j < 8; j++) if (b0 + j < nbands) { st3(tot + (size_t)(b0 + j) * width + c, a0, a1, a2); a0 += t[j].x; a1 += t[j].y; a2 += t[j].z; }
    }
}

__device__ __forceinline__ u3 sat_at(const u3 *__restrict__ L, const u3 *__restrict__ base, int width, int r, int c) {
    u3 v; v.x = 0; v.y = 0; v.z = 0;
    if (r == 0) return v;
    const u3 a = ld3(L + (size_t)r * width + c), b = ld3(base + (size_t)((r - 1) / kBand) * width + c);
    v.x = a.x + b.x; v.y = a.y + b.y; v.z = a.z + b.z;
    return v;
}

// simulateDefocus (K9) -- src/GPUDepthEffect.cu:29-72
__global__ __launch_bounds__(256) void k_defocus(const uint8_t *__restrict__ orig, size_t op, const float *__restrict__ depth, size_t dp,
                                                 const u3 *__restrict__ L, const u3 *__restrict__ base, uint8_t *__restrict__ art, size_t ap,
                                                 int rows, int cols, int kernelSize) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    const float d = ((const float *)((const char *)depth + (size_t)y * dp))[x];
    const double kd = (double)((float)kernelSize * d) / 255.0;      // :43  int*float -> float, / double
    int k;                                                          // (int) of a double: define the UB cases
    if (!(kd > -2147483648.0)) k = 0; else if (kd >= 2147483647.0) k = 2147483647; else k = (int)kd;
    const int h = k / 2;                                            // C division truncates toward zero
    const int ya = max(y - h, 0), yb = min(y + h, rows);
    const int xa = max(x - h, 0), xb = min(x + h, cols);
    const uint8_t *o = orig + (size_t)y * op + 3 * x;
    uint8_t *a = art + (size_t)y * ap + 3 * x;
    if (h <= 0 || yb <= ya || xb <= xa) {                           // count == 0 (:62-66)
        a[0] = o[0]; a[1] = o[1]; a[2] = o[2];
        return;
    }
    const float count = (float)((yb - ya) * (xb - xa));
    const int width = cols + 1;
    const u3 s00 = sat_at(L, base, width, ya, xa), s01 = sat_at(L, base, width, ya, xb);
    const u3 s10 = sat_at(L, base, width, yb, xa), s11 = sat_at(L, base, width, yb, xb);
    // exact: the true window sum is < 2^24, so mod-2^32 arithmetic and the f32 conversion lose nothing (:68-70)
    a[0] = (uint8_t)store_u8((float)(s11.x - s01.x - s10.x + s00.x) / count);
    a[1] = (uint8_t)store_u8((float)(s11.y - s01.y - s10.y + s00.y) / count);
    a[2] = (uint8_t)store_u8((float)(s11.z - s01.z - s10.z + s00.z) / count);
}

static inline dim3 grid64x4(int rows, int cols) { return dim3((cols + 63) / 64, (rows + 3) / 4); }

template <int MODE>
static int launch_blend(rtdd_ctx *ctx, const uint8_t *orig, size_t op, const uint8_t *gray, size_t gp, const float *depth, size_t dp,
                        uint8_t *art, size_t ap, int rows, int cols) {
    const bool aligned = ((uintptr_t)orig % 4 == 0) && ((uintptr_t)art % 4 == 0) && op % 4 == 0 && ap % 4 == 0 &&
                         ((uintptr_t)depth % 16 == 0) && dp % 16 == 0 && (MODE == 1 || (((uintptr_t)gray % 4 == 0) && gp % 4 == 0));
    const bool c = ctx->opt.fp_contract != 0;
    if (aligned) {
        const dim3 grid((cols + 255) / 256, (rows + 3) / 4);
        if (c) hipLaunchKernelGGL((k_blend<MODE, true, true>), grid, dim3(256), 0, ctx->stream, orig, op, gray, gp, depth, dp, art, ap, rows, cols);
        else hipLaunchKernelGGL((k_blend<MODE, false, true>), grid, dim3(256), 0, ctx->stream, orig, op, gray, gp, depth, dp, art, ap, rows, cols);
    } else {
        if (c) hipLaunchKernelGGL((k_blend<MODE, true, false>), grid64x4(rows, cols), dim3(256), 0, ctx->stream, orig, op, gray, gp, depth, dp, art, ap, rows, cols);
        else hipLaunchKernelGGL((k_blend<MODE, false, false>), grid64x4(rows, cols), dim3(256), 0, ctx->stream, orig, op, gray, gp, depth, dp, art, ap, rows, cols);
    }
    RTDD_LAUNCH_CHECK(ctx, "k_blend");
    return RTDD_OK;
}

int launch_desaturate(rtdd_ctx *ctx, const uint8_t *orig, size_t op, const uint8_t *gray, size_t gp, const float *depth, size_t dp,
                      uint8_t *art, size_t ap, int rows, int cols) {
    return launch_blend<0>(ctx, orig, op, gray, gp, depth, dp, art, ap, rows, cols);
}

int launch_haze(rtdd_ctx *ctx, const uint8_t *orig, size_t op, const float *depth, size_t dp, uint8_t *art, size_t ap, int rows, int cols) {
    return launch_blend<1>(ctx, orig, op, nullptr, 0, depth, dp, art, ap, rows, cols);
}

int launch_defocus(rtdd_ctx *ctx, const uint8_t *orig, size_t op, const float *depth, size_t dp, uint8_t *art, size_t ap, int rows, int cols) {
    const int width = cols + 1, nbands = (rows + kBand - 1) / kBand;
    const size_t need = ((size_t)(rows + 1) * width + (size_t)nbands * width) * 3 + 16;     // in u32 words (3-word entries)
    if (ctx->sat_elems < need) {
        if (ctx->sat) { RTDD_HIP(ctx, hipStreamSynchronize(ctx->stream)); RTDD_HIP(ctx, hipFree(ctx->sat)); ctx->sat = nullptr; ctx->sat_elems = 0; }
        RTDD_HIP(ctx, hipMalloc((void **)&ctx->sat, need * sizeof(uint32_t)));
        ctx->sat_elems = need;
    }
    u3 *L = (u3 *)ctx->sat, *base = L + (size_t)(rows + 1) * width;
    const int kernelSize = 0.025 * sqrtf(rows * rows + cols * cols);    // :42, evaluated once on the host (sqrtf is correctly rounded on both)
    int sat_waves = (cols + 64 * kSatGroups - 1) / (64 * kSatGroups);      // one wave per 512-pixel segment, at most 16
    if (sat_waves > 16) sat_waves = 16;
    hipLaunchKernelGGL(k_sat_rows, dim3(rows), dim3(64 * sat_waves), 0, ctx->stream, orig, op, L, rows, cols);
    RTDD_LAUNCH_CHECK(ctx, "k_sat_rows");
    hipLaunchKernelGGL(k_sat_bands, dim3((width + 255) / 256, nbands), dim3(256), 0, ctx->stream, L, base, rows, width);
    RTDD_LAUNCH_CHECK(ctx, "k_sat_bands");
    hipLaunchKernelGGL(k_sat_base, dim3((width + 255) / 256), dim3(256), 0, ctx->stream, base, nbands, width);
    RTDD_LAUNCH_CHECK(ctx, "k_sat_base");
    hipLaunchKernelGGL(k_defocus, grid64x4(rows, cols), dim3(256), 0, ctx->stream, orig, op, depth, dp, L, base, art, ap, rows, cols, kernelSize);
    RTDD_LAUNCH_CHECK(ctx, "k_defocus");
    return RTDD_OK;
}

}  // namespace rtdd
