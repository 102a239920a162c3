// effect_kernels.hip -- depth-driven artistic passes (desaturation, haze, defocus).
//
// Desaturation and haze are pure streaming (11 / 10 B per pixel).  Defocus in the reference is a
// per-pixel O(k^2) gather (up to 48 400 taps at 8K, /root/reference/src/GPUDepthEffect.cu:47-60);
// here it is an exact O(1) lookup in a u32 summed-area table: window sums are < 2^24 so the
// reference's f32 accumulation is exact, and mod-2^32 subtraction of wrapped prefixes is exact
// too, so the results are bit-identical while the cost no longer depends on the blur radius.
#include "rtdd_internal.hpp"

namespace rtdd {

__device__ __forceinline__ uint8_t store_u8(float v) {
    // defined behaviour for the reference's out-of-range float->uchar cast: saturate, then truncate
    if (!(v >= 0.0f)) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)(int)v;
}

// simulateDesaturation (K8) -- src/GPUDepthEffect.cu:8-27
template <bool CONTRACT>
__global__ __launch_bounds__(256) void k_desaturate(const uint8_t *__restrict__ orig, size_t op, const uint8_t *__restrict__ gray, size_t gp,
                                                    const float *__restrict__ depth, size_t dp, uint8_t *__restrict__ art, size_t ap,
                                                    int rows, int cols) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    const float d = ((const float *)((const char *)depth + (size_t)y * dp))[x];
    const float f = (float)((double)d / 255.0);                     // :22 (double divide, narrowed)
    const float g = (float)gray[(size_t)y * gp + x];
    const uint8_t *o = orig + (size_t)y * op + 3 * x;
    uint8_t *a = art + (size_t)y * ap + 3 * x;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float t = (1 - f) * (float)o[c];
        a[c] = store_u8(CONTRACT ? __builtin_fmaf(f, g, t) : f * g + t);
    }
}

// simulateHaze (K10) -- src/GPUDepthEffect.cu:74-93.  exp is evaluated in f64 and rounded once to
// f32: that is the correctly rounded expf in all but ~2^-29 of cases, which is also what the
// host libm the oracle uses delivers -- so the two agree wherever either is correctly rounded.
template <bool CONTRACT>
__global__ __launch_bounds__(256) void k_haze(const uint8_t *__restrict__ orig, size_t op, const float *__restrict__ depth, size_t dp,
                                              uint8_t *__restrict__ art, size_t ap, int rows, int cols) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    const float d = ((const float *)((const char *)depth + (size_t)y * dp))[x];
    const float arg = (float)((double)(-2.0f * d) / 255.0);         // :88
    const float t = (float)exp((double)arg);
    const float w = (1 - t) * 255;
    const uint8_t *o = orig + (size_t)y * op + 3 * x;
    uint8_t *a = art + (size_t)y * ap + 3 * x;
#pragma unroll
    for (int c = 0; c < 3; c++) a[c] = store_u8(CONTRACT ? __builtin_fmaf(t, (float)o[c], w) : t * (float)o[c] + w);
}

// ---- defocus: summed-area table ------------------------------------------------------------------
// S has (rows+1) x (cols+1) entries of 3 x u32 (interleaved), S[0][*] = S[*][0] = 0,
// S[y+1][x+1][c] = sum over y'<=y, x'<=x of orig[y'][x'][c]   (mod 2^32).
//
// pass 1: one workgroup per image row writes the row-wise inclusive prefix into S[y+1][1..].
__global__ __launch_bounds__(256) void k_sat_rows(const uint8_t *__restrict__ orig, size_t op, uint32_t *__restrict__ S, int rows, int cols) {
    __shared__ uint32_t part[256][3];
    const int y = blockIdx.x;
    const int t = threadIdx.x;
    const int chunk = (cols + 255) / 256;
    const int xa = min(t * chunk, cols), xb = min(xa + chunk, cols);
    const uint8_t *o = orig + (size_t)y * op;
    uint32_t s0 = 0, s1 = 0, s2 = 0;
    for (int x = xa; x < xb; x++) { s0 += o[3 * x]; s1 += o[3 * x + 1]; s2 += o[3 * x + 2]; }
    part[t][0] = s0; part[t][1] = s1; part[t][2] = s2;
    __syncthreads();
    // Hillis-Steele inclusive scan over the 256 chunk sums
    for (int off = 1; off < 256; off <<= 1) {
        uint32_t a0 = 0, a1 = 0, a2 = 0;
        if (t >= off) { a0 = part[t - off][0]; a1 = part[t - off][1]; a2 = part[t - off][2]; }
        __syncthreads();
        part[t][0] += a0; part[t][1] += a1; part[t][2] += a2;
        __syncthreads();
    }
    uint32_t r0 = part[t][0] - s0, r1 = part[t][1] - s1, r2 = part[t][2] - s2;    // exclusive prefix of this chunk
    uint32_t *srow = S + ((size_t)(y + 1) * (cols + 1)) * 3;
    if (t == 0) { srow[0] = 0; srow[1] = 0; srow[2] = 0; }
    for (int x = xa; x < xb; x++) {
        r0 += o[3 * x]; r1 += o[3 * x + 1]; r2 += o[3 * x + 2];
        uint32_t *q = srow + (size_t)(x + 1) * 3;
        q[0] = r0; q[1] = r1; q[2] = r2;
    }
    if (y == 0) for (int i = t; i < (cols + 1) * 3; i += 256) S[i] = 0;
}

// pass 2: column-wise prefix in place; thread per (x, channel) word, consecutive threads on
// consecutive words so each row step is one coalesced read-modify-write.
__global__ __launch_bounds__(256) void k_sat_cols(uint32_t *__restrict__ S, int rows, int width3) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= width3) return;
    uint32_t acc = 0;
    uint32_t *p = S + (size_t)width3 + i;          // row 1
    int y = 0;
    for (; y + 4 <= rows; y += 4) {                // 4 independent loads in flight per step
        const uint32_t a = p[0], b = p[(size_t)width3], c = p[(size_t)2 * width3], d = p[(size_t)3 * width3];
        acc += a; p[0] = acc;
        acc += b; p[(size_t)width3] = acc;
        acc += c; p[(size_t)2 * width3] = acc;
        acc += d; p[(size_t)3 * width3] = acc;
        p += (size_t)4 * width3;
    }
    for (; y < rows; y++) { acc += p[0]; p[0] = acc; p += width3; }
}

// simulateDefocus (K9) -- src/GPUDepthEffect.cu:29-72
__global__ __launch_bounds__(256) void k_defocus(const uint8_t *__restrict__ orig, size_t op, const float *__restrict__ depth, size_t dp,
                                                 const uint32_t *__restrict__ S, uint8_t *__restrict__ art, size_t ap,
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
    const size_t w3 = (size_t)(cols + 1) * 3;
    const uint32_t *s00 = S + (size_t)ya * w3 + (size_t)xa * 3, *s01 = S + (size_t)ya * w3 + (size_t)xb * 3;
    const uint32_t *s10 = S + (size_t)yb * w3 + (size_t)xa * 3, *s11 = S + (size_t)yb * w3 + (size_t)xb * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const uint32_t sum = s11[c] - s01[c] - s10[c] + s00[c];     // exact: true window sum < 2^24
        a[c] = store_u8((float)sum / count);                        // :68-70
    }
}

static inline dim3 grid64x4(int rows, int cols) { return dim3((cols + 63) / 64, (rows + 3) / 4); }

int launch_desaturate(rtdd_ctx *ctx, const uint8_t *orig, size_t op, const uint8_t *gray, size_t gp, const float *depth, size_t dp,
                      uint8_t *art, size_t ap, int rows, int cols) {
    if (ctx->opt.fp_contract)
        hipLaunchKernelGGL(k_desaturate<true>, grid64x4(rows, cols), dim3(256), 0, ctx->stream, orig, op, gray, gp, depth, dp, art, ap, rows, cols);
    else
        hipLaunchKernelGGL(k_desaturate<false>, grid64x4(rows, cols), dim3(256), 0, ctx->stream, orig, op, gray, gp, depth, dp, art, ap, rows, cols);
    RTDD_LAUNCH_CHECK(ctx, "k_desaturate");
    return RTDD_OK;
}

int launch_haze(rtdd_ctx *ctx, const uint8_t *orig, size_t op, const float *depth, size_t dp, uint8_t *art, size_t ap, int rows, int cols) {
    if (ctx->opt.fp_contract)
        hipLaunchKernelGGL(k_haze<true>, grid64x4(rows, cols), dim3(256), 0, ctx->stream, orig, op, depth, dp, art, ap, rows, cols);
    else
        hipLaunchKernelGGL(k_haze<false>, grid64x4(rows, cols), dim3(256), 0, ctx->stream, orig, op, depth, dp, art, ap, rows, cols);
    RTDD_LAUNCH_CHECK(ctx, "k_haze");
    return RTDD_OK;
}

int launch_defocus(rtdd_ctx *ctx, const uint8_t *orig, size_t op, const float *depth, size_t dp, uint8_t *art, size_t ap, int rows, int cols) {
    const size_t need = (size_t)(rows + 1) * (cols + 1) * 3;
    if (ctx->sat_elems < need) {
        if (ctx->sat) { RTDD_HIP(ctx, hipStreamSynchronize(ctx->stream)); RTDD_HIP(ctx, hipFree(ctx->sat)); ctx->sat = nullptr; ctx->sat_elems = 0; }
        RTDD_HIP(ctx, hipMalloc((void **)&ctx->sat, need * sizeof(uint32_t)));
        ctx->sat_elems = need;
    }
    const int kernelSize = 0.025 * sqrtf(rows * rows + cols * cols);    // :42, evaluated once on the host (sqrtf is correctly rounded on both)
    hipLaunchKernelGGL(k_sat_rows, dim3(rows), dim3(256), 0, ctx->stream, orig, op, ctx->sat, rows, cols);
    RTDD_LAUNCH_CHECK(ctx, "k_sat_rows");
    const int width3 = (cols + 1) * 3;
    hipLaunchKernelGGL(k_sat_cols, dim3((width3 + 255) / 256), dim3(256), 0, ctx->stream, ctx->sat, rows, width3);
    RTDD_LAUNCH_CHECK(ctx, "k_sat_cols");
    hipLaunchKernelGGL(k_defocus, grid64x4(rows, cols), dim3(256), 0, ctx->stream, orig, op, depth, dp, ctx->sat, art, ap, rows, cols, kernelSize);
    RTDD_LAUNCH_CHECK(ctx, "k_defocus");
    return RTDD_OK;
}

}  // namespace rtdd
