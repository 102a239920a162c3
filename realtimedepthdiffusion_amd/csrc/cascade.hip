// cascade.hip -- SURVEY.md section 8(f) rows 1-2: the pieces of main.cpp's depth-estimate loop
// (/root/reference/src/main.cpp:232-295) that the reference delegates to OpenCV, moved onto the
// device, plus a whole-cascade driver so one estimate is a single stream-ordered launch sequence
// with no host round trips (the reference crosses PCIe 2(P-1)..4(P-1) times per estimate and
// device-syncs P times: main.cpp:244-246, 276-278, src/GPUSolver.cu:314).
//
// The OpenCV ops are THIRD PARTY and unpinned: the kernels below implement
// the published formulas restated in oracle/rtdd_cascade_oracle.c and agree with it bit for bit.
#include <cmath>

#include "rtdd_internal.hpp"
#include "persist_sync.hpp"

namespace rtdd {

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

// cv::cvtColor(BGR2GRAY), 8u fixed point -- src/main.cpp:111,138
__global__ __launch_bounds__(256) void k_bgr2gray(const uint8_t *__restrict__ bgr, size_t bp, uint8_t *__restrict__ gray, size_t gp, int rows, int cols) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + wave_id();
    if (x >= cols || y >= rows) return;
    const uint8_t *p = bgr + (size_t)y * bp + 3 * x;
    gray[(size_t)y * gp + x] = (uint8_t)((p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + (1 << 13)) >> 14);
}

// cv::pyrDown (8u): 5x5 [1 4 6 4 1]/16 separable, reflect-101, (s+128)>>8 -- src/main.cpp:112,144,245
__global__ __launch_bounds__(256) void k_pyrdown_u8(const uint8_t *__restrict__ src, size_t sp, int rows, int cols,
                                                    uint8_t *__restrict__ dst, size_t dp, int drows, int dcols) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + wave_id();
    if (x >= dcols || y >= drows) return;
    const int k[5] = {1, 4, 6, 4, 1};
    int cx[5];
#pragma unroll
    for (int i = 0; i < 5; i++) cx[i] = reflect101(2 * x + i - 2, cols);
    int s = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const uint8_t *row = src + (size_t)reflect101(2 * y + j - 2, rows) * sp;
        int h = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) h += k[i] * row[cx[i]];
        s += k[j] * h;
    }
    dst[(size_t)y * dp + x] = (uint8_t)((s + 128) >> 8);
}

// f32 pyrUp as src/main.cpp:272-279 calls it, with the Dirichlet re-injection of GPUConvertToFloat (src/main.cpp:281-283,
// src/GPUImageProcessing.cu:19) fused in.  The reference uses TWO OpenCV routines: cv::cuda::pyrUp when the destination is exactly
// twice the source (main.cpp:273), the host's cv::pyrUp with the explicit size otherwise (main.cpp:277).  Both mirror at the
// top/left border and REPLICATE at the bottom/right; they differ in the border arithmetic and in that the CUDA one accumulates
// sum = sum + w * s, which nvcc contracts (CONTRACT, as for the solver).  Formulas and what is unknowable about them:
// oracle/rtdd_cascade_oracle.c (third-party code, unpinned).
__device__ __forceinline__ int clamp_abs(int i, int n) { if (i < 0) i = -i; return i < n - 1 ? i : n - 1; }

template <bool CONTRACT>
__device__ __forceinline__ float pyrup_cuda_h(const float *s, int n, int k) {
    const int c = k >> 1;
    float sum = 0.0f;
    if ((k & 1) == 0) {
        const float a = s[clamp_abs(c - 1, n)], b = s[clamp_abs(c, n)], d = s[clamp_abs(c + 1, n)];
        if (CONTRACT) { sum = __builtin_fmaf(0.0625f, a, sum); sum = __builtin_fmaf(0.375f, b, sum); sum = __builtin_fmaf(0.0625f, d, sum); }
        else { sum = sum + 0.0625f * a; sum = sum + 0.375f * b; sum = sum + 0.0625f * d; }
    } else {
        const float a = s[clamp_abs(c, n)], b = s[clamp_abs(c + 1, n)];
        if (CONTRACT) { sum = __builtin_fmaf(0.25f, a, sum); sum = __builtin_fmaf(0.25f, b, sum); }
        else { sum = sum + 0.25f * a; sum = sum + 0.25f * b; }
    }
    return sum;
}

__device__ __forceinline__ float pyrup_host_h(const float *s, int n, int k) {
    if (n == 1) return s[0] * 8;
    if (k >= 2 * n) k = 2 * n - 1;
    const int c = k >> 1;
    if ((k & 1) == 0) {
        if (c == 0) return s[0] * 6 + s[1] * 2;
        if (c == n - 1) return s[n - 2] + s[n - 1] * 7;
        return s[c - 1] + s[c] * 6 + s[c + 1];
    }
    if (c == n - 1) return s[n - 1] * 8;
    return (s[c] + s[c + 1]) * 4;
}

template <bool CONTRACT>
__global__ __launch_bounds__(256) void k_pyrup_inject(const float *__restrict__ src, size_t sp, int rows, int cols,
                                                      float *__restrict__ dst, size_t dp, int drows, int dcols,
                                                      const uint8_t *__restrict__ edited, size_t ep,
                                                      const uint8_t *__restrict__ mask, size_t mp,
                                                      float *__restrict__ coarse_out, size_t cp, int *sync_words, int seq,
                                                      size_t zSrc, size_t zDst, size_t zEdited, size_t zMask, size_t zCoarse) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + wave_id();
    if (solve_is_dead(sync_words, seq, (blockIdx.x | blockIdx.y | blockIdx.z | threadIdx.x) == 0)) return;    // (persist_sync.hpp: the coarse solve gave up)
    if (x >= dcols || y >= drows) return;
    RTDD_Z(src, zSrc); RTDD_Z(dst, zDst); if (mask) { RTDD_Z(edited, zEdited); RTDD_Z(mask, zMask); } if (coarse_out) RTDD_Z(coarse_out, zCoarse);
    // the estimate driver reads the coarse level straight from the solver's plane; the caller-visible coarse depth image (the
    // solver's copy-back, src/GPUSolver.cu:311-312) is written here on the side: fine pixel (2c, 2r) stores coarse pixel (c, r)
    // (the fine level is at least twice the coarse one in both directions, so every coarse pixel has one)
    if (coarse_out && !((x | y) & 1) && (x >> 1) < cols && (y >> 1) < rows)
        ((float *)((char *)coarse_out + (size_t)(y >> 1) * cp))[x >> 1] = ((const float *)((const char *)src + (size_t)(y >> 1) * sp))[x >> 1];
    float *out = (float *)((char *)dst + (size_t)y * dp) + x;
    if (mask && mask[(size_t)y * mp + x] == 255) { *out = (float)edited[(size_t)y * ep + 3 * x]; return; }
#define SROW(r) ((const float *)((const char *)src + (size_t)(r) * sp))
    if (drows == 2 * rows && dcols == 2 * cols) {                                  // cv::cuda::pyrUp
        const int cy = y >> 1;
        float sum = 0.0f;
        if ((y & 1) == 0) {
            const float h0 = pyrup_cuda_h<CONTRACT>(SROW(clamp_abs(cy - 1, rows)), cols, x), h1 = pyrup_cuda_h<CONTRACT>(SROW(clamp_abs(cy, rows)), cols, x),
                        h2 = pyrup_cuda_h<CONTRACT>(SROW(clamp_abs(cy + 1, rows)), cols, x);
            if (CONTRACT) { sum = __builtin_fmaf(0.0625f, h0, sum); sum = __builtin_fmaf(0.375f, h1, sum); sum = __builtin_fmaf(0.0625f, h2, sum); }
            else { sum = sum + 0.0625f * h0; sum = sum + 0.375f * h1; sum = sum + 0.0625f * h2; }
        } else {
            const float h1 = pyrup_cuda_h<CONTRACT>(SROW(clamp_abs(cy, rows)), cols, x), h2 = pyrup_cuda_h<CONTRACT>(SROW(clamp_abs(cy + 1, rows)), cols, x);
            if (CONTRACT) { sum = __builtin_fmaf(0.25f, h1, sum); sum = __builtin_fmaf(0.25f, h2, sum); }
            else { sum = sum + 0.25f * h1; sum = sum + 0.25f * h2; }
        }
        *out = 4.0f * sum;
    } else {                                                                       // cv::pyrUp with the explicit size
        const int yy = y >= 2 * rows ? 2 * rows - 2 : y;
        const int cy = yy >> 1;
        const int r0 = cy - 1 < 0 ? (rows > 1 ? 1 : 0) : cy - 1, r2 = cy + 1 >= rows ? rows - 1 : cy + 1;
        const float v1 = pyrup_host_h(SROW(cy), cols, x), v2 = pyrup_host_h(SROW(r2), cols, x);
        if ((yy & 1) == 0) { const float v0 = pyrup_host_h(SROW(r0), cols, x); *out = (v0 + v1 * 6 + v2) * 0.015625f; }
        else *out = ((v1 + v2) * 4) * 0.015625f;
    }
#undef SROW
}

// The same for the exact-doubling case (cv::cuda::pyrUp: every level pair of an even-sized cascade -- 1080p: three of its four), FOUR
// output pixels per thread: the 4 source columns x 3 (2) source rows a group needs are loaded once (12 loads instead of 36), the
// arithmetic per pixel is pyrup_cuda_h's operation for operation, the results leave as one 16-byte store, mask and edited image as
// dwords.  1920 x 1080: 15.5 -> ~6 us.
template <bool CONTRACT>
__global__ __launch_bounds__(256) void k_pyrup_inject4(const float *__restrict__ src, size_t sp, int rows, int cols,
                                                       float *__restrict__ dst, size_t dp, int drows, int dcols,
                                                       const uint8_t *__restrict__ edited, size_t ep,
                                                       const uint8_t *__restrict__ mask, size_t mp,
                                                       float *__restrict__ coarse_out, size_t cp, int *sync_words, int seq,
                                                       size_t zSrc, size_t zDst, size_t zEdited, size_t zMask, size_t zCoarse) {
    const int x0 = 4 * (blockIdx.x * 64 + (threadIdx.x & 63)), y = blockIdx.y * 4 + wave_id();
    if (solve_is_dead(sync_words, seq, (blockIdx.x | blockIdx.y | blockIdx.z | threadIdx.x) == 0)) return;
    if (x0 >= dcols || y >= drows) return;
    RTDD_Z(src, zSrc); RTDD_Z(dst, zDst); if (mask) { RTDD_Z(edited, zEdited); RTDD_Z(mask, zMask); } if (coarse_out) RTDD_Z(coarse_out, zCoarse);
    const int c0 = x0 >> 1, cy = y >> 1;
    const int ci[4] = {clamp_abs(c0 - 1, cols), clamp_abs(c0, cols), clamp_abs(c0 + 1, cols), clamp_abs(c0 + 2, cols)};
    // horizontal pass of one source row for the four outputs (k = x0 .. x0 + 3: even, odd, even, odd)
    auto hrow = [&](int r, float (&h)[4]) {
        const float *s = (const float *)((const char *)src + (size_t)r * sp);
        const float v0 = s[ci[0]], v1 = s[ci[1]], v2 = s[ci[2]], v3 = s[ci[3]];
        float e0 = 0.0f, o0 = 0.0f, e1 = 0.0f, o1 = 0.0f;
        if (CONTRACT) {
            e0 = __builtin_fmaf(0.0625f, v0, e0); e0 = __builtin_fmaf(0.375f, v1, e0); e0 = __builtin_fmaf(0.0625f, v2, e0);
            o0 = __builtin_fmaf(0.25f, v1, o0); o0 = __builtin_fmaf(0.25f, v2, o0);
            e1 = __builtin_fmaf(0.0625f, v1, e1); e1 = __builtin_fmaf(0.375f, v2, e1); e1 = __builtin_fmaf(0.0625f, v3, e1);
            o1 = __builtin_fmaf(0.25f, v2, o1); o1 = __builtin_fmaf(0.25f, v3, o1);
        } else {
            e0 = e0 + 0.0625f * v0; e0 = e0 + 0.375f * v1; e0 = e0 + 0.0625f * v2;
            o0 = o0 + 0.25f * v1; o0 = o0 + 0.25f * v2;
            e1 = e1 + 0.0625f * v1; e1 = e1 + 0.375f * v2; e1 = e1 + 0.0625f * v3;
            o1 = o1 + 0.25f * v2; o1 = o1 + 0.25f * v3;
        }
        h[0] = e0; h[1] = o0; h[2] = e1; h[3] = o1;
    };
    float out[4];
    if ((y & 1) == 0) {
        float h0[4], h1[4], h2[4];
        hrow(clamp_abs(cy - 1, rows), h0); hrow(clamp_abs(cy, rows), h1); hrow(clamp_abs(cy + 1, rows), h2);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float sum = 0.0f;
            if (CONTRACT) { sum = __builtin_fmaf(0.0625f, h0[i], sum); sum = __builtin_fmaf(0.375f, h1[i], sum); sum = __builtin_fmaf(0.0625f, h2[i], sum); }
            else { sum = sum + 0.0625f * h0[i]; sum = sum + 0.375f * h1[i]; sum = sum + 0.0625f * h2[i]; }
            out[i] = 4.0f * sum;
        }
        if (coarse_out) {               // the coarse level's caller-visible image on the side: fine pixels (x0, y) and (x0 + 2, y) store coarse (c0, cy), (c0 + 1, cy)
            const float *s = (const float *)((const char *)src + (size_t)cy * sp);
            float *q = (float *)((char *)coarse_out + (size_t)cy * cp);
            q[c0] = s[c0]; q[c0 + 1] = s[c0 + 1];
        }
    } else {
        float h1[4], h2[4];
        hrow(clamp_abs(cy, rows), h1); hrow(clamp_abs(cy + 1, rows), h2);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float sum = 0.0f;
            if (CONTRACT) { sum = __builtin_fmaf(0.25f, h1[i], sum); sum = __builtin_fmaf(0.25f, h2[i], sum); }
            else { sum = sum + 0.25f * h1[i]; sum = sum + 0.25f * h2[i]; }
            out[i] = 4.0f * sum;
        }
    }
    if (mask) {                         // GPUConvertToFloat fused in: Dirichlet pixels take their label (src/GPUImageProcessing.cu:19)
        const uint32_t m4 = *(const uint32_t *)(mask + (size_t)y * mp + x0);
        if (m4 & 0x80808080u) {         // (some byte >= 128: look closer; 255 is the only value that counts)
            const uint32_t *e3 = (const uint32_t *)(edited + (size_t)y * ep + 3 * (size_t)x0);
            const uint32_t w0 = e3[0], w1 = e3[1], w2 = e3[2];
            const uint32_t lab[4] = {w0 & 255, w0 >> 24, (w1 >> 16) & 255, (w2 >> 8) & 255};      // channel 0 of pixels x0 .. x0 + 3
#pragma unroll
            for (int i = 0; i < 4; i++) if (((m4 >> (8 * i)) & 255) == 255) out[i] = (float)lab[i];
        }
    }
    *(float4 *)((char *)dst + (size_t)y * dp + 4 * (size_t)x0) = make_float4(out[0], out[1], out[2], out[3]);
}

// GpuMat::convertTo(CV_8UC1): saturate(round-half-even) -- src/main.cpp:290
__global__ __launch_bounds__(256) void k_depth_to_u8(const float *__restrict__ src, size_t sp, uint8_t *__restrict__ dst, size_t dp, int rows, int cols) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + wave_id();
    if (x >= cols || y >= rows) return;
    const float r = __builtin_rintf(((const float *)((const char *)src + (size_t)y * sp))[x]);
    dst[(size_t)y * dp + x] = !(r >= 0.0f) ? 0 : (r >= 255.0f ? 255 : (uint8_t)(int)r);
}

// Annotation decode rule of main.cpp:160-168: edited := image; where annotation != 32: edited B=G=R := label, scribble := 255;
// elsewhere scribble keeps the annotation value (only == 255 is ever tested downstream).
__global__ __launch_bounds__(256) void k_decode_annotation(const uint8_t *__restrict__ bgr, size_t bp, const uint8_t *__restrict__ ann, size_t ap,
                                                           uint8_t *__restrict__ edited, size_t ep, uint8_t *__restrict__ scribble, size_t sp,
                                                           int rows, int cols) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + wave_id();
    if (x >= cols || y >= rows) return;
    const uint8_t a = ann[(size_t)y * ap + x];
    const uint8_t *p = bgr + (size_t)y * bp + 3 * x;
    uint8_t *e = edited + (size_t)y * ep + 3 * x;
    if (a != 32) { e[0] = a; e[1] = a; e[2] = a; scribble[(size_t)y * sp + x] = 255; }
    else { e[0] = p[0]; e[1] = p[1]; e[2] = p[2]; scribble[(size_t)y * sp + x] = a; }
}

__global__ __launch_bounds__(256) void k_fill_f32(float *__restrict__ dst, size_t dp, int rows, int cols, float v) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + wave_id();
    if (x >= cols || y >= rows) return;
    ((float *)((char *)dst + (size_t)y * dp))[x] = v;
}

static inline dim3 grid64x4(int rows, int cols, int images = 1) { return dim3((cols + 63) / 64, (rows + 3) / 4, images); }

int launch_bgr2gray(rtdd_ctx *ctx, const uint8_t *bgr, size_t bp, uint8_t *gray, size_t gp, int rows, int cols) {
    hipLaunchKernelGGL(k_bgr2gray, grid64x4(rows, cols), dim3(256), 0, ctx->stream, bgr, bp, gray, gp, rows, cols);
    RTDD_LAUNCH_CHECK(ctx, "k_bgr2gray");
    return RTDD_OK;
}
int launch_pyrdown_u8(rtdd_ctx *ctx, const uint8_t *src, size_t sp, int rows, int cols, uint8_t *dst, size_t dp) {
    const int drows = (rows + 1) / 2, dcols = (cols + 1) / 2;
    hipLaunchKernelGGL(k_pyrdown_u8, grid64x4(drows, dcols), dim3(256), 0, ctx->stream, src, sp, rows, cols, dst, dp, drows, dcols);
    RTDD_LAUNCH_CHECK(ctx, "k_pyrdown_u8");
    return RTDD_OK;
}
int launch_pyrup_inject(rtdd_ctx *ctx, const float *src, size_t sp, int rows, int cols, float *dst, size_t dp, int drows, int dcols,
                        const uint8_t *edited, size_t ep, const uint8_t *mask, size_t mp, float *coarse_out, size_t cp, int guard_seq, const PyrupBatch *pb) {
    static const PyrupBatch one;
    const PyrupBatch &Z = pb ? *pb : one;
    int *sw = guard_seq != 0 ? ctx->sync_words : nullptr;
    if (guard_seq != 0) note_publisher(ctx, guard_seq);
    const int seq = guard_seq;
    if (coarse_out && (drows < 2 * rows || dcols < 2 * cols)) return fail(ctx, RTDD_ERR_INVALID, "pyrUp: the fine level must be at least twice the coarse one");
    const bool doubling = drows == 2 * rows && dcols == 2 * cols && dcols % 4 == 0 && rows >= 2 && cols >= 2;
    const bool aligned = (uintptr_t)dst % 16 == 0 && dp % 16 == 0 && Z.dst % 16 == 0 && (!mask || ((uintptr_t)mask % 4 == 0 && mp % 4 == 0 && (uintptr_t)edited % 4 == 0 && ep % 4 == 0 && Z.mask % 4 == 0 && Z.edited % 4 == 0));
    if (doubling && aligned) {          // four pixels per thread
        if (ctx->opt.fp_contract) hipLaunchKernelGGL(k_pyrup_inject4<true>, grid64x4(drows, dcols / 4, Z.n), dim3(256), 0, ctx->stream, src, sp, rows, cols, dst, dp, drows, dcols, edited, ep, mask, mp, coarse_out, cp, sw, seq, Z.src, Z.dst, Z.edited, Z.mask, Z.coarse);
        else hipLaunchKernelGGL(k_pyrup_inject4<false>, grid64x4(drows, dcols / 4, Z.n), dim3(256), 0, ctx->stream, src, sp, rows, cols, dst, dp, drows, dcols, edited, ep, mask, mp, coarse_out, cp, sw, seq, Z.src, Z.dst, Z.edited, Z.mask, Z.coarse);
        RTDD_LAUNCH_CHECK(ctx, "k_pyrup_inject4");
        return RTDD_OK;
    }
    if (ctx->opt.fp_contract) hipLaunchKernelGGL(k_pyrup_inject<true>, grid64x4(drows, dcols, Z.n), dim3(256), 0, ctx->stream, src, sp, rows, cols, dst, dp, drows, dcols, edited, ep, mask, mp, coarse_out, cp, sw, seq, Z.src, Z.dst, Z.edited, Z.mask, Z.coarse);
    else hipLaunchKernelGGL(k_pyrup_inject<false>, grid64x4(drows, dcols, Z.n), dim3(256), 0, ctx->stream, src, sp, rows, cols, dst, dp, drows, dcols, edited, ep, mask, mp, coarse_out, cp, sw, seq, Z.src, Z.dst, Z.edited, Z.mask, Z.coarse);
    RTDD_LAUNCH_CHECK(ctx, "k_pyrup_inject");
    return RTDD_OK;
}
int launch_depth_to_u8(rtdd_ctx *ctx, const float *src, size_t sp, uint8_t *dst, size_t dp, int rows, int cols) {
    hipLaunchKernelGGL(k_depth_to_u8, grid64x4(rows, cols), dim3(256), 0, ctx->stream, src, sp, dst, dp, rows, cols);
    RTDD_LAUNCH_CHECK(ctx, "k_depth_to_u8");
    return RTDD_OK;
}
int launch_decode_annotation(rtdd_ctx *ctx, const uint8_t *bgr, size_t bp, const uint8_t *ann, size_t ap, uint8_t *edited, size_t ep,
                             uint8_t *scribble, size_t sp, int rows, int cols) {
    hipLaunchKernelGGL(k_decode_annotation, grid64x4(rows, cols), dim3(256), 0, ctx->stream, bgr, bp, ann, ap, edited, ep, scribble, sp, rows, cols);
    RTDD_LAUNCH_CHECK(ctx, "k_decode_annotation");
    return RTDD_OK;
}
int launch_fill_f32(rtdd_ctx *ctx, float *dst, size_t dp, int rows, int cols, float v) {
    hipLaunchKernelGGL(k_fill_f32, grid64x4(rows, cols), dim3(256), 0, ctx->stream, dst, dp, rows, cols, v);
    RTDD_LAUNCH_CHECK(ctx, "k_fill_f32");
    return RTDD_OK;
}

}  // namespace rtdd
