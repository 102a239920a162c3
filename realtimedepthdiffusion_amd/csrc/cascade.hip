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

namespace rtdd {

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

// cv::cvtColor(BGR2GRAY), 8u fixed point -- src/main.cpp:111,138
__global__ __launch_bounds__(256) void k_bgr2gray(const uint8_t *__restrict__ bgr, size_t bp, uint8_t *__restrict__ gray, size_t gp, int rows, int cols) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    const uint8_t *p = bgr + (size_t)y * bp + 3 * x;
    gray[(size_t)y * gp + x] = (uint8_t)((p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + (1 << 13)) >> 14);
}

// cv::pyrDown (8u): 5x5 [1 4 6 4 1]/16 separable, reflect-101, (s+128)>>8 -- src/main.cpp:112,144,245
__global__ __launch_bounds__(256) void k_pyrdown_u8(const uint8_t *__restrict__ src, size_t sp, int rows, int cols,
                                                    uint8_t *__restrict__ dst, size_t dp, int drows, int dcols) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
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

// cv::pyrUp (32f) to an explicit destination size -- src/main.cpp:273,277 -- with the Dirichlet
// re-injection of GPUConvertToFloat (src/main.cpp:281-283, src/GPUImageProcessing.cu:19) fused in.
// Accumulation order: rows outer, columns inner, ascending (the same as the CPU restatement it is tested against).
__global__ __launch_bounds__(256) void k_pyrup_inject(const float *__restrict__ src, size_t sp, int rows, int cols,
                                                      float *__restrict__ dst, size_t dp, int drows, int dcols,
                                                      const uint8_t *__restrict__ edited, size_t ep,
                                                      const uint8_t *__restrict__ mask, size_t mp) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dcols || y >= drows) return;
    float *out = (float *)((char *)dst + (size_t)y * dp) + x;
    if (mask && mask[(size_t)y * mp + x] == 255) { *out = (float)edited[(size_t)y * ep + 3 * x]; return; }
    int cy[3], cxx[3]; float wy[3], wx[3]; int ny, nx;
    if ((y & 1) == 0) { ny = 3; cy[0] = y / 2 - 1; cy[1] = y / 2; cy[2] = y / 2 + 1; wy[0] = 0.125f; wy[1] = 0.75f; wy[2] = 0.125f; }
    else { ny = 2; cy[0] = y / 2; cy[1] = y / 2 + 1; cy[2] = 0; wy[0] = 0.5f; wy[1] = 0.5f; wy[2] = 0.0f; }
    if ((x & 1) == 0) { nx = 3; cxx[0] = x / 2 - 1; cxx[1] = x / 2; cxx[2] = x / 2 + 1; wx[0] = 0.125f; wx[1] = 0.75f; wx[2] = 0.125f; }
    else { nx = 2; cxx[0] = x / 2; cxx[1] = x / 2 + 1; cxx[2] = 0; wx[0] = 0.5f; wx[1] = 0.5f; wx[2] = 0.0f; }
    float acc = 0.0f;
    for (int j = 0; j < ny; j++) {
        const float *srow = (const float *)((const char *)src + (size_t)reflect101(cy[j], rows) * sp);
        float h = 0.0f;
        for (int i = 0; i < nx; i++) h = h + wx[i] * srow[reflect101(cxx[i], cols)];
        acc = acc + wy[j] * h;
    }
    *out = acc;
}

// GpuMat::convertTo(CV_8UC1): saturate(round-half-even) -- src/main.cpp:290
__global__ __launch_bounds__(256) void k_depth_to_u8(const float *__restrict__ src, size_t sp, uint8_t *__restrict__ dst, size_t dp, int rows, int cols) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    const float r = __builtin_rintf(((const float *)((const char *)src + (size_t)y * sp))[x]);
    dst[(size_t)y * dp + x] = !(r >= 0.0f) ? 0 : (r >= 255.0f ? 255 : (uint8_t)(int)r);
}

// Annotation decode rule of main.cpp:160-168: edited := image; where annotation != 32: edited B=G=R := label, scribble := 255;
// elsewhere scribble keeps the annotation value (only == 255 is ever tested downstream).
__global__ __launch_bounds__(256) void k_decode_annotation(const uint8_t *__restrict__ bgr, size_t bp, const uint8_t *__restrict__ ann, size_t ap,
                                                           uint8_t *__restrict__ edited, size_t ep, uint8_t *__restrict__ scribble, size_t sp,
                                                           int rows, int cols) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    const uint8_t a = ann[(size_t)y * ap + x];
    const uint8_t *p = bgr + (size_t)y * bp + 3 * x;
    uint8_t *e = edited + (size_t)y * ep + 3 * x;
    if (a != 32) { e[0] = a; e[1] = a; e[2] = a; scribble[(size_t)y * sp + x] = 255; }
    else { e[0] = p[0]; e[1] = p[1]; e[2] = p[2]; scribble[(size_t)y * sp + x] = a; }
}

__global__ __launch_bounds__(256) void k_fill_f32(float *__restrict__ dst, size_t dp, int rows, int cols, float v) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    ((float *)((char *)dst + (size_t)y * dp))[x] = v;
}

static inline dim3 grid64x4(int rows, int cols) { return dim3((cols + 63) / 64, (rows + 3) / 4); }

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
                        const uint8_t *edited, size_t ep, const uint8_t *mask, size_t mp) {
    hipLaunchKernelGGL(k_pyrup_inject, grid64x4(drows, dcols), dim3(256), 0, ctx->stream, src, sp, rows, cols, dst, dp, drows, dcols, edited, ep, mask, mp);
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
