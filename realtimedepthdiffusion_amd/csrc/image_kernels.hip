// image_kernels.hip -- annotation-side passes: Dirichlet injection, 2x annotation downsample,
// square brush.  All three are tiny, HBM-latency-bound byte kernels; one wave covers 64
// consecutive pixels of a row so mask/depth accesses coalesce.
#include "rtdd_internal.hpp"

namespace rtdd {

// convert (K5) -- /root/reference/src/GPUImageProcessing.cu:8-21
__global__ __launch_bounds__(256) void k_convert(const uint8_t *__restrict__ src, size_t srcPitch, float *__restrict__ dst, size_t dstPitch,
                                                 const uint8_t *__restrict__ mask, size_t maskPitch, int rows, int cols, size_t zSrc, size_t zDst, size_t zMask) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + wave_id();
    if (x >= cols || y >= rows) return;
    RTDD_Z(src, zSrc); RTDD_Z(dst, zDst); RTDD_Z(mask, zMask);
    if (mask[(size_t)y * maskPitch + x] == 255)
        ((float *)((char *)dst + (size_t)y * dstPitch))[x] = (float)src[(size_t)y * srcPitch + 3 * x];
}

// pyrDown (K6) -- src/GPUImageProcessing.cu:23-49.  Scan order py outer, px inner, later hits
// overwrite earlier ones; nothing is ever cleared; channels 1,2 of the coarse image untouched.
__global__ __launch_bounds__(256) void k_pyrdown_annotation(const uint8_t *__restrict__ ps, size_t psp, const uint8_t *__restrict__ pe, size_t pep,
                                                            int prows, int pcols, uint8_t *__restrict__ cs, size_t csp,
                                                            uint8_t *__restrict__ ce, size_t cep, int crows, int ccols, size_t zPs, size_t zPe, size_t zCs, size_t zCe) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + wave_id();
    if (x >= ccols || y >= crows) return;
    RTDD_Z(ps, zPs); RTDD_Z(pe, zPe); RTDD_Z(cs, zCs); RTDD_Z(ce, zCe);
    int hit = -1;
#pragma unroll
    for (int j = -1; j <= 0; j++)
#pragma unroll
        for (int i = -1; i <= 0; i++) {
            const int px = 2 * x + i, py = 2 * y + j;
            if (px >= 0 && py >= 0 && px < pcols && py < prows && ps[(size_t)py * psp + px] == 255)
                hit = pe[(size_t)py * pep + 3 * px];
        }
    if (hit >= 0) {
        cs[(size_t)y * csp + x] = 255;
        ce[(size_t)y * cep + 3 * x] = (uint8_t)hit;
    }
}

// paintImage (K7) -- src/GPUImageProcessing.cu:51-70.  Launched over the brush's bounding box only
// (the reference launches the whole image and discards all but the brush).
__global__ __launch_bounds__(256) void k_paint(int x0, int y0, int x1, int y1, int color, uint8_t *__restrict__ edited, size_t editedPitch,
                                               uint8_t *__restrict__ scribble, size_t scribblePitch) {
    const int x = x0 + blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = y0 + blockIdx.y * 4 + wave_id();
    if (x > x1 || y > y1) return;
    uint8_t *e = edited + (size_t)y * editedPitch + 3 * x;
    e[0] = (uint8_t)color; e[1] = (uint8_t)color; e[2] = (uint8_t)color;
    scribble[(size_t)y * scribblePitch + x] = 255;
}

static inline dim3 grid64x4(int rows, int cols, int images = 1) { return dim3((cols + 63) / 64, (rows + 3) / 4, images); }

int launch_convert(rtdd_ctx *ctx, const uint8_t *src, size_t srcPitch, float *dst, size_t dstPitch,
                   const uint8_t *mask, size_t maskPitch, int rows, int cols, int images, size_t zSrc, size_t zDst, size_t zMask) {
    hipLaunchKernelGGL(k_convert, grid64x4(rows, cols, images), dim3(256), 0, ctx->stream, src, srcPitch, dst, dstPitch, mask, maskPitch, rows, cols, zSrc, zDst, zMask);
    RTDD_LAUNCH_CHECK(ctx, "k_convert");
    return RTDD_OK;
}

int launch_pyrdown_annotation(rtdd_ctx *ctx, const uint8_t *ps, size_t psp, const uint8_t *pe, size_t pep, int prows, int pcols,
                              uint8_t *cs, size_t csp, uint8_t *ce, size_t cep, int crows, int ccols, int images, size_t zPs, size_t zPe, size_t zCs, size_t zCe) {
    hipLaunchKernelGGL(k_pyrdown_annotation, grid64x4(crows, ccols, images), dim3(256), 0, ctx->stream, ps, psp, pe, pep, prows, pcols, cs, csp, ce, cep, crows, ccols, zPs, zPe, zCs, zCe);
    RTDD_LAUNCH_CHECK(ctx, "k_pyrdown_annotation");
    return RTDD_OK;
}

int launch_paint(rtdd_ctx *ctx, int x, int y, int color, int radius, uint8_t *edited, size_t editedPitch,
                 uint8_t *scribble, size_t scribblePitch, int rows, int cols) {
    const int h = radius / 2;                          // C integer division, as the reference (:58-59)
    const int x0 = x - h < 0 ? 0 : x - h, y0 = y - h < 0 ? 0 : y - h;
    const int x1 = x + h > cols - 1 ? cols - 1 : x + h, y1 = y + h > rows - 1 ? rows - 1 : y + h;
    if (x1 < x0 || y1 < y0) return RTDD_OK;            // brush entirely outside the image (or negative radius)
    hipLaunchKernelGGL(k_paint, grid64x4(y1 - y0 + 1, x1 - x0 + 1), dim3(256), 0, ctx->stream, x0, y0, x1, y1, color, edited, editedPitch, scribble, scribblePitch);
    RTDD_LAUNCH_CHECK(ctx, "k_paint");
    return RTDD_OK;
}

}  // namespace rtdd
