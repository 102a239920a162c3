/*
 * rtdd_cascade_oracle.c -- CPU restatement of the THIRD-PARTY (OpenCV 4.5.1) image ops that
 * main.cpp interleaves with the GPU* calls: BGR->gray, u8 Gaussian pyrDown, f32 pyrUp, f32->u8.
 *
 * TEST INFRASTRUCTURE ONLY (see rtdd_oracle.c).  PARITY UNPINNED: OpenCV is not vendored in
 * /root/reference (README.md:11 names 4.5.1 in prose only) and is not installed here, so these
 * follow OpenCV's PUBLISHED formulas, not its binaries:
 *   cvtColor BGR2GRAY (8u): Y = (B*1868 + G*9617 + R*4899 + 2^13) >> 14      (src/main.cpp:111,138)
 *   pyrDown (8u):  5x5 separable [1 4 6 4 1]/16, BORDER_REFLECT_101, (s+128)>>8,
 *                  dst size ((w+1)/2,(h+1)/2)                                  (src/main.cpp:112,144,245)
 *   pyrUp (32f):   zero-insert x2, 5x5 separable [1 4 6 4 1]/8 per axis, mirror at the top/left and
 *                  replicate at the bottom/right; cv::cuda::pyrUp for exact doubling, else cv::pyrUp
 *                  with the explicit size -- see orc_pyrup_f32                  (src/main.cpp:273,277)
 *   convertTo 8U:  saturate(round-half-even(v))                                (src/main.cpp:290)
 * Parity claims for the product start at the GPU* function boundary; these exist so the
 * cascade harness and its tests have a definition to agree with.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define ORC_API __attribute__((visibility("default")))

static inline int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) { if (i < 0) i = -i; else i = 2 * n - 2 - i; }
    return i;
}

ORC_API void orc_bgr2gray(const uint8_t *bgr, size_t bgrPitch, uint8_t *gray, size_t grayPitch, int rows, int cols) {
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            const uint8_t *p = bgr + (size_t)y * bgrPitch + 3 * x;
            gray[(size_t)y * grayPitch + x] = (uint8_t)((p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + (1 << 13)) >> 14);
        }
}

/* dst is ((cols+1)/2) x ((rows+1)/2) */
ORC_API void orc_pyrdown_u8(const uint8_t *src, size_t srcPitch, int rows, int cols, uint8_t *dst, size_t dstPitch) {
    int drows = (rows + 1) / 2, dcols = (cols + 1) / 2;
    static const int k[5] = {1, 4, 6, 4, 1};
    for (int y = 0; y < drows; y++)
        for (int x = 0; x < dcols; x++) {
            int s = 0;
            for (int j = -2; j <= 2; j++) {
                int sy = reflect101(2 * y + j, rows);
                int h = 0;
                for (int i = -2; i <= 2; i++) h += k[i + 2] * src[(size_t)sy * srcPitch + reflect101(2 * x + i, cols)];
                s += k[j + 2] * h;
            }
            dst[(size_t)y * dstPitch + x] = (uint8_t)((s + 128) >> 8);
        }
}

/* f32 pyrUp as src/main.cpp:272-279 calls it -- TWO different OpenCV routines, chosen by the sizes:
 *   dst exactly twice src in both dimensions  ->  cv::cuda::pyrUp (main.cpp:273; opencv_contrib cudawarping pyr_up.cu):
 *       source index clamp min(n-1, |i|) (mirror at the top/left, REPLICATE at the bottom/right); horizontal pass on the
 *       source rows with weights 1/16, 3/8, 1/16 (even columns) or 1/4, 1/4 (odd columns), accumulated left to right as
 *       sum = sum + w * s -- which nvcc's default -fmad=true contracts to fma(w, s, sum): the `contract` flag, as for the
 *       solver --, the same vertically, result x 4;
 *   otherwise (an odd fine size)  ->  cv::pyrUp on the host with the explicit size (main.cpp:277; imgproc pyramids.cpp,
 *       pyrUp_): per source row  even: s[c-1] + s[c]*6 + s[c+1], odd: (s[c] + s[c+1])*4, with the border forms
 *       x = 0: s[0]*6 + s[1]*2 and x = n-1: s[n-2] + s[n-1]*7 / s[n-1]*8 (so again mirror left, replicate right) and a
 *       single source column giving s[0]*8 twice; rows above the first mirror (row 1), rows below the last replicate;
 *       even rows (r0 + r1*6 + r2)/64, odd rows ((r1 + r2)*4)/64; when dst is 2n+1 wide the extra column repeats column
 *       2n-1, when it is 2H+1 tall the extra row repeats row 2H-2 (sic).
 * Both follow my reading of OpenCV 4.5.1's sources, which are NOT under /root/reference (README.md:11 names the version
 * in prose only) and not installed here: UNPINNED.  Known unknowables: whether the host build vectorises the vertical
 * pass (its SIMD form adds r0 + (r1*6 + r2)) and with which FMA dispatch.  Away from the borders and without contraction
 * the two branches agree bit for bit (all scale factors are powers of two). */
static inline int clamp_abs(int i, int n) { if (i < 0) i = -i; return i < n - 1 ? i : n - 1; }

static float pyrup_cuda_h(const float *s, int n, int k, int contract) {          /* horizontal pass of cv::cuda::pyrUp */
    const int c = k >> 1;
    float sum = 0.0f;
    if ((k & 1) == 0) {
        const float a = s[clamp_abs(c - 1, n)], b = s[clamp_abs(c, n)], d = s[clamp_abs(c + 1, n)];
        if (contract) { sum = fmaf(0.0625f, a, sum); sum = fmaf(0.375f, b, sum); sum = fmaf(0.0625f, d, sum); }
        else { sum = sum + 0.0625f * a; sum = sum + 0.375f * b; sum = sum + 0.0625f * d; }
    } else {
        const float a = s[clamp_abs(c, n)], b = s[clamp_abs(c + 1, n)];
        if (contract) { sum = fmaf(0.25f, a, sum); sum = fmaf(0.25f, b, sum); }
        else { sum = sum + 0.25f * a; sum = sum + 0.25f * b; }
    }
    return sum;
}

static float pyrup_host_h(const float *s, int n, int k) {                         /* one entry of pyrUp_'s row buffer */
    if (n == 1) return s[0] * 8;
    if (k >= 2 * n) k = 2 * n - 1;                                                 /* the extra column of an odd width */
    const int c = k >> 1;
    if ((k & 1) == 0) {
        if (c == 0) return s[0] * 6 + s[1] * 2;
        if (c == n - 1) return s[n - 2] + s[n - 1] * 7;
        return s[c - 1] + s[c] * 6 + s[c + 1];
    }
    if (c == n - 1) return s[n - 1] * 8;
    return (s[c] + s[c + 1]) * 4;
}

ORC_API void orc_pyrup_f32(const float *src, size_t srcPitch, int rows, int cols,
                           float *dst, size_t dstPitch, int drows, int dcols, int contract) {
#define SROW(r) ((const float *)((const char *)src + (size_t)(r) * srcPitch))
    const int exact = drows == 2 * rows && dcols == 2 * cols;
    for (int y = 0; y < drows; y++)
        for (int x = 0; x < dcols; x++) {
            float out;
            if (exact) {                                                           /* cv::cuda::pyrUp */
                const int cy = y >> 1;
                float sum = 0.0f;
                if ((y & 1) == 0) {
                    const float h0 = pyrup_cuda_h(SROW(clamp_abs(cy - 1, rows)), cols, x, contract), h1 = pyrup_cuda_h(SROW(clamp_abs(cy, rows)), cols, x, contract),
                                h2 = pyrup_cuda_h(SROW(clamp_abs(cy + 1, rows)), cols, x, contract);
                    if (contract) { sum = fmaf(0.0625f, h0, sum); sum = fmaf(0.375f, h1, sum); sum = fmaf(0.0625f, h2, sum); }
                    else { sum = sum + 0.0625f * h0; sum = sum + 0.375f * h1; sum = sum + 0.0625f * h2; }
                } else {
                    const float h1 = pyrup_cuda_h(SROW(clamp_abs(cy, rows)), cols, x, contract), h2 = pyrup_cuda_h(SROW(clamp_abs(cy + 1, rows)), cols, x, contract);
                    if (contract) { sum = fmaf(0.25f, h1, sum); sum = fmaf(0.25f, h2, sum); }
                    else { sum = sum + 0.25f * h1; sum = sum + 0.25f * h2; }
                }
                out = 4.0f * sum;
            } else {                                                               /* cv::pyrUp, explicit size */
                const int yy = y >= 2 * rows ? 2 * rows - 2 : y;                   /* the extra row of an odd height */
                const int cy = yy >> 1;
                const int r0 = cy - 1 < 0 ? (rows > 1 ? 1 : 0) : cy - 1, r2 = cy + 1 >= rows ? rows - 1 : cy + 1;
                const float v1 = pyrup_host_h(SROW(cy), cols, x), v2 = pyrup_host_h(SROW(r2), cols, x);
                if ((yy & 1) == 0) { const float v0 = pyrup_host_h(SROW(r0), cols, x); out = (v0 + v1 * 6 + v2) * 0.015625f; }
                else out = ((v1 + v2) * 4) * 0.015625f;
            }
            ((float *)((char *)dst + (size_t)y * dstPitch))[x] = out;
        }
#undef SROW
}

ORC_API void orc_depth_to_u8(const float *src, size_t srcPitch, uint8_t *dst, size_t dstPitch, int rows, int cols) {
    for (int y = 0; y < rows; y++) {
        const float *srow = (const float *)((const char *)src + (size_t)y * srcPitch);
        for (int x = 0; x < cols; x++) {
            float r = nearbyintf(srow[x]);           /* round-half-even in the default rounding mode */
            dst[(size_t)y * dstPitch + x] = !(r >= 0.0f) ? 0 : (r >= 255.0f ? 255 : (uint8_t)r);
        }
    }
}
