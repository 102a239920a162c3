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
 *   pyrUp (32f):   zero-insert x2, 5x5 separable [1 4 6 4 1]/8 per axis, reflect-101 on the
 *                  coarse grid, explicit dst size                              (src/main.cpp:273,277)
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

/* dst (drows x dcols) given explicitly, as main.cpp:277 does; coarse index of a fine sample
 * outside the coarse grid is reflected (101).  Accumulation order is fixed (rows outer,
 * columns inner, ascending) so the device kernel can repeat it op-for-op. */
ORC_API void orc_pyrup_f32(const float *src, size_t srcPitch, int rows, int cols,
                           float *dst, size_t dstPitch, int drows, int dcols) {
    for (int y = 0; y < drows; y++)
        for (int x = 0; x < dcols; x++) {
            /* taps of the zero-inserted signal: even fine index e=2c -> (c-1:1, c:6, c+1:1)/8; odd e=2c+1 -> (c:4, c+1:4)/8 */
            int cy[3], cx[3]; float wy[3], wx[3]; int ny, nx;
            if ((y & 1) == 0) { ny = 3; cy[0] = y / 2 - 1; cy[1] = y / 2; cy[2] = y / 2 + 1; wy[0] = 0.125f; wy[1] = 0.75f; wy[2] = 0.125f; }
            else { ny = 2; cy[0] = y / 2; cy[1] = y / 2 + 1; wy[0] = 0.5f; wy[1] = 0.5f; }
            if ((x & 1) == 0) { nx = 3; cx[0] = x / 2 - 1; cx[1] = x / 2; cx[2] = x / 2 + 1; wx[0] = 0.125f; wx[1] = 0.75f; wx[2] = 0.125f; }
            else { nx = 2; cx[0] = x / 2; cx[1] = x / 2 + 1; wx[0] = 0.5f; wx[1] = 0.5f; }
            float acc = 0.0f;
            for (int j = 0; j < ny; j++) {
                const float *srow = (const float *)((const char *)src + (size_t)reflect101(cy[j], rows) * srcPitch);
                float h = 0.0f;
                for (int i = 0; i < nx; i++) h = h + wx[i] * srow[reflect101(cx[i], cols)];
                acc = acc + wy[j] * h;
            }
            ((float *)((char *)dst + (size_t)y * dstPitch))[x] = acc;
        }
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
