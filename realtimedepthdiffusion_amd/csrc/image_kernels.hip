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

// The whole annotation pyramid of an estimate in ONE launch (src/main.cpp:249-259: P - 1 GPUPyrDownAnnotation calls, then
// GPUConvertToFloat on the coarsest level).  A coarse pixel x looks at the fine pixels 2x - 1 and 2x (above), so the level-l pixels
// [a 2^l - (2^l - 1), a 2^l] and nothing else feed level-(l + k) pixel a ... : the footprints of the coarsest level's pixels tile
// EVERY level disjointly.  A workgroup therefore owns kApB x kApB pixels of the coarsest level and, level by level, all the pixels of the
// finer levels under them: it writes a level, __syncthreads(), and reads it back for the next one -- no other workgroup touches those
// pixels, nothing is cleared (the coarse images accumulate over the frames exactly as with one launch per level), the scan order and
// "last hit wins" are pyrDown's above.  Pixels of a fine level beyond the last coarsest pixel's footprint (the level sizes are floors)
// belong to the workgroups of the next tile row / column: the grid is one tile larger than the coarsest level needs.
// Five launches of ~7 us each (dependent, tiny) -> one: a live 1080p frame 1.18 -> 1.15 ms of GPU time (profiles/r05_live_timeline_*).
constexpr int kApMaxLevels = 12, kApB = 2;
struct AnnotationPyramid {
    int levels;                                   // P
    uint8_t *scribble[kApMaxLevels], *edited[kApMaxLevels];
    size_t sp[kApMaxLevels], ep[kApMaxLevels];    // pitches
    size_t zs[kApMaxLevels], ze[kApMaxLevels];    // byte strides between the images of a batch (blockIdx.z)
    int rows[kApMaxLevels], cols[kApMaxLevels];
    float *depth; size_t dp, zd;                  // the coarsest level's depth image (src/main.cpp:257-259)
};

__global__ __launch_bounds__(256) void k_annotation_pyramid(AnnotationPyramid A) {
    const int P = A.levels, top = P - 1;
    const size_t z = blockIdx.z;
    for (int l = 1; l <= top; l++) {
        const int f = 1 << (top - l);                                // level-l pixels per coarsest pixel, per direction
        // this workgroup's pixels of level l: [X0 f - (f - 1), (X0 + kApB - 1) f] x the same in y, clipped to the level
        const int xa = max((int)blockIdx.x * kApB * f - (f - 1), 0), xb = min(((int)blockIdx.x * kApB + kApB - 1) * f, A.cols[l] - 1);
        const int ya = max((int)blockIdx.y * kApB * f - (f - 1), 0), yb = min(((int)blockIdx.y * kApB + kApB - 1) * f, A.rows[l] - 1);
        const int w = xb - xa + 1, h = yb - ya + 1;
        if (w > 0 && h > 0) {
            const uint8_t *ps = A.scribble[l - 1] + z * A.zs[l - 1], *pe = A.edited[l - 1] + z * A.ze[l - 1];
            uint8_t *cs = A.scribble[l] + z * A.zs[l], *ce = A.edited[l] + z * A.ze[l];
            const int prows = A.rows[l - 1], pcols = A.cols[l - 1];
            for (int i = threadIdx.x; i < w * h; i += 256) {
                const int x = xa + i % w, y = ya + i / w;
                int hit = -1;
#pragma unroll
                for (int jj = -1; jj <= 0; jj++)
#pragma unroll
                    for (int ii = -1; ii <= 0; ii++) {
                        const int px = 2 * x + ii, py = 2 * y + jj;
                        if (px >= 0 && py >= 0 && px < pcols && py < prows && ps[(size_t)py * A.sp[l - 1] + px] == 255)
                            hit = pe[(size_t)py * A.ep[l - 1] + 3 * px];
                    }
                if (hit >= 0) {
                    cs[(size_t)y * A.sp[l] + x] = 255;
                    ce[(size_t)y * A.ep[l] + 3 * x] = (uint8_t)hit;
                }
            }
        }
        __syncthreads();                                             // the level is read back by this workgroup only
    }
    // convert (K5) on the coarsest level: the workgroup's kApB x kApB pixels
    const int x = (int)blockIdx.x * kApB + (int)(threadIdx.x % kApB), y = (int)blockIdx.y * kApB + (int)(threadIdx.x / kApB);
    if (threadIdx.x < kApB * kApB && A.depth && x < A.cols[top] && y < A.rows[top]) {
        const uint8_t *m = A.scribble[top] + z * A.zs[top], *e = A.edited[top] + z * A.ze[top];
        if (m[(size_t)y * A.sp[top] + x] == 255)
            ((float *)((char *)A.depth + z * A.zd + (size_t)y * A.dp))[x] = (float)e[(size_t)y * A.ep[top] + 3 * x];
    }
}

// The same pyramid with the chain in LDS (pyramids of up to six levels: every size up to 4K).  In footprint-local coordinates (origin =
// the unclipped first pixel of the workgroup's footprint on that level) level-l pixel (lx, ly) reads level-(l - 1) pixels
// (2 lx + {0, 1}, 2 ly + {0, 1}), and all a level hands to the next is "the edited value where the scribble flag is 255, else nothing":
// one short per pixel.  Every global load of the workgroup -- its level-0 footprint and, because the coarse images accumulate over the
// frames, the OLD state of every coarser level -- is issued before the first is used (ONE memory round trip; a first version that
// stored each level's map to LDS as it arrived paid one per level and measured no faster than the global chain: EXPERIMENTS.md), then
// the levels are walked in LDS and the pixels that were hit stored.
template <int TOP>
__global__ __launch_bounds__(256) void k_annotation_pyramid_lds(AnnotationPyramid A) {
    constexpr int kN0 = kApB << TOP, kE0 = (kN0 * kN0 + 255) / 256;  // level-0 footprint edge, its entries per thread
    constexpr int kTotal = (4 * kN0 * kN0 - kApB * kApB) / 3;       // sum over the levels of (kApB << (TOP - l))^2
    __shared__ short map[kTotal];
    const int tid = threadIdx.x;
    const size_t z = blockIdx.z;
    int flag[TOP + 1][kE0], val[TOP + 1][kE0];
#pragma unroll
    for (int l = 0; l <= TOP; l++) {
        const int f = 1 << (TOP - l), n = kApB * f;
        const int x0 = (int)blockIdx.x * kApB * f - (f - 1), y0 = (int)blockIdx.y * kApB * f - (f - 1);
        const uint8_t *ps = A.scribble[l] + z * A.zs[l], *pe = A.edited[l] + z * A.ze[l];
#pragma unroll
        for (int k = 0; k < kE0; k++) {
            if (k * 256 >= n * n) continue;                          // (compile time after unrolling: the coarser levels have fewer entries)
            const int i = tid + 256 * k, lx = i & (n - 1), ly = i / n, x = x0 + lx, y = y0 + ly;
            const bool in = i < n * n && x >= 0 && y >= 0 && x < A.cols[l] && y < A.rows[l];
            const int xc = in ? x : 0, yc = in ? y : 0;
            flag[l][k] = ps[(size_t)yc * A.sp[l] + xc]; val[l][k] = pe[(size_t)yc * A.ep[l] + 3 * xc];
        }
    }
    int off = 0;
#pragma unroll
    for (int l = 0; l <= TOP; l++) {
        const int f = 1 << (TOP - l), n = kApB * f;
        const int x0 = (int)blockIdx.x * kApB * f - (f - 1), y0 = (int)blockIdx.y * kApB * f - (f - 1);
#pragma unroll
        for (int k = 0; k < kE0; k++) {
            if (k * 256 >= n * n) continue;
            const int i = tid + 256 * k, lx = i & (n - 1), ly = i / n, x = x0 + lx, y = y0 + ly;
            const bool in = x >= 0 && y >= 0 && x < A.cols[l] && y < A.rows[l];
            if (i < n * n) map[off + i] = (short)((in && flag[l][k] == 255) ? val[l][k] : -1);
        }
        off += n * n;
    }
    __syncthreads();
    int poff = 0;
#pragma unroll
    for (int l = 1; l <= TOP; l++) {
        const int f = 1 << (TOP - l), n = kApB * f, pn = 2 * n, coff = poff + pn * pn;
        const int x0 = (int)blockIdx.x * kApB * f - (f - 1), y0 = (int)blockIdx.y * kApB * f - (f - 1);
        uint8_t *cs = A.scribble[l] + z * A.zs[l], *ce = A.edited[l] + z * A.ze[l];
#pragma unroll
        for (int k = 0; k < kE0; k++) {
            if (k * 256 >= n * n) continue;
            const int i = tid + 256 * k;
            if (i < n * n) {
                const int lx = i & (n - 1), ly = i / n, x = x0 + lx, y = y0 + ly;
                const short *q = map + poff + (2 * ly) * pn + 2 * lx;
                int hit = -1;                                        // scan order py outer, px inner, the last hit wins (k_pyrdown_annotation)
                if (q[0] >= 0) hit = q[0];
                if (q[1] >= 0) hit = q[1];
                if (q[pn] >= 0) hit = q[pn];
                if (q[pn + 1] >= 0) hit = q[pn + 1];
                if (hit >= 0 && x >= 0 && y >= 0 && x < A.cols[l] && y < A.rows[l]) {
                    map[coff + i] = (short)hit;
                    cs[(size_t)y * A.sp[l] + x] = 255;
                    ce[(size_t)y * A.ep[l] + 3 * x] = (uint8_t)hit;
                }
            }
        }
        __syncthreads();
        poff = coff;
    }
    // convert (K5) on the coarsest level: the workgroup's kApB x kApB pixels
    const int x = (int)blockIdx.x * kApB + (tid % kApB), y = (int)blockIdx.y * kApB + (tid / kApB);
    if (tid < kApB * kApB && A.depth && x < A.cols[TOP] && y < A.rows[TOP]) {
        const int v = map[poff + tid];
        if (v >= 0) ((float *)((char *)A.depth + z * A.zd + (size_t)y * A.dp))[x] = (float)v;
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

// rows of `width` bytes from one pitch to another (copy_h2d / copy_d2h, rtdd_live_submit: one side is a contiguous buffer whose row length is
// no multiple of four).  Four bytes per thread: one dword on whichever side allows it (gfx950 loads and stores dwords at any address:
// the dword goes to the side whose rows are NOT aligned only if that is the only way), bytes on the other; 8 MB in ~10 us.
typedef uint32_t __attribute__((aligned(1))) u32_any;
__global__ __launch_bounds__(256) void k_repitch(const uint8_t *__restrict__ src, size_t sp, uint8_t *__restrict__ dst, size_t dp, size_t width, int rows) {
    const size_t x = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const int y0 = blockIdx.y * 8;
    if (x >= width) return;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (y0 + i >= rows) break;
        const uint8_t *s = src + (size_t)(y0 + i) * sp + x;
        uint8_t *d = dst + (size_t)(y0 + i) * dp + x;
        if (x + 3 < width) *(u32_any *)d = *(const u32_any *)s;
        else for (size_t k = 0; x + k < width; k++) d[k] = s[k];
    }
}

int launch_repitch(rtdd_ctx *ctx, hipStream_t stream, const void *src, size_t srcPitch, void *dst, size_t dstPitch, size_t widthBytes, int rows) {
    if (rows <= 0 || widthBytes == 0) return RTDD_OK;
    hipLaunchKernelGGL(k_repitch, dim3((unsigned)((widthBytes + 1023) / 1024), (unsigned)((rows + 7) / 8)), dim3(256), 0, stream, (const uint8_t *)src, srcPitch, (uint8_t *)dst, dstPitch, widthBytes, rows);
    RTDD_LAUNCH_CHECK(ctx, "k_repitch");
    return RTDD_OK;
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

int launch_annotation_pyramid(rtdd_ctx *ctx, int levels, uint8_t *const *scribble, const size_t *sp, const size_t *zs, uint8_t *const *edited, const size_t *ep, const size_t *ze,
                              const int *rows, const int *cols, float *depth, size_t dp, size_t zd, int images) {
    if (levels < 1 || levels > kApMaxLevels) return fail(ctx, RTDD_ERR_INVALID, "annotation pyramid: too many levels");
    AnnotationPyramid A{};
    A.levels = levels;
    for (int l = 0; l < levels; l++) { A.scribble[l] = scribble[l]; A.edited[l] = edited[l]; A.sp[l] = sp[l]; A.ep[l] = ep[l]; A.zs[l] = zs[l]; A.ze[l] = ze[l]; A.rows[l] = rows[l]; A.cols[l] = cols[l]; }
    A.depth = depth; A.dp = dp; A.zd = zd;
    // tiles of kApB x kApB coarsest pixels, one tile more than the coarsest level needs in either direction: the finer levels' last
    // pixels (their sizes are floors) lie under coarsest pixels that do not exist
    const int top = levels - 1;
    int gx = 1, gy = 1;
    for (int l = 1; l <= top; l++) {                                 // the largest tile index any level's last pixel falls into
        const int f = 1 << (top - l);
        const int tx = (cols[l] - 1 + f - 1) / f / kApB + 1, ty = (rows[l] - 1 + f - 1) / f / kApB + 1;
        if (cols[l] > 0 && tx > gx) gx = tx;
        if (rows[l] > 0 && ty > gy) gy = ty;
    }
    if (top == 0) { gx = (cols[0] + kApB - 1) / kApB; gy = (rows[0] + kApB - 1) / kApB; }
    if (gx < 1 || gy < 1) return RTDD_OK;
    const dim3 grid(gx, gy, images);
    const int lds = ctx->opt.annotation_lds;
    if (lds && top == 1) hipLaunchKernelGGL(k_annotation_pyramid_lds<1>, grid, dim3(256), 0, ctx->stream, A);
    else if (lds && top == 2) hipLaunchKernelGGL(k_annotation_pyramid_lds<2>, grid, dim3(256), 0, ctx->stream, A);
    else if (lds && top == 3) hipLaunchKernelGGL(k_annotation_pyramid_lds<3>, grid, dim3(256), 0, ctx->stream, A);
    else if (lds && top == 4) hipLaunchKernelGGL(k_annotation_pyramid_lds<4>, grid, dim3(256), 0, ctx->stream, A);
    else if (lds && top == 5) hipLaunchKernelGGL(k_annotation_pyramid_lds<5>, grid, dim3(256), 0, ctx->stream, A);
    else hipLaunchKernelGGL(k_annotation_pyramid, grid, dim3(256), 0, ctx->stream, A);      // (deeper pyramids: the levels through global memory)
    RTDD_LAUNCH_CHECK(ctx, "k_annotation_pyramid");
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
