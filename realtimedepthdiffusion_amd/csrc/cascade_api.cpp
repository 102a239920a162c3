// cascade_api.cpp -- pyramid state and the whole-estimate driver (include/rtdd.h, "whole-estimate
// driver").  Follows /root/reference/src/main.cpp:92-155 (setup) and :232-295 (estimate).
#include <cmath>
#include <cstring>
#include <new>

#include "rtdd_internal.hpp"
#include "persist_sync.hpp"

namespace rtdd {

struct Image {
    void *ptr = nullptr;
    size_t pitch = 0;
    int rows = 0, cols = 0, elem = 1;
    // a batched pyramid (rtdd_pyramid_create_batch) holds every image `images` times, copy b `stride` bytes behind copy b - 1
    size_t stride = 0;
    int images = 1;
    void *at(int b) const { return ptr ? (char *)ptr + (size_t)b * stride : nullptr; }
};

struct Pyramid {
    int levels = 0, rows = 0, cols = 0;
    Image original, depth_u8, artistic;
    std::vector<Image> gray, scribble, edited, depth;
    // The coarse annotation levels and the coarsest level's injection depend on the annotation only (src/main.cpp:249-259;
    // GPUPyrDownAnnotation only ever adds, SURVEY A.8, and the solver never moves a Dirichlet pixel): they are brought up to date
    // by the first estimate after the annotation changed, not by every estimate.
    bool annotation_dirty = true;         // (of ANY image of a batch: bringing an up-to-date image up to date again changes nothing)
    int images = 1, sel = 0;              // batch size; the image the single-image entry points address (rtdd_pyramid_select)
    std::vector<rtdd_solve_info> level_info;      // what the most recent estimate ran per level (rtdd_pyramid_level_info)
    std::vector<int> level_launch_images;
    struct Live *live = nullptr;          // rtdd_live_submit's second stream, staging images and events (created on first use)
};

// Live mode (src/main.cpp:232-295, the body of the while loop, one frame): upload of the scribble and edited images (:236-237), the
// estimate, download of the u8 map (:290-291).  Frames are pipelined two deep: the copies run on a stream of their own, frame N+1's
// upload while frame N computes, frame N's download while frame N+1 computes.  Slot k = frame number % 2.
struct Live {
    // uploads and downloads each on a stream of their own: frame N+1's upload must not queue behind frame N's download
    hipStream_t up = nullptr, copy = nullptr;
    Image scribble_stage[2], edited_stage[2], u8_stage[2];
    hipEvent_t h2d_done[2] = {nullptr, nullptr}, est_done[2] = {nullptr, nullptr}, d2h_done[2] = {nullptr, nullptr};
    int *status_host = nullptr;           // page-locked, 2 x 8 ints: the kernels' control words as they were behind each frame's estimate
    // a frame's sticky depth effect (rtdd_live_submit_ex) renders here; allocated with the first such frame
    Image art_stage[2];
    struct Frame {
        uint8_t *host = nullptr; size_t pitch = 0;
        bool in_flight = false, direct = false;
        unsigned long long op_id = 0;
        int effect = 0;                   // RTDD_EFFECT_*: the frame carries an artistic image too
        uint8_t *art_host = nullptr; size_t art_pitch = 0;
        // its download was queued at submit on the compute stream (no other frame in flight); else rtdd_live_wait issues it
        bool art_queued = false;
    } frame[2];
    unsigned long long submitted = 0, waited = 0;
    // Round 5: no device-to-device staging copies.  A frame's annotation is uploaded into the staging pair that is NOT the pyramid's
    // current level-0 annotation, and the pyramid's level-0 scribble / edited images then simply BECOME that pair (pointer swap); the
    // frame's u8 map is written by the estimate's copy-back into RTDD_IMG_DEPTH_U8 AND into the frame's slot (k_finish: two targets).
    // The pyramid's own annotation buffers are kept here and put back before the pyramid is freed.
    void *own_scribble = nullptr, *own_edited = nullptr, *own_artistic = nullptr;
    Bounce bounce_art;                      // (an artistic image on its way to a host image with an unaligned pitch, compute stream)
    // ([frame parity][scribble, edited]: uploads of images with an unaligned host pitch, rtdd_live_submit)
    Bounce bounce_up[2][2];
    // page-locked: a staged map on its way to a host image with an unaligned pitch (live_fetch)
    uint8_t *host_bounce = nullptr; size_t host_bounce_bytes = 0;
};

// The device's name for a page-locked host image (every byte of rows x cols at `pitch` inside one registered range), or nullptr.  Asked
// for every frame (two driver queries, ~2 us of host time): a buffer freed and allocated again at the same address may not be page-locked
// any more.
static uint8_t *live_device_view(const uint8_t *host, size_t pitch, int rows, int cols) {
    uint8_t *dev = nullptr;
    hipPointerAttribute_t a0, a1;
    const uint8_t *last = host + (size_t)(rows - 1) * pitch + (size_t)cols - 1;
    if (hipPointerGetAttributes(&a0, host) == hipSuccess && hipPointerGetAttributes(&a1, last) == hipSuccess && a0.type == hipMemoryTypeHost
        && a1.type == hipMemoryTypeHost &&
        a0.devicePointer && a1.devicePointer && (const uint8_t *)a1.devicePointer - (const uint8_t *)a0.devicePointer == last - host)
        dev = (uint8_t *)a0.devicePointer;
    else (void)hipGetLastError();                                   // (an ordinary host pointer: the query fails, nothing else has)
    return dev;
}

// hipMemcpy2DAsync between host and device takes ~9 us PER ROW when the host pitch is no multiple of four -- 8 ms for one plane of a
// 910-pixel-wide
// image (the dataset's Arara) where an aligned pitch takes 25 us (profiles/r05_copy2d_pitch.txt).  A host image with such a pitch whose
// rows are contiguous (what a continuous cv::Mat or a numpy array is) moves as ONE linear copy through a contiguous device buffer and a
// re-pitching kernel on the same stream; any other layout takes the runtime's copy as it is.
static bool wants_bounce(size_t hostPitch, size_t widthBytes, int rows) { return hostPitch % 4 != 0 && hostPitch == widthBytes
    && rows > 1; }
static int bounce_reserve(rtdd_ctx *ctx, Bounce &b, size_t bytes, hipStream_t stream) {
    if (b.bytes >= bytes) return RTDD_OK;
    if (b.ptr) { RTDD_HIP(ctx, hipStreamSynchronize(stream)); RTDD_HIP(ctx, hipFree(b.ptr)); b.ptr = nullptr; b.bytes = 0; }
    RTDD_HIP(ctx, hipMalloc(&b.ptr, bytes));
    b.bytes = bytes;
    return RTDD_OK;
}
int copy_h2d(rtdd_ctx *ctx, Bounce &b, void *dev, size_t devPitch, const void *host, size_t hostPitch, size_t widthBytes, int rows,
    hipStream_t stream) {
    if (!wants_bounce(hostPitch, widthBytes, rows)) { RTDD_HIP(ctx,
        hipMemcpy2DAsync(dev, devPitch, host, hostPitch, widthBytes, rows, hipMemcpyHostToDevice, stream)); return RTDD_OK; }
    const int rc = bounce_reserve(ctx, b, widthBytes * (size_t)rows, stream);
    if (rc != RTDD_OK) return rc;
    RTDD_HIP(ctx, hipMemcpyAsync(b.ptr, host, widthBytes * (size_t)rows, hipMemcpyHostToDevice, stream));
    return launch_repitch(ctx, stream, b.ptr, widthBytes, dev, devPitch, widthBytes, rows);
}
int copy_d2h(rtdd_ctx *ctx, Bounce &b, void *host, size_t hostPitch, const void *dev, size_t devPitch, size_t widthBytes, int rows,
    hipStream_t stream) {
    if (!wants_bounce(hostPitch, widthBytes, rows)) { RTDD_HIP(ctx,
        hipMemcpy2DAsync(host, hostPitch, dev, devPitch, widthBytes, rows, hipMemcpyDeviceToHost, stream)); return RTDD_OK; }
    int rc = bounce_reserve(ctx, b, widthBytes * (size_t)rows, stream);
    if (rc != RTDD_OK) return rc;
    if ((rc = launch_repitch(ctx, stream, dev, devPitch, b.ptr, widthBytes, widthBytes, rows)) != RTDD_OK) return rc;
    RTDD_HIP(ctx, hipMemcpyAsync(host, b.ptr, widthBytes * (size_t)rows, hipMemcpyDeviceToHost, stream));
    return RTDD_OK;
}

static bool inside(const Image &im, const void *p) {
    return im.ptr && (const char *)p >= (const char *)im.ptr && (const char *)p < (const char *)im.ptr + im.stride * (size_t)im.images;
}

// Called by the entry points that write an annotation image: is it one of this context's pyramid?  After an uploading live frame the
// pyramid's level-0 scribble / edited images ARE the uploaded staging pair (Live, below), so a pointer rtdd_pyramid_image handed out
// before that frame names a buffer no estimate reads any more: writing it through the library would lose the strokes silently --
// RTDD_ERR_STATE instead (ask rtdd_pyramid_image again).
static bool stale_live_pointer(const Pyramid *p, const void *q) {
    const Live *v = p->live;
    if (!v || !q) return false;
    auto in_image = [](const void *base, const Image &like, const void *r) {
        return base && (const char *)r >= (const char *)base
            && (const char *)r < (const char *)base + like.pitch * (size_t)(like.rows > 0 ? like.rows : 1);
    };
    const void *olds[3] = {v->own_scribble, v->scribble_stage[0].ptr, v->scribble_stage[1].ptr};
    const void *olde[3] = {v->own_edited, v->edited_stage[0].ptr, v->edited_stage[1].ptr};
    for (int i = 0; i < 3; i++)
        if ((olds[i] != p->scribble[0].ptr && in_image(olds[i], p->scribble[0], q))
            || (olde[i] != p->edited[0].ptr && in_image(olde[i], p->edited[0], q)))
            return true;
    return false;
}
static const char *kStaleText = "this level-0 annotation image is no longer the pyramid's: a live frame has uploaded a new pair since "
                                "rtdd_pyramid_image handed the pointer out -- ask rtdd_pyramid_image again";
// (an entry point that only READS annotation images: the same staleness rule)
int pyramid_check_read(rtdd_ctx *ctx, const void *a, const void *b) {
    if (ctx->pyr && (stale_live_pointer(ctx->pyr, a) || stale_live_pointer(ctx->pyr, b))) return fail(ctx, RTDD_ERR_STATE, kStaleText);
    return RTDD_OK;
}
int pyramid_note_write(rtdd_ctx *ctx, const void *scribble, const void *edited) {
    Pyramid *p = ctx->pyr;
    if (!p) return RTDD_OK;
    if (stale_live_pointer(p, scribble) || stale_live_pointer(p, edited)) return fail(ctx, RTDD_ERR_STATE, kStaleText);
    for (int l = 0; l < p->levels; l++)
        if (inside(p->scribble[l], scribble) || inside(p->edited[l], edited) || inside(p->scribble[l], edited)
            || inside(p->edited[l], scribble))
            p->annotation_dirty = true;
    // the coarsest depth image carries the injected labels of src/main.cpp:257-259, which an estimate only renews when the annotation
    // changed: whoever overwrites it through the library (rtdd_upload, rtdd_convert_to_float, rtdd_pyrup_depth) makes the next
    // estimate inject again
    if (p->levels > 0 && (inside(p->depth[p->levels - 1], scribble) || inside(p->depth[p->levels - 1], edited))) p->annotation_dirty = true;
    return RTDD_OK;
}

static int alloc_image(rtdd_ctx *ctx, Image &im, int rows, int cols, int elem, int fill, int images = 1) {
    im.rows = rows; im.cols = cols; im.elem = elem; im.images = images;
    im.pitch = ((size_t)cols * elem + 511) / 512 * 512;       // like cudaMallocPitch: rows padded to 512 B
    im.stride = im.pitch * (size_t)(rows > 0 ? rows : 1);
    RTDD_HIP(ctx, hipMalloc(&im.ptr, im.stride * (size_t)images));
    RTDD_HIP(ctx, hipMemsetAsync(im.ptr, fill, im.stride * (size_t)images, ctx->stream));
    return RTDD_OK;
}

static void free_image(Image &im) {
    if (im.ptr) (void)hipFree(im.ptr);
    im = Image();
}

static void live_free(Pyramid *p) {
    Live *v = p->live;
    if (!v) return;
    if (v->own_scribble) { p->scribble[0].ptr = v->own_scribble; p->edited[0].ptr = v->own_edited; }
    if (v->own_artistic) p->artistic.ptr = v->own_artistic;
    if (v->up) { (void)hipStreamSynchronize(v->up); (void)hipStreamDestroy(v->up); }
    if (v->copy) { (void)hipStreamSynchronize(v->copy); (void)hipStreamDestroy(v->copy); }
    for (int k = 0; k < 2; k++) {
        free_image(v->scribble_stage[k]); free_image(v->edited_stage[k]); free_image(v->u8_stage[k]); free_image(v->art_stage[k]);
        for (hipEvent_t e : {v->h2d_done[k], v->est_done[k], v->d2h_done[k]}) if (e) (void)hipEventDestroy(e);
    }
    if (v->status_host) (void)hipHostFree(v->status_host);
    for (auto &bb : v->bounce_up) for (Bounce &b : bb) if (b.ptr) (void)hipFree(b.ptr);
    if (v->host_bounce) (void)hipHostFree(v->host_bounce);
    if (v->bounce_art.ptr) (void)hipFree(v->bounce_art.ptr);
    delete v;
    p->live = nullptr;
}

void pyramid_free(rtdd_ctx *ctx) {
    if (!ctx->pyr) return;
    Pyramid *p = ctx->pyr;
    live_free(p);
    free_image(p->original); free_image(p->depth_u8); free_image(p->artistic);
    for (auto *v : {&p->gray, &p->scribble, &p->edited, &p->depth})
        for (auto &im : *v) free_image(im);
    delete p;
    ctx->pyr = nullptr;
}

}  // namespace rtdd

using namespace rtdd;

#define REQUIRE(ctx, cond, msg) \
    do { if (!(cond)) return fail((ctx), RTDD_ERR_INVALID, msg); } while (0)

#pragma GCC visibility push(default)
extern "C" {

int rtdd_pyramid_levels(int rows, int cols) {
    if (rows <= 0 || cols <= 0) return 0;
    const int m = rows < cols ? rows : cols;
    return (int)(log2((double)((m / 45) > 1 ? (m / 45) : 1)) + 1);          // src/main.cpp:95 (integer /45)
}

int rtdd_pyramid_create(rtdd_ctx *ctx, int rows, int cols) { return rtdd_pyramid_create_batch(ctx, rows, cols, 1); }

int rtdd_pyramid_batch(rtdd_ctx *ctx) { return ctx && ctx->pyr ? ctx->pyr->images : 0; }

int rtdd_pyramid_select(rtdd_ctx *ctx, int index) {
    if (!ctx) return RTDD_ERR_INVALID;
    if (!ctx->pyr) return fail(ctx, RTDD_ERR_STATE, "rtdd_pyramid_create has not been called");
    REQUIRE(ctx, index >= 0 && index < ctx->pyr->images, "image index outside the batch");
    REQUIRE(ctx, !ctx->pyr->live || ctx->pyr->live->submitted == ctx->pyr->live->waited, "live frames are in flight");
    ctx->pyr->sel = index;
    return RTDD_OK;
}

int rtdd_pyramid_create_batch(rtdd_ctx *ctx, int rows, int cols, int images) {
    if (!ctx) return RTDD_ERR_INVALID;
    REQUIRE(ctx, rows > 0 && cols > 0, "rows and cols must be positive");
    REQUIRE(ctx, images >= 1 && images <= 4096, "the batch must hold 1..4096 images");
    DeviceGuard g(ctx->device);
    { const int rc_ = settle_pending(ctx); if (rc_ != RTDD_OK) return rc_; }
    RTDD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    pyramid_free(ctx);
    Pyramid *p = new (std::nothrow) Pyramid();
    if (!p) return fail(ctx, RTDD_ERR_NOMEM, "pyramid");
    ctx->pyr = p;
    p->rows = rows; p->cols = cols; p->levels = rtdd_pyramid_levels(rows, cols); p->images = images;
    p->level_info.assign(p->levels, rtdd_solve_info{}); p->level_launch_images.assign(p->levels, 0);
    p->gray.resize(p->levels); p->scribble.resize(p->levels); p->edited.resize(p->levels); p->depth.resize(p->levels);
    int rc;
    if ((rc = alloc_image(ctx, p->original, rows, cols, 3, 0, images)) != RTDD_OK) return rc;
    if ((rc = alloc_image(ctx, p->artistic, rows, cols, 3, 0, images)) != RTDD_OK) return rc;
    if ((rc = alloc_image(ctx, p->depth_u8, rows, cols, 1, 255, images)) != RTDD_OK) return rc;
    int gr = rows, gc = cols;
    for (int l = 0; l < p->levels; l++) {
        const int lr = (int)(rows / powf(2, l)), lc = (int)(cols / powf(2, l));   // src/main.cpp:103,129
        if ((rc = alloc_image(ctx, p->gray[l], gr, gc, 1, 0, images)) != RTDD_OK) return rc;          // ceil chain
        if ((rc = alloc_image(ctx, p->scribble[l], lr, lc, 1, 0, images)) != RTDD_OK) return rc;      // :132-133
        if ((rc = alloc_image(ctx, p->edited[l], lr, lc, 3, 0, images)) != RTDD_OK) return rc;        // :130-131
        if ((rc = alloc_image(ctx, p->depth[l], lr, lc, 4, 0, images)) != RTDD_OK) return rc;
        for (int b = 0; b < images; b++)
            // :136
            if (lr > 0 && lc > 0 && (rc = launch_fill_f32(ctx, (float *)p->depth[l].at(b), p->depth[l].pitch, lr, lc,
                255.0f)) != RTDD_OK) return rc;
        gr = (gr + 1) / 2; gc = (gc + 1) / 2;
    }
    ctx->alloc_images = images;
    rc = rtdd_allocate(ctx, rows, cols, p->levels);                          // :149 (syncs)
    return rc;
}

int rtdd_pyramid_destroy(rtdd_ctx *ctx) {
    if (!ctx) return RTDD_ERR_INVALID;
    DeviceGuard g(ctx->device);
    { const int rc_ = settle_pending(ctx); if (rc_ != RTDD_OK) return rc_; }
    RTDD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    pyramid_free(ctx);
    return RTDD_OK;
}

int rtdd_pyramid_set_image(rtdd_ctx *ctx, const uint8_t *bgr, size_t pitch) {
    if (!ctx) return RTDD_ERR_INVALID;
    if (!ctx->pyr) return fail(ctx, RTDD_ERR_STATE, "rtdd_pyramid_create has not been called");
    Pyramid *p = ctx->pyr;
    REQUIRE(ctx, bgr && pitch >= (size_t)p->cols * 3, "bad image");
    DeviceGuard g(ctx->device);
    // (a new image resets the warm-start state a logged estimate ran on)
    { const int rc_ = settle_pending(ctx); if (rc_ != RTDD_OK) return rc_; }
    const int b = p->sel;                          // (a batched pyramid: the selected image)
    RTDD_HIP(ctx, hipMemcpy2DAsync(p->original.at(b), p->original.pitch, bgr, pitch, (size_t)p->cols * 3, p->rows, hipMemcpyDeviceToDevice,
        ctx->stream));
    // :158
    RTDD_HIP(ctx, hipMemcpy2DAsync(p->edited[0].at(b), p->edited[0].pitch, bgr, pitch, (size_t)p->cols * 3, p->rows,
        hipMemcpyDeviceToDevice, ctx->stream));
    // A new image is a new problem (the reference loads one image per process, src/main.cpp:93): everything an estimate carries
    // over to the next one -- the depth pyramid it warm-starts from (:136) and the coarse annotation levels, which
    // GPUPyrDownAnnotation only ever adds to -- goes back to its initial state.
    p->annotation_dirty = true;
    for (int l = 0; l < p->levels; l++) {
        if (p->scribble[l].ptr) RTDD_HIP(ctx,
            hipMemsetAsync(p->scribble[l].at(b), 0, p->scribble[l].pitch * p->scribble[l].rows, ctx->stream));
        if (l > 0 && p->edited[l].ptr) RTDD_HIP(ctx,
            hipMemsetAsync(p->edited[l].at(b), 0, p->edited[l].pitch * p->edited[l].rows, ctx->stream));
        if (p->depth[l].rows > 0 && p->depth[l].cols > 0) {
            const int rc_ = launch_fill_f32(ctx, (float *)p->depth[l].at(b), p->depth[l].pitch, p->depth[l].rows, p->depth[l].cols, 255.0f);
            if (rc_ != RTDD_OK) return rc_;
        }
    }
    int rc = launch_bgr2gray(ctx, (const uint8_t *)p->original.at(b), p->original.pitch, (uint8_t *)p->gray[0].at(b), p->gray[0].pitch,
        p->rows, p->cols);
    // the gray pyramid depends on the image only: built once here instead of once per estimate (:241-247)
    for (int l = 1; l < p->levels && rc == RTDD_OK; l++)
        rc = launch_pyrdown_u8(ctx, (const uint8_t *)p->gray[l - 1].at(b), p->gray[l - 1].pitch, p->gray[l - 1].rows, p->gray[l - 1].cols,
                               (uint8_t *)p->gray[l].at(b), p->gray[l].pitch);
    return rc;
}

int rtdd_pyramid_set_annotation(rtdd_ctx *ctx, const uint8_t *annotation, size_t pitch) {
    if (!ctx) return RTDD_ERR_INVALID;
    if (!ctx->pyr) return fail(ctx, RTDD_ERR_STATE, "rtdd_pyramid_create has not been called");
    Pyramid *p = ctx->pyr;
    REQUIRE(ctx, annotation && pitch >= (size_t)p->cols, "bad annotation");
    DeviceGuard g(ctx->device);
    p->annotation_dirty = true;
    return launch_decode_annotation(ctx, (const uint8_t *)p->original.at(p->sel), p->original.pitch, annotation, pitch,
                                    (uint8_t *)p->edited[0].at(p->sel), p->edited[0].pitch, (uint8_t *)p->scribble[0].at(p->sel),
                                        p->scribble[0].pitch, p->rows, p->cols);
}

int rtdd_pyramid_image(rtdd_ctx *ctx, int kind, int level, void **ptr, size_t *pitch, int *rows, int *cols) {
    if (!ctx) return RTDD_ERR_INVALID;
    if (!ctx->pyr) return fail(ctx, RTDD_ERR_STATE, "rtdd_pyramid_create has not been called");
    Pyramid *p = ctx->pyr;
    const Image *im = nullptr;
    REQUIRE(ctx, level >= 0 && level < p->levels, "level out of range");
    switch (kind) {
        case RTDD_IMG_ORIGINAL: im = level == 0 ? &p->original : nullptr; break;
        case RTDD_IMG_DEPTH_U8: im = level == 0 ? &p->depth_u8 : nullptr; break;
        case RTDD_IMG_ARTISTIC: im = level == 0 ? &p->artistic : nullptr; break;
        case RTDD_IMG_GRAY: im = &p->gray[level]; break;
        case RTDD_IMG_SCRIBBLE: im = &p->scribble[level]; break;
        case RTDD_IMG_EDITED: im = &p->edited[level]; break;
        case RTDD_IMG_DEPTH: im = &p->depth[level]; break;
        default: break;
    }
    REQUIRE(ctx, im != nullptr, "no such pyramid image");
    if (ptr) *ptr = im->at(p->sel);
    if (pitch) *pitch = im->pitch;
    if (rows) *rows = im->rows;
    if (cols) *cols = im->cols;
    return RTDD_OK;
}

int rtdd_pyramid_level_info(rtdd_ctx *ctx, int level, rtdd_solve_info *info, int *imagesPerLaunch) {
    if (!ctx || !info) return RTDD_ERR_INVALID;
    if (!ctx->pyr) return fail(ctx, RTDD_ERR_STATE, "rtdd_pyramid_create has not been called");
    REQUIRE(ctx, level >= 0 && level < ctx->pyr->levels, "level out of range");
    *info = ctx->pyr->level_info[level];
    if (imagesPerLaunch) *imagesPerLaunch = ctx->pyr->level_launch_images[level];
    return RTDD_OK;
}

int rtdd_pyramid_annotation_changed(rtdd_ctx *ctx) {
    if (!ctx) return RTDD_ERR_INVALID;
    if (!ctx->pyr) return fail(ctx, RTDD_ERR_STATE, "rtdd_pyramid_create has not been called");
    ctx->pyr->annotation_dirty = true;
    return RTDD_OK;
}

// first, n: the images of the context's (batched) pyramid the estimate covers -- the selected one, or all of them in the same launches.
// live: a live frame's own images (its annotation pair, the second target of its u8 map, its depth effect); default: not a live frame.
static int estimate_submit(rtdd_ctx *ctx, int maxIterations, unsigned long long *op_id, bool whole_batch, const LiveTargets &live) {
    if (!ctx->pyr) return fail(ctx, RTDD_ERR_STATE, "rtdd_pyramid_create has not been called");
    REQUIRE(ctx, maxIterations >= 0, "maxIterations must be >= 0");
    Pyramid *p = ctx->pyr;
    const int P = p->levels;
    const int first = whole_batch ? 0 : p->sel, n = whole_batch ? p->images : 1;
    int rc = RTDD_OK;
    if (p->annotation_dirty) {
        // every image of the batch in one launch per step, whatever the estimate covers: the flag is one for the whole batch, and the
        // down-sampling only ever adds (an up-to-date image stays as it is)
        DeviceGuard g(ctx->device);
        uint8_t *sc[32], *ed[32]; size_t sp[32], ep[32], zs[32], ze[32]; int lr[32], lc[32];
        if (P > 12) return fail(ctx, RTDD_ERR_INVALID, "more than 12 pyramid levels");
        for (int l = 0; l < P; l++) {
            sc[l] = (uint8_t *)p->scribble[l].ptr; ed[l] = (uint8_t *)p->edited[l].ptr; sp[l] = p->scribble[l].pitch;
            ep[l] = p->edited[l].pitch;
            zs[l] = p->scribble[l].stride; ze[l] = p->edited[l].stride; lr[l] = p->edited[l].rows; lc[l] = p->edited[l].cols;
        }
        // src/main.cpp:249-259: the P - 1 annotation levels and the coarsest level's injection, one launch (image_kernels.hip)
        rc = launch_annotation_pyramid(ctx, P, sc, sp, zs, ed, ep, ze, lr, lc, (float *)p->depth[P - 1].ptr, p->depth[P - 1].pitch,
                                       p->depth[P - 1].stride, p->images);
        if (rc != RTDD_OK) return rc;
        p->annotation_dirty = false;
    }
    PendingOp op;
    op.kind = PendingOp::kEstimate; op.opt = ctx->opt; op.maxIterations = maxIterations;
    op.batch_first = first; op.batch_n = n; op.live = live;
    rc = estimate_levels(ctx, maxIterations, P - 1, op.level_seq, first, n, live);
    if (rc == RTDD_OK && live.effect) rc = live_effect(ctx, live);      // src/main.cpp:190-230: the sticky effect, on this frame's map
    if (rc == RTDD_OK && !ctx->healing && ctx->opt.timeout_heal) {
        prune_confirmed(ctx);
        if (ctx->pending.size() >= kMaxPendingOps) { ctx->pending.clear(); ctx->pending_overflow = true; }
        op.id = ++ctx->op_counter;
        if (op_id) *op_id = op.id;
        ctx->pending.push_back(op);
    }
    return rc;
}

int rtdd_estimate_depth(rtdd_ctx *ctx, int maxIterations) {
    if (!ctx) return RTDD_ERR_INVALID;
    return estimate_submit(ctx, maxIterations, nullptr, /*whole_batch=*/false, LiveTargets());
}

int rtdd_estimate_depth_batch(rtdd_ctx *ctx, int maxIterations) {
    if (!ctx) return RTDD_ERR_INVALID;
    return estimate_submit(ctx, maxIterations, nullptr, /*whole_batch=*/true, LiveTargets());
}

// ---- live mode ---------------------------------------------------------------------------------------------------------------------
int rtdd_host_alloc(void **ptr, size_t bytes) {
    if (!ptr) return RTDD_ERR_INVALID;
    *ptr = nullptr;
    return hipHostMalloc(ptr, bytes > 0 ? bytes : 1, hipHostMallocDefault) == hipSuccess ? RTDD_OK : RTDD_ERR_NOMEM;
}

int rtdd_host_free(void *ptr) { return !ptr || hipHostFree(ptr) == hipSuccess ? RTDD_OK : RTDD_ERR_HIP; }

static int live_create(rtdd_ctx *ctx) {
    Pyramid *p = ctx->pyr;
    if (p->live) return RTDD_OK;
    Live *v = new (std::nothrow) Live();
    if (!v) return fail(ctx, RTDD_ERR_NOMEM, "live state");
    p->live = v;
    // (never joins the null stream implicitly: the context's stream may be it)
    RTDD_HIP(ctx, hipStreamCreateWithFlags(&v->copy, hipStreamNonBlocking));
    RTDD_HIP(ctx, hipStreamCreateWithFlags(&v->up, hipStreamNonBlocking));
    int rc;
    for (int k = 0; k < 2; k++) {
        if ((rc = alloc_image(ctx, v->scribble_stage[k], p->rows, p->cols, 1, 0)) != RTDD_OK) return rc;
        if ((rc = alloc_image(ctx, v->edited_stage[k], p->rows, p->cols, 3, 0)) != RTDD_OK) return rc;
        if ((rc = alloc_image(ctx, v->u8_stage[k], p->rows, p->cols, 1, 0)) != RTDD_OK) return rc;
        RTDD_HIP(ctx, hipEventCreateWithFlags(&v->h2d_done[k], hipEventDisableTiming));
        RTDD_HIP(ctx, hipEventCreateWithFlags(&v->est_done[k], hipEventDisableTiming));
        RTDD_HIP(ctx, hipEventCreateWithFlags(&v->d2h_done[k], hipEventDisableTiming));
    }
    if (v->scribble_stage[0].pitch != p->scribble[0].pitch || v->edited_stage[0].pitch != p->edited[0].pitch
        || v->u8_stage[0].pitch != p->depth_u8.pitch)
        return fail(ctx, RTDD_ERR_STATE, "live staging images and pyramid images differ in pitch");
    v->own_scribble = p->scribble[0].ptr; v->own_edited = p->edited[0].ptr;
    RTDD_HIP(ctx, hipHostMalloc((void **)&v->status_host, 2 * 8 * sizeof(int), hipHostMallocDefault));
    for (int i = 0; i < 16; i++) v->status_host[i] = 0;
    RTDD_HIP(ctx, hipStreamSynchronize(ctx->stream));                                // the staging images' fills
    return RTDD_OK;
}

int rtdd_live_pending(rtdd_ctx *ctx) {
    if (!ctx || !ctx->pyr || !ctx->pyr->live) return 0;
    return (int)(ctx->pyr->live->submitted - ctx->pyr->live->waited);
}

// Frame k's staged u8 map (`map`) and / or its artistic image (`art`), and the control words, to the host on the copy stream, and wait.
// A host image whose pitch is no multiple of four: the staged image comes over WHOLE -- its rows, padding included, are one contiguous
// block -- into a page-locked buffer and the host copies the rows out; the runtime's 2-D copy takes 9 us per row for such a pitch, and
// a re-pitching kernel on the copy stream would queue behind the next frame's persistent launches (1921 x 1081: 1.39 ms per pipelined
// frame).
static int live_fetch(rtdd_ctx *ctx, Live *v, int k, bool map, bool art) {
    Pyramid *p = ctx->pyr;
    const Live::Frame &f = v->frame[k];
    struct Part { bool on; uint8_t *host; size_t pitch, width; const Image *st; size_t off; } parts[2] = {
        {map, f.host, f.pitch, (size_t)p->cols, &v->u8_stage[k], 0},
            {art && f.effect != 0, f.art_host, f.art_pitch, (size_t)p->cols * 3, &v->art_stage[k], 0}};
    size_t bounce = 0;
    for (Part &q : parts)
        // (off - 1: its place in the bounce buffer)
        if (q.on && q.pitch % 4 != 0 && p->rows > 1) { q.off = bounce + 1; bounce += q.st->pitch * (size_t)p->rows; }
    if (bounce > v->host_bounce_bytes) {
        if (v->host_bounce) {
            RTDD_HIP(ctx, hipStreamSynchronize(v->copy)); RTDD_HIP(ctx, hipHostFree(v->host_bounce));
            v->host_bounce = nullptr; v->host_bounce_bytes = 0;
        }
        RTDD_HIP(ctx, hipHostMalloc((void **)&v->host_bounce, bounce, hipHostMallocDefault));
        v->host_bounce_bytes = bounce;
    }
    for (const Part &q : parts) {
        if (!q.on) continue;
        if (q.off) RTDD_HIP(ctx,
            hipMemcpyAsync(v->host_bounce + q.off - 1, q.st->ptr, q.st->pitch * (size_t)p->rows, hipMemcpyDeviceToHost, v->copy));
        else RTDD_HIP(ctx, hipMemcpy2DAsync(q.host, q.pitch, q.st->ptr, q.st->pitch, q.width, p->rows, hipMemcpyDeviceToHost, v->copy));
    }
    RTDD_HIP(ctx, hipMemcpyAsync(v->status_host + 8 * k, ctx->sync_words, 8 * sizeof(int), hipMemcpyDeviceToHost, v->copy));
    RTDD_HIP(ctx, hipStreamSynchronize(v->copy));
    for (const Part &q : parts)
        if (q.on && q.off)
            for (int y = 0; y < p->rows; y++) std::memcpy(q.host + (size_t)y * q.pitch,
                v->host_bounce + q.off - 1 + (size_t)y * q.st->pitch, q.width);
    return RTDD_OK;
}

int rtdd_live_wait(rtdd_ctx *ctx) {
    if (!ctx) return RTDD_ERR_INVALID;
    if (!ctx->pyr || !ctx->pyr->live || ctx->pyr->live->submitted == ctx->pyr->live->waited) return fail(ctx, RTDD_ERR_STATE,
        "no frame in flight");
    Pyramid *p = ctx->pyr;
    Live *v = p->live;
    DeviceGuard g(ctx->device);
    const int k = (int)(v->waited % 2);
    const Live::Frame &fr = v->frame[k];
    const bool fetch_map = !fr.direct, fetch_art = fr.effect != 0 && !fr.art_queued;
    if (!fetch_map && !fetch_art) RTDD_HIP(ctx, hipEventSynchronize(v->d2h_done[k]));
    else {
        // (a frame with one part stored / queued on the compute stream and the other staged: the one event behind both was recorded there)
        RTDD_HIP(ctx, hipEventSynchronize(fr.direct ? v->d2h_done[k] : v->est_done[k]));
        { const int rc_ = live_fetch(ctx, v, k, fetch_map, fetch_art); if (rc_ != RTDD_OK) return rc_; }
    }
    // the usual case: the frame is good, and so is everything logged before it
    if (v->status_host[8 * k + kSyncStatus] == 0) {
        size_t n = 0;
        while (n < ctx->pending.size() && ctx->pending[n].id <= v->frame[k].op_id) n++;
        ctx->pending.erase(ctx->pending.begin(), ctx->pending.begin() + n);
        v->frame[k].in_flight = false; v->waited++;
        return RTDD_OK;
    }
    // A sweep launch in front of this frame's download gave up (include/rtdd.h, RTDD_ERR_TIMEOUT): drain both streams, let the
    // status check run the logged estimates again (each brings its staged u8 map and its effect with it), then fetch every frame in
    // flight again.
    RTDD_HIP(ctx, hipStreamSynchronize(v->up));
    RTDD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    RTDD_HIP(ctx, hipStreamSynchronize(v->copy));
    note_status_writer(ctx);
    const int rc = check_persistent_status(ctx);
    if (rc != RTDD_OK) return rc;
    for (unsigned long long f = v->waited; f < v->submitted; f++) {
        const int j = (int)(f % 2);
        // (a frame whose map the copy-back kernel stores in the host's buffer itself has just been run again into that buffer; an
        // artistic image is always rendered into its staging slot)
        { const int rc_ = live_fetch(ctx, v, j, !v->frame[j].direct, true); if (rc_ != RTDD_OK) return rc_; }
        v->status_host[8 * j + kSyncStatus] = 0;
    }
    v->frame[k].in_flight = false; v->waited++;
    return RTDD_OK;
}

int rtdd_live_submit(rtdd_ctx *ctx, const uint8_t *hostScribble, size_t scribblePitch, const uint8_t *hostEdited, size_t editedPitch,
                     int maxIterations, uint8_t *hostDepthU8, size_t depthPitch) {
    return rtdd_live_submit_ex(ctx, hostScribble, scribblePitch, hostEdited, editedPitch, maxIterations, hostDepthU8, depthPitch,
        RTDD_EFFECT_NONE, nullptr, 0);
}

int rtdd_live_submit_ex(rtdd_ctx *ctx, const uint8_t *hostScribble, size_t scribblePitch, const uint8_t *hostEdited, size_t editedPitch,
                        int maxIterations, uint8_t *hostDepthU8, size_t depthPitch, int effect, uint8_t *hostArtistic,
                            size_t artisticPitch) {
    if (!ctx) return RTDD_ERR_INVALID;
    if (!ctx->pyr) return fail(ctx, RTDD_ERR_STATE, "rtdd_pyramid_create has not been called");
    Pyramid *p = ctx->pyr;
    REQUIRE(ctx, (hostScribble == nullptr) == (hostEdited == nullptr),
        "scribble and edited images come together (or neither: the annotation is unchanged)");
    REQUIRE(ctx, !hostScribble || (scribblePitch >= (size_t)p->cols && editedPitch >= (size_t)p->cols * 3), "pitch smaller than a row");
    REQUIRE(ctx, hostDepthU8 && depthPitch >= (size_t)p->cols && maxIterations >= 0, "bad output buffer or iteration count");
    REQUIRE(ctx, effect >= RTDD_EFFECT_NONE && effect <= RTDD_EFFECT_HAZE, "unknown effect");
    REQUIRE(ctx, effect == RTDD_EFFECT_NONE || (hostArtistic && artisticPitch >= (size_t)p->cols * 3),
        "an effect needs a host image for its result");
    REQUIRE(ctx, p->images == 1, "live frames run on a single-image pyramid (rtdd_pyramid_create)");
    DeviceGuard g(ctx->device);
    int rc = live_create(ctx);
    if (rc != RTDD_OK) return rc;
    Live *v = p->live;
    // two frames in flight at most: the slot is free again
    if (v->submitted - v->waited >= 2 && (rc = rtdd_live_wait(ctx)) != RTDD_OK) return rc;
    const int k = (int)(v->submitted % 2);
    const bool lone = v->submitted == v->waited;          // no other frame in flight: nothing for a copy to overlap
    if (effect != RTDD_EFFECT_NONE && !v->art_stage[0].ptr) {          // the first frame with an effect: its two staging images
        for (int j = 0; j < 2; j++)
            if ((rc = alloc_image(ctx, v->art_stage[j], p->rows, p->cols, 3, 0)) != RTDD_OK) return rc;
        if (v->art_stage[0].pitch != p->artistic.pitch) return fail(ctx, RTDD_ERR_STATE,
            "live staging images and pyramid images differ in pitch");
        v->own_artistic = p->artistic.ptr;
    }
    if (hostScribble) {
        // upload into the staging pair the pyramid is NOT looking at: the only frame that can still be in flight reads the other one
        // (at most two frames in flight, and the older one has been waited for above)
        const int u = p->scribble[0].ptr == v->scribble_stage[0].ptr ? 1 : 0;
        // (no other frame in flight: the upload goes on the compute stream itself, no event between the streams: 1080p 1.335 -> 1.324 ms
        // one frame at a time, 4K 2.21 -> 2.18)
        hipStream_t us = lone ? ctx->stream : v->up;
        // A host image whose pitch is no multiple of four (copy_h2d above) goes as one linear copy into a contiguous buffer of this
        // frame's parity on the upload stream, and is re-pitched into the staging image by a kernel on the COMPUTE stream, behind the
        // upload's event: on the upload stream that kernel would queue behind the running frame's persistent launches.  (The buffer is
        // written again two frames later, after this frame has been waited for.)
        struct { const uint8_t *host; size_t hp, width; Image *dst; Bounce *b; bool bounce; } ups[2] = {
            {hostScribble, scribblePitch, (size_t)p->cols, &v->scribble_stage[u], &v->bounce_up[k][0], false},
            {hostEdited, editedPitch, (size_t)p->cols * 3, &v->edited_stage[u], &v->bounce_up[k][1], false}};
        for (auto &q : ups) {
            q.bounce = wants_bounce(q.hp, q.width, p->rows);
            if (!q.bounce) { RTDD_HIP(ctx,
                hipMemcpy2DAsync(q.dst->ptr, q.dst->pitch, q.host, q.hp, q.width, p->rows, hipMemcpyHostToDevice, us)); continue; }
            if ((rc = bounce_reserve(ctx, *q.b, q.width * (size_t)p->rows, ctx->stream)) != RTDD_OK) return rc;
            RTDD_HIP(ctx, hipMemcpyAsync(q.b->ptr, q.host, q.width * (size_t)p->rows, hipMemcpyHostToDevice, us));
        }
        if (!lone) {
            RTDD_HIP(ctx, hipEventRecord(v->h2d_done[k], v->up));
            RTDD_HIP(ctx, hipStreamWaitEvent(ctx->stream, v->h2d_done[k], 0));
        }
        for (auto &q : ups)
            if (q.bounce && (rc = launch_repitch(ctx, ctx->stream, q.b->ptr, q.width, q.dst->ptr, q.dst->pitch, q.width,
                p->rows)) != RTDD_OK) return rc;
        // the pyramid's level-0 annotation IS the uploaded pair: no copy
        p->scribble[0].ptr = v->scribble_stage[u].ptr; p->edited[0].ptr = v->edited_stage[u].ptr;
        p->annotation_dirty = true;
    }
    unsigned long long op_id = ctx->op_counter;
    // Where the map goes.  What it must not do is make a copy stream wait for the compute stream ON THE DEVICE: that costs the compute
    // stream ~80 us per frame on this runtime (with both copies removed and only the event pair left: still 80; EXPERIMENTS.md round 5).
    //   * another frame in flight (a pipelined loop): the map goes to the frame's staging slot, and rtdd_live_wait -- the HOST, which is
    //     idle for 0.9 of such a frame -- downloads it once the estimate's event has come; the copy overlaps the next frame's arithmetic
    //     (1080p 1.21 -> 1.14 ms per frame, 4K 1.57 -> 1.52);
    //   * no other frame in flight (one frame at a time: nothing could overlap a download) and a page-locked host buffer (rtdd_host_alloc,
    //     hipHostMalloc, hipHostRegister: device-visible): the estimate's copy-back kernel stores the u8 map straight into it, ~20 us
    //     per MB of posted writes, and the frame ends with an event on the compute stream (1080p 1.42 -> 1.33 ms, 4K 2.31 -> 2.20).
    // The artistic image of a frame with an effect (6.2 MB at 1080p) is rendered into the frame's staging slot and follows the same
    // rule: downloaded by rtdd_live_wait from the host when pipelined, queued behind the effect on the compute stream when alone.
    const bool want_direct = ctx->opt.live_zero_copy == 2 || (ctx->opt.live_zero_copy == 1 && lone);
    uint8_t *direct = want_direct ? live_device_view(hostDepthU8, depthPitch, p->rows, p->cols) : nullptr;
    // (the estimate's copy-back writes this frame's map into RTDD_IMG_DEPTH_U8 and into the frame's second target: estimate_levels)
    LiveTargets lt;
    lt.scribble = p->scribble[0].ptr; lt.edited = p->edited[0].ptr;
    lt.u8 = direct ? direct : (uint8_t *)v->u8_stage[k].ptr; lt.u8_pitch = direct ? depthPitch : 0;
    lt.effect = effect;
    if (effect != RTDD_EFFECT_NONE) {
        lt.artistic = (uint8_t *)v->art_stage[k].ptr; lt.artistic_pitch = v->art_stage[k].pitch;
        // RTDD_IMG_ARTISTIC names the newest frame's image (like the annotation pair: no copy)
        p->artistic.ptr = v->art_stage[k].ptr;
    }
    rc = estimate_submit(ctx, maxIterations, &op_id, /*whole_batch=*/false, lt);
    if (rc != RTDD_OK) return rc;
    const bool art_queued = effect != RTDD_EFFECT_NONE && lone;
    if (art_queued) {
        rc = copy_d2h(ctx, v->bounce_art, hostArtistic, artisticPitch, v->art_stage[k].ptr, v->art_stage[k].pitch, (size_t)p->cols * 3,
            p->rows, ctx->stream);
        if (rc != RTDD_OK) return rc;
    }
    if (direct) {
        RTDD_HIP(ctx, hipMemcpyAsync(v->status_host + 8 * k, ctx->sync_words, 8 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        RTDD_HIP(ctx, hipEventRecord(v->d2h_done[k], ctx->stream));
    } else {
        // staged: the download is issued by rtdd_live_wait, from the HOST, once this event has come -- a copy stream made to wait for the
        // compute stream on the device costs the compute stream ~80 us per frame (above); the host has the time (it is idle for 0.9 of a
        // pipelined frame), and the copy still overlaps the next frame's arithmetic
        RTDD_HIP(ctx, hipEventRecord(v->est_done[k], ctx->stream));
    }
    Live::Frame &fr = v->frame[k];
    fr.host = hostDepthU8; fr.pitch = depthPitch; fr.in_flight = true; fr.op_id = op_id; fr.direct = direct != nullptr;
    fr.effect = effect; fr.art_host = hostArtistic; fr.art_pitch = artisticPitch; fr.art_queued = art_queued;
    v->submitted++;
    return RTDD_OK;
}

}  // extern "C"
#pragma GCC visibility pop

namespace rtdd {

int estimate_levels(rtdd_ctx *ctx, int maxIterations, int from_level, int *level_seq, int first, int n, const LiveTargets &live) {
    Pyramid *p = ctx->pyr;
    if (!p) return fail(ctx, RTDD_ERR_STATE, "the pyramid is gone");
    if (first < 0 || n < 1 || first + n > p->images) return fail(ctx, RTDD_ERR_INVALID, "images outside the pyramid's batch");
    const int P = p->levels;
    if (from_level > P - 1) from_level = P - 1;
    int rc = RTDD_OK;
    for (int l = from_level; l >= 0 && rc == RTDD_OK; l--) {                   // src/main.cpp:261-288
        const int iters = (int)(maxIterations / powf(2.0, (P - 1) - l));       // :263
        const bool solved = p->depth[l].rows > 0 && p->depth[l].cols > 0;
        // Two launches less per level than the calls spelt out: above the finest level the solver's copy-back is left to the pyrUp
        // kernel (which reads the result plane directly and writes depth[l] on the side); at the finest level the copy-back also
        // writes the u8 map (:290).  Same values in the same buffers (tests/test_gpu_cascade.py compares every level's image).
        SolveTargets t;
        t.logged = false;                                                       // (the estimate is logged as one call)
        t.defer_finish = solved && l > 0;
        if (l == 0) {
            t.u8 = (uint8_t *)p->depth_u8.at(first); t.u8_pitch = p->depth_u8.pitch;
            // (a live frame: its staging slot, or the host's buffer, too)
            t.u8b = live.u8; t.u8b_pitch = live.u8_pitch ? live.u8_pitch : p->depth_u8.pitch;
        }
        // the level of images first .. first + n - 1 in the same launches (blockIdx.z = image: Batch, rtdd_internal.hpp)
        t.batch.n = n; t.batch.first = first;
        t.batch.depth = p->depth[l].stride; t.batch.scribble = p->scribble[l].stride; t.batch.gray = p->gray[l].stride;
        t.batch.u8 = p->depth_u8.stride;
        SolveOutcome done;
        if (solved) {
            // GPUMatrixFreeSolver(..., beta, CUDAIteration, CUDAThreshold, level): exactly `iters` sweeps (:266-268)
            rtdd_solve_params sp;
            sp.method = RTDD_METHOD_CHEBYSHEV_JACOBI; sp.maxIterations = iters; sp.tolerance = 0.0f; sp.checkEvery = 0;
            sp.relaxation = 0.0f;
            rc = solve_with(ctx, (float *)p->depth[l].at(first), p->depth[l].pitch, (const uint8_t *)p->scribble[l].at(first),
                p->scribble[l].pitch,
                            (const uint8_t *)p->gray[l].at(first), p->gray[l].pitch, p->depth[l].rows, p->depth[l].cols, l, &sp, nullptr, t,
                                &done);
            if (level_seq && l < 32) level_seq[l] = done.seq;
            if (rc == RTDD_OK) {
                p->level_info[l] = ctx->last_info; p->level_launch_images[l] = ctx->last_launch_images;
                if (ctx->last_info.kernel == 2) p->level_info[l].temporal_depth = ctx->last_nominal_depth;
            }
        }
        if (rc == RTDD_OK && l > 0) {
            DeviceGuard g(ctx->device);
            const float *src = (const float *)p->depth[l].at(first); size_t sp = p->depth[l].pitch;
            float *coarse_out = nullptr;
            PyrupBatch pb;
            pb.n = n; pb.src = p->depth[l].stride; pb.dst = p->depth[l - 1].stride; pb.edited = p->edited[l - 1].stride;
            pb.mask = p->scribble[l - 1].stride; pb.coarse = p->depth[l].stride;
            if (t.defer_finish) {                                               // the level's result is still in the solver's plane
                const size_t ip = plane_pitch(p->depth[l].cols);
                const Level Lv = ctx->levels[l].view(first);
                src = Lv.P(done.plane, ip); sp = ip * sizeof(float); pb.src = Lv.elems * sizeof(float);
                coarse_out = (float *)p->depth[l].at(first);
            }
            // guarded like k_finish (with the solve's sequence number): when level l's sweeps gave up it stores nothing, depth[l] keeps
            // level
            // l's input and the level can be run again
            rc = launch_pyrup_inject(ctx, src, sp, p->depth[l].rows, p->depth[l].cols,
                                     (float *)p->depth[l - 1].at(first), p->depth[l - 1].pitch, p->depth[l - 1].rows, p->depth[l - 1].cols,
                                     (const uint8_t *)p->edited[l - 1].at(first), p->edited[l - 1].pitch,
                                     (const uint8_t *)p->scribble[l - 1].at(first), p->scribble[l - 1].pitch, coarse_out, p->depth[l].pitch,
                                     /*guard_seq=*/solved ? done.seq : 0, &pb);     // :272-283
        }
    }
    if (rc != RTDD_OK) return rc;
    if (p->depth[0].rows > 0 && p->depth[0].cols > 0) return RTDD_OK;         // the u8 map left the solver's copy-back (above)
    DeviceGuard g(ctx->device);
    // (an image too small for a finest level: nothing above ran either)
    for (int b = first; b < first + n && rc == RTDD_OK; b++)
        rc = launch_depth_to_u8(ctx, (const float *)p->depth[0].at(b), p->depth[0].pitch, (uint8_t *)p->depth_u8.at(b), p->depth_u8.pitch,
                                p->rows, p->cols);   // :290
    return rc;
}

// A live frame's sticky effect (src/main.cpp:190-230) on the pyramid's level-0 images, into the frame's staging image.  Queued on the
// compute stream right behind the estimate's copy-back: the f32 map it reads is still in the L2s / the Infinity Cache.
int live_effect(rtdd_ctx *ctx, const LiveTargets &live) {
    Pyramid *p = ctx->pyr;
    if (!p) return fail(ctx, RTDD_ERR_STATE, "the pyramid is gone");
    if (p->rows <= 0 || p->cols <= 0 || !live.artistic) return RTDD_OK;
    DeviceGuard g(ctx->device);
    const uint8_t *orig = (const uint8_t *)p->original.ptr; const float *depth = (const float *)p->depth[0].ptr;
    switch (live.effect) {
        case RTDD_EFFECT_DEFOCUS:
            return launch_defocus(ctx, orig, p->original.pitch, depth, p->depth[0].pitch, live.artistic, live.artistic_pitch, p->rows,
                p->cols);
        case RTDD_EFFECT_DESATURATION:
            return launch_desaturate(ctx, orig, p->original.pitch, (const uint8_t *)p->gray[0].ptr, p->gray[0].pitch, depth,
                p->depth[0].pitch,
                                     live.artistic, live.artistic_pitch, p->rows, p->cols);
        case RTDD_EFFECT_HAZE:
            return launch_haze(ctx, orig, p->original.pitch, depth, p->depth[0].pitch, live.artistic, live.artistic_pitch, p->rows,
                p->cols);
        default: return RTDD_OK;
    }
}

int estimate_replay(rtdd_ctx *ctx, const PendingOp &op, int failed_seq) {
    Pyramid *p = ctx->pyr;
    if (!p) return fail(ctx, RTDD_ERR_STATE, "the pyramid is gone");
    int from = -1;                                  // the level whose solve gave up; an estimate queued behind the failed call: every level
    for (int l = 0; l < 32; l++) if (failed_seq != 0 && op.level_seq[l] == failed_seq) from = l;
    if (from < 0) for (int l = 31; l >= 0 && from < 0; l--) if (op.level_seq[l] != 0) from = l;
    // a live frame is run again on ITS annotation pair into ITS u8 slot and ITS artistic image (the newest frame's replay comes last: the
    // pyramid ends up naming its images)
    if (op.live.scribble) { p->scribble[0].ptr = op.live.scribble; p->edited[0].ptr = op.live.edited; }
    if (op.live.effect && op.live.artistic) p->artistic.ptr = op.live.artistic;
    int rc = from >= 0 ? estimate_levels(ctx, op.maxIterations, from, nullptr, op.batch_first, op.batch_n, op.live) : RTDD_OK;
    if (rc == RTDD_OK && op.live.effect) rc = live_effect(ctx, op.live);
    return rc;
}

}  // namespace rtdd

#pragma GCC visibility push(default)
extern "C" {

int rtdd_refine_depth(rtdd_ctx *ctx, const rtdd_solve_params *params, rtdd_solve_info *info) {
    if (!ctx) return RTDD_ERR_INVALID;
    if (!ctx->pyr) return fail(ctx, RTDD_ERR_STATE, "rtdd_pyramid_create has not been called");
    Pyramid *p = ctx->pyr;
    // the u8 map is written by the solve's own copy-back (k_finish: the same rounding as k_depth_to_u8), so that a solve that has to be
    // run again after a timed-out persistent launch brings the map with it
    const int b = p->sel;
    SolveTargets t;
    t.u8 = (uint8_t *)p->depth_u8.at(b); t.u8_pitch = p->depth_u8.pitch;
    t.batch.first = b;
    return solve_with(ctx, (float *)p->depth[0].at(b), p->depth[0].pitch, (const uint8_t *)p->scribble[0].at(b), p->scribble[0].pitch,
                      (const uint8_t *)p->gray[0].at(b), p->gray[0].pitch, p->rows, p->cols, 0, params, info, t, nullptr);
}

int rtdd_bgr2gray(rtdd_ctx *ctx, const uint8_t *bgr, size_t bgrPitch, uint8_t *gray, size_t grayPitch, int rows, int cols) {
    if (!ctx) return RTDD_ERR_INVALID;
    REQUIRE(ctx, bgr && gray && rows > 0 && cols > 0 && bgrPitch >= (size_t)cols * 3 && grayPitch >= (size_t)cols, "bad argument");
    DeviceGuard g(ctx->device);
    return launch_bgr2gray(ctx, bgr, bgrPitch, gray, grayPitch, rows, cols);
}

int rtdd_pyrdown_gray(rtdd_ctx *ctx, const uint8_t *src, size_t srcPitch, int rows, int cols, uint8_t *dst, size_t dstPitch) {
    if (!ctx) return RTDD_ERR_INVALID;
    REQUIRE(ctx, src && dst && rows > 0 && cols > 0 && srcPitch >= (size_t)cols && dstPitch >= (size_t)((cols + 1) / 2), "bad argument");
    DeviceGuard g(ctx->device);
    return launch_pyrdown_u8(ctx, src, srcPitch, rows, cols, dst, dstPitch);
}

int rtdd_pyrup_depth(rtdd_ctx *ctx, const float *src, size_t srcPitch, int rows, int cols, float *dst, size_t dstPitch, int dstRows,
    int dstCols) {
    if (!ctx) return RTDD_ERR_INVALID;
    REQUIRE(ctx, src && dst && rows > 0 && cols > 0 && dstRows > 0 && dstCols > 0 && srcPitch >= (size_t)cols * 4
        && dstPitch >= (size_t)dstCols * 4, "bad argument");
    DeviceGuard g(ctx->device);
    // (as rtdd_index_to_weight: `src` may be the output of a logged solve whose persistent launch gave up -- the spelt-out cascade,
    // solve -> pyrUp -> inject -> solve, queued asynchronously)
    { const int rc_ = settle_pending(ctx); if (rc_ != RTDD_OK) return rc_; }
    { const int rc_ = pyramid_note_write(ctx, dst, dst); if (rc_ != RTDD_OK) return rc_; }
    return launch_pyrup_inject(ctx, src, srcPitch, rows, cols, dst, dstPitch, dstRows, dstCols, nullptr, 0, nullptr, 0);
}

int rtdd_depth_to_u8(rtdd_ctx *ctx, const float *src, size_t srcPitch, uint8_t *dst, size_t dstPitch, int rows, int cols) {
    if (!ctx) return RTDD_ERR_INVALID;
    REQUIRE(ctx, src && dst && rows > 0 && cols > 0 && srcPitch >= (size_t)cols * 4 && dstPitch >= (size_t)cols, "bad argument");
    DeviceGuard g(ctx->device);
    { const int rc_ = settle_pending(ctx); if (rc_ != RTDD_OK) return rc_; }     // (as rtdd_pyrup_depth)
    return launch_depth_to_u8(ctx, src, srcPitch, dst, dstPitch, rows, cols);
}

int rtdd_upload(rtdd_ctx *ctx, void *dev, size_t devPitch, const void *host, size_t hostPitch, size_t widthBytes, int rows) {
    if (!ctx) return RTDD_ERR_INVALID;
    REQUIRE(ctx, dev && host && rows >= 0 && devPitch >= widthBytes && hostPitch >= widthBytes, "bad argument");
    DeviceGuard g(ctx->device);
    // the destination may be an input of a logged call that still has to be run again: settle first (this call synchronises anyway)
    { const int rc_ = settle_pending(ctx); if (rc_ != RTDD_OK) return rc_; }
    { const int rc_ = pyramid_note_write(ctx, dev, dev); if (rc_ != RTDD_OK) return rc_; }
    { const int rc_ = copy_h2d(ctx, ctx->bounce, dev, devPitch, host, hostPitch, widthBytes, rows,
        ctx->stream); if (rc_ != RTDD_OK) return rc_; }
    RTDD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RTDD_OK;
}

int rtdd_download(rtdd_ctx *ctx, void *host, size_t hostPitch, const void *dev, size_t devPitch, size_t widthBytes, int rows) {
    if (!ctx) return RTDD_ERR_INVALID;
    REQUIRE(ctx, dev && host && rows >= 0 && devPitch >= widthBytes && hostPitch >= widthBytes, "bad argument");
    DeviceGuard g(ctx->device);
    { const int rc_ = copy_d2h(ctx, ctx->bounce, host, hostPitch, dev, devPitch, widthBytes, rows,
        ctx->stream); if (rc_ != RTDD_OK) return rc_; }
    RTDD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int heals = ctx->heals;
    const int rc = check_persistent_status(ctx);  // what was just downloaded may come from a persistent launch that gave up ...
    if (rc != RTDD_OK || ctx->heals == heals) return rc;
    // ... and has been run again since
    { const int rc_ = copy_d2h(ctx, ctx->bounce, host, hostPitch, dev, devPitch, widthBytes, rows,
        ctx->stream); if (rc_ != RTDD_OK) return rc_; }
    RTDD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RTDD_OK;
}

}  // extern "C"
#pragma GCC visibility pop
