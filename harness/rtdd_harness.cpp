// rtdd_harness -- headless stand-in for the reference's interactive shell (/root/reference/src/main.cpp).
//
// main.cpp is an OpenCV HighGUI event loop over cv::cuda::GpuMat and cannot be built on a ROCm box
// (SURVEY.md section 0); this program makes the SAME sequence of calls through librtdd.so's C ABI
// with plain files instead of windows:
//     -i image.jpg  -a annotation.png      (main.cpp:81-90; JPEG (jpeg_reader.hpp: libjpeg's default decode, restated), 8-bit PNG, binary PPM / PGM)
//     key 'd'  -> one depth estimate        (main.cpp:232-295)  -> <out>DepthMap.pgm     (main.cpp:306-310)
//     key 's'  -> also the annotated image   (main.cpp:298-303)  -> <out>AnnotatedImage.ppm: the image with the scribbles painted in
//     key 'b'/'g'/'h' -> --effect defocus|desaturation|haze     -> <out>ArtisticEffect.ppm (main.cpp:190-230, 312-316)
//     key 't'  -> prints "Processing Time"  (main.cpp:320-322; wall clock here, the reference uses clock()); the process's one-time costs
//                 (~20 ms: code objects, first allocations) are paid by a warm-up on a context of its own first -- --cold leaves it out
//     --paint x,y,label,radius  = a mouse drag sample (main.cpp:46-62), repeatable; --paint-at F:x,y,label,radius = the same while --live
//                                 runs, in front of frame F (the user painting into a live view)
//     --refine sor|mg|auto [--tolerance T] = extension: converge the finest level after the estimate (rtdd_refine_depth)
// and adds what the reference cannot do: --devices N --batch B runs B independent estimates
// round-robin over N GPUs, one host thread + one HIP stream + one rtdd_ctx per GPU, no collective.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <zlib.h>

#include "rtdd.h"
#include "jpeg_reader.hpp"

struct Pnm { int w = 0, h = 0, ch = 0; std::vector<unsigned char> px; };

static bool read_pnm(const std::string &path, Pnm &im) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    char magic[3] = {0, 0, 0};
    int maxv = 0;
    auto skip = [&]() { int c; while ((c = std::fgetc(f)) != EOF) { if (c == '#') { while ((c = std::fgetc(f)) != EOF && c != '\n') {} } else if (c > ' ') { std::ungetc(c, f); break; } } };
    if (std::fscanf(f, "%2s", magic) != 1) { std::fclose(f); return false; }
    im.ch = !std::strcmp(magic, "P6") ? 3 : (!std::strcmp(magic, "P5") ? 1 : 0);
    if (!im.ch) { std::fclose(f); return false; }
    skip(); if (std::fscanf(f, "%d", &im.w) != 1) { std::fclose(f); return false; }
    skip(); if (std::fscanf(f, "%d", &im.h) != 1) { std::fclose(f); return false; }
    skip(); if (std::fscanf(f, "%d", &maxv) != 1 || maxv != 255) { std::fclose(f); return false; }
    std::fgetc(f);
    if (im.w <= 0 || im.h <= 0 || im.w > 65535 || im.h > 65535) { std::fclose(f); return false; }
    const long at = std::ftell(f);
    std::fseek(f, 0, SEEK_END);
    const long left = std::ftell(f) - at;                          // (a header must not make us allocate what the file cannot hold)
    std::fseek(f, at, SEEK_SET);
    if (at < 0 || left < 0 || (unsigned long long)left < (unsigned long long)im.w * im.h * im.ch) { std::fclose(f); return false; }
    im.px.resize((size_t)im.w * im.h * im.ch);
    const bool ok = std::fread(im.px.data(), 1, im.px.size(), f) == im.px.size();
    std::fclose(f);
    return ok;
}

static bool write_pnm(const std::string &path, int w, int h, int ch, const unsigned char *px) {
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    std::fprintf(f, "%s\n%d %d\n255\n", ch == 3 ? "P6" : "P5", w, h);
    const bool ok = std::fwrite(px, 1, (size_t)w * h * ch, f) == (size_t)w * h * ch;
    std::fclose(f);
    return ok;
}

// ---- PNG (the dataset's annotations are PNG, src/main.cpp:158-161; the reference saves PNG, :312-316) on zlib alone -----------
// Reader: 8-bit, non-interlaced, gray / gray+alpha / RGB / RGBA / palette; alpha is dropped.  Writer: 8-bit gray or RGB, filter 0.
static uint32_t be32(const unsigned char *p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

static bool read_png(const std::string &path, Pnm &im) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::vector<unsigned char> file;
    unsigned char buf[65536]; size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + n);
    std::fclose(f);
    static const unsigned char sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (file.size() < 8 || std::memcmp(file.data(), sig, 8)) return false;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<unsigned char> idat, plte;
    for (size_t p = 8; p + 12 <= file.size();) {
        const uint32_t len = be32(&file[p]);
        if (p + 12 + len > file.size()) return false;
        const char *type = (const char *)&file[p + 4];
        const unsigned char *data = &file[p + 8];
        if (!std::memcmp(type, "IHDR", 4) && len >= 13) { im.w = (int)be32(data); im.h = (int)be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12]; }
        else if (!std::memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!std::memcmp(type, "IEND", 4)) break;
        p += 12 + len;
    }
    const int spp = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;      // samples per pixel in the file
    if (depth != 8 || !spp || interlace || im.w <= 0 || im.h <= 0 || im.w > 65535 || im.h > 65535) return false;
    const size_t stride = (size_t)im.w * spp;
    // (deflate expands by at most ~1032 : 1: a header that promises more than its IDAT can hold is refused before anything is allocated)
    if ((unsigned long long)(stride + 1) * im.h > (unsigned long long)idat.size() * 1100ull + 65536ull) return false;
    std::vector<unsigned char> raw((stride + 1) * im.h);
    uLongf rawlen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK || rawlen != raw.size()) return false;
    std::vector<unsigned char> prev(stride, 0), cur(stride);
    im.ch = (ctype == 0 || ctype == 4) ? 1 : 3;
    im.px.resize((size_t)im.w * im.h * im.ch);
    for (int y = 0; y < im.h; y++) {
        const unsigned char *line = &raw[(stride + 1) * y];
        const int ft = line[0];
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= (size_t)spp ? cur[i - spp] : 0, b = prev[i], c = i >= (size_t)spp ? prev[i - spp] : 0;
            int pred = 0;
            if (ft == 1) pred = a; else if (ft == 2) pred = b; else if (ft == 3) pred = (a + b) >> 1;
            else if (ft == 4) { const int pp = a + b - c, pa = std::abs(pp - a), pb = std::abs(pp - b), pc = std::abs(pp - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
            else if (ft != 0) return false;
            cur[i] = (unsigned char)(line[1 + i] + pred);
        }
        for (int x = 0; x < im.w; x++) {
            unsigned char *o = &im.px[((size_t)y * im.w + x) * im.ch];
            const unsigned char *s = &cur[(size_t)x * spp];
            if (ctype == 0 || ctype == 4) o[0] = s[0];
            else if (ctype == 3) { if ((size_t)s[0] * 3 + 2 >= plte.size()) return false; o[0] = plte[s[0] * 3]; o[1] = plte[s[0] * 3 + 1]; o[2] = plte[s[0] * 3 + 2]; }
            else { o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; }
        }
        prev.swap(cur);
    }
    return true;
}

static bool write_png(const std::string &path, int w, int h, int ch, const unsigned char *px) {
    std::vector<unsigned char> raw(((size_t)w * ch + 1) * h);
    for (int y = 0; y < h; y++) { raw[((size_t)w * ch + 1) * y] = 0; std::memcpy(&raw[((size_t)w * ch + 1) * y + 1], px + (size_t)y * w * ch, (size_t)w * ch); }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<unsigned char> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    auto chunk = [&](const char *type, const unsigned char *data, uint32_t len) {
        unsigned char hdr[8] = {(unsigned char)(len >> 24), (unsigned char)(len >> 16), (unsigned char)(len >> 8), (unsigned char)len, (unsigned char)type[0], (unsigned char)type[1], (unsigned char)type[2], (unsigned char)type[3]};
        uLong crc = crc32(0L, hdr + 4, 4);
        if (len) crc = crc32(crc, data, len);
        const unsigned char tail[4] = {(unsigned char)(crc >> 24), (unsigned char)(crc >> 16), (unsigned char)(crc >> 8), (unsigned char)crc};
        std::fwrite(hdr, 1, 8, f); if (len) std::fwrite(data, 1, len, f); std::fwrite(tail, 1, 4, f);
    };
    static const unsigned char sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    std::fwrite(sig, 1, 8, f);
    const unsigned char ihdr[13] = {(unsigned char)(w >> 24), (unsigned char)(w >> 16), (unsigned char)(w >> 8), (unsigned char)w, (unsigned char)(h >> 24), (unsigned char)(h >> 16), (unsigned char)(h >> 8), (unsigned char)h,
                                    8, (unsigned char)(ch == 3 ? 2 : 0), 0, 0, 0};
    chunk("IHDR", ihdr, 13); chunk("IDAT", comp.data(), (uint32_t)clen); chunk("IEND", nullptr, 0);
    return std::fclose(f) == 0;
}

static bool ends_with(const std::string &s, const char *suffix) { const size_t n = std::strlen(suffix); return s.size() >= n && !s.compare(s.size() - n, n, suffix); }
static bool read_jpeg(const std::string &path, Pnm &im) {           // (cv::imread's decode, restated: harness/jpeg_reader.hpp)
    std::string why;
    if (rtdd_jpeg::read_file(path, im.w, im.h, im.ch, im.px, &why)) return true;
    std::fprintf(stderr, "%s: %s\n", path.c_str(), why.c_str());
    return false;
}
static bool read_image(const std::string &path, Pnm &im) {
    if (ends_with(path, ".png")) return read_png(path, im);
    if (ends_with(path, ".jpg") || ends_with(path, ".jpeg") || ends_with(path, ".JPG") || ends_with(path, ".JPEG")) return read_jpeg(path, im);
    return read_pnm(path, im);
}
static bool write_image(const std::string &path, int w, int h, int ch, const unsigned char *px) { return ends_with(path, ".png") ? write_png(path, w, h, ch, px) : write_pnm(path, w, h, ch, px); }

#define CK(call) do { int rc_ = (call); if (rc_ != RTDD_OK) { std::printf("%s: %s (%s)\n", #call, rtdd_status_string(rc_), rtdd_last_error(ctx)); return rc_; } } while (0)

struct Paint { int x, y, label, radius, frame; };      // frame: --paint-at (live mode: in front of that frame); --paint: before the first estimate
struct Job {
    Pnm bgr, ann;                 // bgr is BGR-interleaved like cv::imread's Mat
    bool has_ann = false;
    std::vector<Paint> paints, live_paints;
    std::string effect;
    int iters = 1000;
    std::string refine;           // "" | "sor" | "mg": rtdd_refine_depth after every estimate
    float tolerance = 1e-4f;
    bool sequential = false;      // --sequential: a --batch as one estimate after the other (default: rtdd_estimate_depth_batch, all images in the same launches)
    bool cold = false;            // --cold: no warm-up: the first (and, without --live, only) estimate pays the one-time costs
};

// What a process pays ONCE on a device -- the runtime's first stream and allocations, every kernel's code object on its first launch,
// ~20 ms together -- and what "Processing Time" is not about (the reference's first key press pays CUDA's likewise): a 96 x 128 estimate
// and each effect on a context of its own, destroyed again.  Nothing of the job's is touched; --cold leaves it out.
static void warm_up(int device) {
    rtdd_ctx *w = nullptr;
    if (rtdd_ctx_create(device, &w) != RTDD_OK) return;
    const int rows = 96, cols = 128;
    std::vector<unsigned char> bgr((size_t)rows * cols * 3, 90), ann((size_t)rows * cols, 32);
    for (int x = 20; x < 40; x++) { ann[(size_t)30 * cols + x] = 254; ann[(size_t)70 * cols + x] = 0; }
    unsigned char *d_bgr = nullptr, *d_ann = nullptr;
    if (hipSetDevice(device) == hipSuccess && hipMalloc((void **)&d_bgr, bgr.size()) == hipSuccess && hipMalloc((void **)&d_ann, ann.size()) == hipSuccess &&
        rtdd_load_weights(w, 0.4f) == RTDD_OK && rtdd_pyramid_create(w, rows, cols) == RTDD_OK &&
        rtdd_upload(w, d_bgr, (size_t)cols * 3, bgr.data(), (size_t)cols * 3, (size_t)cols * 3, rows) == RTDD_OK && rtdd_pyramid_set_image(w, d_bgr, (size_t)cols * 3) == RTDD_OK &&
        rtdd_upload(w, d_ann, cols, ann.data(), cols, cols, rows) == RTDD_OK && rtdd_pyramid_set_annotation(w, d_ann, cols) == RTDD_OK && rtdd_estimate_depth(w, 1000) == RTDD_OK) {
        void *po, *pg, *pd, *pa; size_t io, ig, id, ia;
        if (rtdd_pyramid_image(w, RTDD_IMG_ORIGINAL, 0, &po, &io, nullptr, nullptr) == RTDD_OK && rtdd_pyramid_image(w, RTDD_IMG_GRAY, 0, &pg, &ig, nullptr, nullptr) == RTDD_OK &&
            rtdd_pyramid_image(w, RTDD_IMG_DEPTH, 0, &pd, &id, nullptr, nullptr) == RTDD_OK && rtdd_pyramid_image(w, RTDD_IMG_ARTISTIC, 0, &pa, &ia, nullptr, nullptr) == RTDD_OK) {
            (void)rtdd_simulate_defocus(w, (const uint8_t *)po, io, (const float *)pd, id, (uint8_t *)pa, ia, rows, cols);
            (void)rtdd_simulate_desaturation(w, (const uint8_t *)po, io, (const uint8_t *)pg, ig, (const float *)pd, id, (uint8_t *)pa, ia, rows, cols);
            (void)rtdd_simulate_haze(w, (const uint8_t *)po, io, (const float *)pd, id, (uint8_t *)pa, ia, rows, cols);
        }
        std::vector<unsigned char> back((size_t)rows * cols);
        void *pu; size_t iu;
        if (rtdd_pyramid_image(w, RTDD_IMG_DEPTH_U8, 0, &pu, &iu, nullptr, nullptr) == RTDD_OK) (void)rtdd_download(w, back.data(), cols, pu, iu, cols, rows);
    }
    (void)rtdd_ctx_synchronize(w);
    if (d_bgr) (void)hipFree(d_bgr);
    if (d_ann) (void)hipFree(d_ann);
    (void)rtdd_ctx_destroy(w);
}

// One GPU: context + stream + device staging, runs `count` estimates; keeps the last result on the host.
static int run_device(int device, const Job &job, int count, bool live, std::vector<unsigned char> *depth_u8, std::vector<unsigned char> *art, double *ms_per_estimate,
                      std::vector<std::vector<unsigned char>> *every_map = nullptr, std::vector<unsigned char> *annotated = nullptr) {
    // everything this function owns, released on EVERY return path (the CK() early returns included)
    struct Owned {
        rtdd_ctx *ctx = nullptr; hipStream_t stream = nullptr; unsigned char *d_bgr = nullptr, *d_ann = nullptr;
        ~Owned() {
            if (d_bgr) (void)hipFree(d_bgr);
            if (d_ann) (void)hipFree(d_ann);
            if (ctx) rtdd_ctx_destroy(ctx);           // synchronises the stream first
            if (stream) (void)hipStreamDestroy(stream);
        }
    } own;
    if (!job.cold) warm_up(device);
    int rc = rtdd_ctx_create(device, &own.ctx);
    if (rc != RTDD_OK) { std::printf("rtdd_ctx_create(%d): %s\n", device, rtdd_status_string(rc)); return rc; }
    rtdd_ctx *ctx = own.ctx;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&own.stream) != hipSuccess) { std::printf("device %d: cannot create a stream\n", device); return RTDD_ERR_HIP; }
    rtdd_ctx_set_stream(ctx, own.stream);
    const int rows = job.bgr.h, cols = job.bgr.w;
    CK(rtdd_load_weights(ctx, 0.4f));                                   // main.cpp:152-155
    // --batch B without an effect or a refinement: this device's images as ONE batched pyramid -- every level of all of them in the same
    // launches (rtdd_estimate_depth_batch: the coarse levels of one 1080p image use a quarter of the chip)
    const bool batched = !live && count > 1 && !job.sequential && job.effect.empty() && job.refine.empty();
    if (batched) {
        CK(rtdd_pyramid_create_batch(ctx, rows, cols, count));
        if (hipMalloc((void **)&own.d_bgr, (size_t)rows * cols * 3) != hipSuccess || hipMalloc((void **)&own.d_ann, (size_t)rows * cols) != hipSuccess) { std::printf("device %d: out of memory\n", device); return RTDD_ERR_NOMEM; }
        auto t0 = std::chrono::steady_clock::now();
        for (int n = 0; n < count; n++) {                               // every image is independent: its own upload, annotation and strokes
            void *ps, *pe; size_t pis, pie;
            CK(rtdd_pyramid_select(ctx, n));
            CK(rtdd_upload(ctx, own.d_bgr, (size_t)cols * 3, job.bgr.px.data(), (size_t)cols * 3, (size_t)cols * 3, rows));
            CK(rtdd_pyramid_set_image(ctx, own.d_bgr, (size_t)cols * 3));
            if (job.has_ann) {
                CK(rtdd_upload(ctx, own.d_ann, cols, job.ann.px.data(), cols, cols, rows));
                CK(rtdd_pyramid_set_annotation(ctx, own.d_ann, cols));
            }
            CK(rtdd_pyramid_image(ctx, RTDD_IMG_SCRIBBLE, 0, &ps, &pis, nullptr, nullptr));
            CK(rtdd_pyramid_image(ctx, RTDD_IMG_EDITED, 0, &pe, &pie, nullptr, nullptr));
            for (const Paint &p : job.paints) CK(rtdd_paint_image(ctx, p.x, p.y, p.label, p.radius, (uint8_t *)pe, pie, (uint8_t *)ps, pis, rows, cols));
        }
        CK(rtdd_estimate_depth_batch(ctx, job.iters));                  // main.cpp:239-291, for every image
        depth_u8->resize((size_t)rows * cols);
        for (int n = 0; n < count; n++) {
            void *pu; size_t piu;
            CK(rtdd_pyramid_select(ctx, n));
            CK(rtdd_pyramid_image(ctx, RTDD_IMG_DEPTH_U8, 0, &pu, &piu, nullptr, nullptr));
            CK(rtdd_download(ctx, depth_u8->data(), cols, pu, piu, cols, rows));      // main.cpp:291 (synchronises)
            if (every_map) every_map->push_back(*depth_u8);
        }
        *ms_per_estimate = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / count;
        if (annotated) {
            void *pe; size_t pie;
            CK(rtdd_pyramid_select(ctx, 0));
            CK(rtdd_pyramid_image(ctx, RTDD_IMG_EDITED, 0, &pe, &pie, nullptr, nullptr));
            annotated->resize((size_t)rows * cols * 3); CK(rtdd_download(ctx, annotated->data(), (size_t)cols * 3, pe, pie, (size_t)cols * 3, rows));
        }
        CK(rtdd_ctx_synchronize(ctx));
        return RTDD_OK;
    }
    CK(rtdd_pyramid_create(ctx, rows, cols));                           // main.cpp:92-149
    if (hipMalloc((void **)&own.d_bgr, (size_t)rows * cols * 3) != hipSuccess || hipMalloc((void **)&own.d_ann, (size_t)rows * cols) != hipSuccess) { std::printf("device %d: out of memory\n", device); return RTDD_ERR_NOMEM; }
    unsigned char *d_bgr = own.d_bgr, *d_ann = own.d_ann;
    void *p_scr, *p_ed, *p_orig, *p_gray, *p_depth, *p_art, *p_u8;
    size_t pi_scr, pi_ed, pi_orig, pi_gray, pi_depth, pi_art, pi_u8;
    CK(rtdd_pyramid_image(ctx, RTDD_IMG_SCRIBBLE, 0, &p_scr, &pi_scr, nullptr, nullptr));
    CK(rtdd_pyramid_image(ctx, RTDD_IMG_EDITED, 0, &p_ed, &pi_ed, nullptr, nullptr));
    CK(rtdd_pyramid_image(ctx, RTDD_IMG_ORIGINAL, 0, &p_orig, &pi_orig, nullptr, nullptr));
    CK(rtdd_pyramid_image(ctx, RTDD_IMG_GRAY, 0, &p_gray, &pi_gray, nullptr, nullptr));
    CK(rtdd_pyramid_image(ctx, RTDD_IMG_DEPTH, 0, &p_depth, &pi_depth, nullptr, nullptr));
    CK(rtdd_pyramid_image(ctx, RTDD_IMG_ARTISTIC, 0, &p_art, &pi_art, nullptr, nullptr));
    CK(rtdd_pyramid_image(ctx, RTDD_IMG_DEPTH_U8, 0, &p_u8, &pi_u8, nullptr, nullptr));

    // --live N: the reference's frame loop as it clocks it (main.cpp:232-295) -- every frame uploads the host's scribble and edited
    // images (:236-237), estimates, downloads the u8 map (:290-291) -- two frames in flight (rtdd_live_submit_ex): the copies of one frame
    // overlap the arithmetic of the other.  With --effect X every frame also renders the sticky effect from its own depth map and brings
    // the artistic image to the host (main.cpp:190-230 runs in every iteration of the loop once a key has switched the effect on).
    if (live && job.refine.empty()) {
        struct Pinned { void *p = nullptr; ~Pinned() { if (p) rtdd_host_free(p); } } h_scr, h_ed, h_u8[2], h_art[2];
        const int fx = job.effect == "defocus" ? RTDD_EFFECT_DEFOCUS : job.effect == "desaturation" ? RTDD_EFFECT_DESATURATION : job.effect == "haze" ? RTDD_EFFECT_HAZE : RTDD_EFFECT_NONE;
        if (fx != RTDD_EFFECT_NONE) { CK(rtdd_host_alloc(&h_art[0].p, (size_t)rows * cols * 3)); CK(rtdd_host_alloc(&h_art[1].p, (size_t)rows * cols * 3)); art->resize((size_t)rows * cols * 3); }
        CK(rtdd_upload(ctx, d_bgr, (size_t)cols * 3, job.bgr.px.data(), (size_t)cols * 3, (size_t)cols * 3, rows));
        CK(rtdd_pyramid_set_image(ctx, d_bgr, (size_t)cols * 3));
        if (job.has_ann) {
            CK(rtdd_upload(ctx, d_ann, cols, job.ann.px.data(), cols, cols, rows));
            CK(rtdd_pyramid_set_annotation(ctx, d_ann, cols));
        }
        for (const Paint &p : job.paints)
            CK(rtdd_paint_image(ctx, p.x, p.y, p.label, p.radius, (uint8_t *)p_ed, pi_ed, (uint8_t *)p_scr, pi_scr, rows, cols));
        CK(rtdd_host_alloc(&h_scr.p, (size_t)rows * cols)); CK(rtdd_host_alloc(&h_ed.p, (size_t)rows * cols * 3));
        CK(rtdd_host_alloc(&h_u8[0].p, (size_t)rows * cols)); CK(rtdd_host_alloc(&h_u8[1].p, (size_t)rows * cols));
        CK(rtdd_download(ctx, h_scr.p, cols, p_scr, pi_scr, cols, rows));            // the host's Mats (main.cpp:160-168: decoded on the host there)
        CK(rtdd_download(ctx, h_ed.p, (size_t)cols * 3, p_ed, pi_ed, (size_t)cols * 3, rows));
        depth_u8->resize((size_t)rows * cols);
        auto landed = [&](int f) {
            std::memcpy(depth_u8->data(), h_u8[f % 2].p, depth_u8->size());
            if (fx != RTDD_EFFECT_NONE) std::memcpy(art->data(), h_art[f % 2].p, art->size());
            if (every_map) every_map->push_back(*depth_u8);
        };
        auto t0 = std::chrono::steady_clock::now();
        for (int n = 0; n < count; n++) {
            bool painted = false;
            for (const Paint &p : job.live_paints)
                if (p.frame == n) {
                    // main.cpp:46-62: the mouse callback paints the DEVICE images and downloads them into the host's Mats, which the next
                    // estimate uploads again (:236-237).  The frames in flight are let land first: they read the images being painted.
                    if (!painted) {
                        while (rtdd_live_pending(ctx) > 0) { const int f = n - rtdd_live_pending(ctx); CK(rtdd_live_wait(ctx)); landed(f); }
                        // (an uploaded annotation pair becomes the pyramid's level-0 images: ask where they are now, include/rtdd.h)
                        CK(rtdd_pyramid_image(ctx, RTDD_IMG_SCRIBBLE, 0, &p_scr, &pi_scr, nullptr, nullptr));
                        CK(rtdd_pyramid_image(ctx, RTDD_IMG_EDITED, 0, &p_ed, &pi_ed, nullptr, nullptr));
                    }
                    CK(rtdd_paint_image(ctx, p.x, p.y, p.label, p.radius, (uint8_t *)p_ed, pi_ed, (uint8_t *)p_scr, pi_scr, rows, cols));
                    painted = true;
                }
            if (painted) {
                CK(rtdd_download(ctx, h_scr.p, cols, p_scr, pi_scr, cols, rows));
                CK(rtdd_download(ctx, h_ed.p, (size_t)cols * 3, p_ed, pi_ed, (size_t)cols * 3, rows));
            }
            if (rtdd_live_pending(ctx) >= 2) { const int f = n - 2; CK(rtdd_live_wait(ctx)); landed(f); }              // frame n-2's buffer is about to be reused
            CK(rtdd_live_submit_ex(ctx, (const uint8_t *)h_scr.p, cols, (const uint8_t *)h_ed.p, (size_t)cols * 3, job.iters, (uint8_t *)h_u8[n % 2].p, cols,
                                   fx, (uint8_t *)h_art[n % 2].p, (size_t)cols * 3));
        }
        while (rtdd_live_pending(ctx) > 0) { const int f = count - rtdd_live_pending(ctx); CK(rtdd_live_wait(ctx)); landed(f); }
        *ms_per_estimate = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / (count > 0 ? count : 1);
        CK(rtdd_pyramid_image(ctx, RTDD_IMG_EDITED, 0, &p_ed, &pi_ed, nullptr, nullptr));
        if (annotated) { annotated->resize((size_t)rows * cols * 3); CK(rtdd_download(ctx, annotated->data(), (size_t)cols * 3, p_ed, pi_ed, (size_t)cols * 3, rows)); }
        CK(rtdd_ctx_synchronize(ctx));
        return RTDD_OK;
    }

    auto t0 = std::chrono::steady_clock::now();
    for (int n = 0; n < count; n++) {
        if (n == 0 || !live) {                                          // a new image (--batch: every image is independent -- rtdd_pyramid_set_image
                                                                        // also resets the warm-start state): upload + (re)build the pyramid inputs
            CK(rtdd_upload(ctx, d_bgr, (size_t)cols * 3, job.bgr.px.data(), (size_t)cols * 3, (size_t)cols * 3, rows));
            CK(rtdd_pyramid_set_image(ctx, d_bgr, (size_t)cols * 3));
            if (job.has_ann) {
                CK(rtdd_upload(ctx, d_ann, cols, job.ann.px.data(), cols, cols, rows));
                CK(rtdd_pyramid_set_annotation(ctx, d_ann, cols));
            }
            for (const Paint &p : job.paints)                           // main.cpp:55-57
                CK(rtdd_paint_image(ctx, p.x, p.y, p.label, p.radius, (uint8_t *)p_ed, pi_ed, (uint8_t *)p_scr, pi_scr, rows, cols));
        }
        CK(rtdd_estimate_depth(ctx, job.iters));                        // main.cpp:239-291
        if (!job.refine.empty()) {                                      // extension: converge the finest level
            rtdd_solve_params sp;
            sp.method = job.refine == "mg" ? RTDD_METHOD_MULTIGRID : job.refine == "auto" ? RTDD_METHOD_AUTO : RTDD_METHOD_RED_BLACK_GS;
            sp.maxIterations = job.refine == "mg" ? 100 : 400000; sp.tolerance = job.tolerance; sp.checkEvery = 0; sp.relaxation = RTDD_RELAXATION_AUTO;
            rtdd_solve_info si;
            CK(rtdd_refine_depth(ctx, &sp, &si));
            if (n == 0 && job.refine == "auto") std::printf("refine auto: %d cycles + %d sweeps, residual %g\n", si.cycles, si.iterations, si.residual);
            else if (n == 0) std::printf("refine %s: %d %s, residual %g\n", job.refine.c_str(), si.iterations, job.refine == "mg" ? "cycles" : "sweeps", si.residual);
        }
        if (job.effect == "defocus") CK(rtdd_simulate_defocus(ctx, (uint8_t *)p_orig, pi_orig, (float *)p_depth, pi_depth, (uint8_t *)p_art, pi_art, rows, cols));
        else if (job.effect == "desaturation") CK(rtdd_simulate_desaturation(ctx, (uint8_t *)p_orig, pi_orig, (uint8_t *)p_gray, pi_gray, (float *)p_depth, pi_depth, (uint8_t *)p_art, pi_art, rows, cols));
        else if (job.effect == "haze") CK(rtdd_simulate_haze(ctx, (uint8_t *)p_orig, pi_orig, (float *)p_depth, pi_depth, (uint8_t *)p_art, pi_art, rows, cols));
        depth_u8->resize((size_t)rows * cols);
        CK(rtdd_download(ctx, depth_u8->data(), cols, p_u8, pi_u8, cols, rows));   // main.cpp:291 (synchronises)
        if (!job.effect.empty()) { art->resize((size_t)rows * cols * 3); CK(rtdd_download(ctx, art->data(), (size_t)cols * 3, p_art, pi_art, (size_t)cols * 3, rows)); }
        if (every_map) every_map->push_back(*depth_u8);                 // --write-all: the n-th estimate of this device
    }
    *ms_per_estimate = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / (count > 0 ? count : 1);
    if (annotated) { annotated->resize((size_t)rows * cols * 3); CK(rtdd_download(ctx, annotated->data(), (size_t)cols * 3, p_ed, pi_ed, (size_t)cols * 3, rows)); }   // editedImage[0], main.cpp:298-303
    CK(rtdd_ctx_synchronize(ctx));                                      // (a persistent launch that gave up has been healed by now: include/rtdd.h RTDD_ERR_TIMEOUT)
    return RTDD_OK;
}

int main(int argc, const char *argv[]) {
    if (argc == 1) { std::printf("Usage: rtdd_harness -i image.(jpg|png|ppm) [-a annotation.(png|pgm)] [-o prefix] [--effect defocus|desaturation|haze] [--iters N] [--refine sor|mg|auto [--tolerance T]]\n"
                                 "                    [--paint x,y,label,radius]... [--live N [--paint-at frame:x,y,label,radius]...] [--devices D --batch B [--sequential] [--write-all]] [--png] [--cold]\n"
                                 "       rtdd_harness --convert in.(jpg|png|ppm|pgm) out.(png|ppm|pgm)   (JPEG / 8-bit PNG / PNM -> PNG / PNM, no GPU)\n"); return 0; }
    if (argc == 4 && !std::strcmp(argv[1], "--convert")) {               // file format conversion only (no GPU): JPEG / PNG / PNM -> PNG / PNM
        Pnm im;
        if (!read_image(argv[2], im)) { std::printf("cannot read %s\n", argv[2]); return 2; }
        return write_image(argv[3], im.w, im.h, im.ch, im.px.data()) ? 0 : 5;
    }
    Job job;
    std::string in, an, out = "";
    bool png = false, write_all = false;
    int devices = 1, batch = 1, live = 0;
    for (int i = 1; i < argc; i++) {
        auto next = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
        if (!std::strcmp(argv[i], "-i")) in = next();
        else if (!std::strcmp(argv[i], "-a")) an = next();
        else if (!std::strcmp(argv[i], "-o")) out = next();
        else if (!std::strcmp(argv[i], "--effect")) job.effect = next();
        else if (!std::strcmp(argv[i], "--cold")) job.cold = true;
        else if (!std::strcmp(argv[i], "--iters")) job.iters = std::atoi(next());
        else if (!std::strcmp(argv[i], "--refine")) job.refine = next();          // sor | mg | auto
        else if (!std::strcmp(argv[i], "--tolerance")) job.tolerance = (float)std::atof(next());
        else if (!std::strcmp(argv[i], "--devices")) devices = std::atoi(next());
        else if (!std::strcmp(argv[i], "--batch")) batch = std::atoi(next());
        else if (!std::strcmp(argv[i], "--live")) live = std::atoi(next());
        else if (!std::strcmp(argv[i], "--sequential")) job.sequential = true;      // a --batch one estimate after the other (the round-4 behaviour)
        else if (!std::strcmp(argv[i], "--write-all")) write_all = true;           // every estimate of a --batch: <out>DepthMap_<b>.pgm|png
        else if (!std::strcmp(argv[i], "--png")) png = true;                       // DepthMap.png / ArtisticEffect.png like the reference
        else if (!std::strcmp(argv[i], "--paint")) { Paint p{0, 0, 0, 0, -1}; if (std::sscanf(next(), "%d,%d,%d,%d", &p.x, &p.y, &p.label, &p.radius) == 4) job.paints.push_back(p); }
        else if (!std::strcmp(argv[i], "--paint-at")) { Paint p{0, 0, 0, 0, 0}; if (std::sscanf(next(), "%d:%d,%d,%d,%d", &p.frame, &p.x, &p.y, &p.label, &p.radius) == 5) job.live_paints.push_back(p); }
        else if (!std::strcmp(argv[i], "-h")) std::printf("Usage:\n -i input image (JPEG, 8-bit PNG, binary PPM)\n -a annotated image (8-bit PNG, binary PGM)\n");
    }
    Pnm rgb;
    if (!read_image(in, rgb)) { std::printf("cannot read %s as a binary PPM / PGM, an 8-bit PNG or a JPEG\n", in.c_str()); return 2; }
    if (rgb.ch == 1) {                                                       // cv::imread's default flag gives three channels whatever the file holds
        Pnm c; c.w = rgb.w; c.h = rgb.h; c.ch = 3; c.px.resize(rgb.px.size() * 3);
        for (size_t i = 0; i < rgb.px.size(); i++) c.px[3 * i] = c.px[3 * i + 1] = c.px[3 * i + 2] = rgb.px[i];
        rgb = c;
    }
    job.bgr = rgb;
    for (size_t i = 0; i < rgb.px.size(); i += 3) { job.bgr.px[i] = rgb.px[i + 2]; job.bgr.px[i + 2] = rgb.px[i]; }   // cv::imread gives BGR
    if (!an.empty()) {
        if (!read_image(an, job.ann) || job.ann.w != rgb.w || job.ann.h != rgb.h) { std::printf("cannot read %s as a binary PGM / 8-bit PNG of the image's size\n", an.c_str()); return 2; }
        if (job.ann.ch == 3) {                                               // cv::imread(path, 0) of a colour file: BT.601 gray, as for the image
            Pnm g1; g1.w = job.ann.w; g1.h = job.ann.h; g1.ch = 1; g1.px.resize((size_t)g1.w * g1.h);
            for (size_t i = 0; i < g1.px.size(); i++) g1.px[i] = (unsigned char)((job.ann.px[3 * i] * 4899 + job.ann.px[3 * i + 1] * 9617 + job.ann.px[3 * i + 2] * 1868 + 8192) >> 14);
            job.ann = g1;
        }
        job.has_ann = true;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { std::printf("no HIP device: %s\n", rtdd_status_string(RTDD_ERR_NO_DEVICE)); return 3; }
    if (devices > ndev) devices = ndev;

    std::vector<std::vector<unsigned char>> depth(devices), art(devices), annotated(devices);
    std::vector<std::vector<std::vector<unsigned char>>> every(devices);   // [device][n-th estimate on it]: image b = n * devices + device
    std::vector<double> ms(devices, 0.0);
    std::vector<int> rcs(devices, 0), counts(devices, 0);
    for (int b = 0; b < batch; b++) counts[b % devices]++;               // image b -> device b % D
    if (live > 0) { counts.assign(devices, 0); counts[0] = live; }
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int d = 0; d < devices; d++)
        th.emplace_back([&, d]() { rcs[d] = counts[d] ? run_device(d, job, counts[d], live > 0, &depth[d], &art[d], &ms[d], write_all ? &every[d] : nullptr, d == 0 ? &annotated[0] : nullptr) : 0; });
    for (auto &t : th) t.join();
    const double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (int d = 0; d < devices; d++) if (rcs[d]) return 4;

    const int total = live > 0 ? live : batch;
    // (the reference's clock() brackets the FIRST estimate of the process, first-call costs included: that figure is --cold's; the default
    // runs a warm-up estimate and the three effects on a throw-away context first, so that what is printed is the steady cost)
    std::printf("Processing Time: %.3f ms per estimate on device 0 (upload + estimate%s + download; %s); %d estimate(s) on %d device(s) in %.1f ms wall incl. setup\n",
                ms[0], job.effect.empty() ? "" : " + effect",
                job.cold ? "COLD: the process's first estimate, first-call costs included, like the reference's clock()" : "WARM: after a warm-up on a throw-away context; --cold for the first-call figure",
                total, devices, wall);
    if (live > 0 && job.refine.empty())
        std::printf("Live: %.1f frames/s (%d frames, annotation upload + estimate%s + map download per frame, two frames in flight)\n", 1e3 / (ms[0] > 0 ? ms[0] : 1), live,
                    job.effect.empty() ? "" : " + sticky effect and its image's download");
    if (!write_image(out + (png ? "DepthMap.png" : "DepthMap.pgm"), rgb.w, rgb.h, 1, depth[0].data())) { std::printf("cannot write %sDepthMap\n", out.c_str()); return 5; }
    {                                                                    // main.cpp:298-303: editedImage[0] -- the image with the scribbles in it
        std::vector<unsigned char> o(annotated[0]);
        for (size_t i = 0; i + 2 < o.size(); i += 3) { o[i] = annotated[0][i + 2]; o[i + 2] = annotated[0][i]; }     // BGR -> RGB for the file
        if (!o.empty() && !write_image(out + (png ? "AnnotatedImage.png" : "AnnotatedImage.ppm"), rgb.w, rgb.h, 3, o.data())) return 5;
    }
    if (!job.effect.empty()) {
        std::vector<unsigned char> o(art[0]);
        for (size_t i = 0; i < o.size(); i += 3) { o[i] = art[0][i + 2]; o[i + 2] = art[0][i]; }     // BGR -> RGB for the file
        if (!write_image(out + (png ? "ArtisticEffect.png" : "ArtisticEffect.ppm"), rgb.w, rgb.h, 3, o.data())) return 5;
    }
    if (write_all)
        for (int d = 0; d < devices; d++)
            for (size_t n = 0; n < every[d].size(); n++) {
                const std::string name = out + "DepthMap_" + std::to_string(n * devices + d) + (png ? ".png" : ".pgm");
                if (!write_image(name, rgb.w, rgb.h, 1, every[d][n].data())) { std::printf("cannot write %s\n", name.c_str()); return 5; }
            }
    std::printf("Saving images...\n");                                   // main.cpp:317
    return 0;
}
