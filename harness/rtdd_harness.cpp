// rtdd_harness -- headless stand-in for the reference's interactive shell (/root/reference/src/main.cpp).
//
// main.cpp is an OpenCV HighGUI event loop over cv::cuda::GpuMat and cannot be built on a ROCm box
// (SURVEY.md section 0); this program makes the SAME sequence of calls through librtdd.so's C ABI
// with plain files instead of windows:
//     -i image.ppm  -a annotation.pgm      (main.cpp:81-90; binary PPM/PGM instead of JPEG/PNG)
//     key 'd'  -> one depth estimate        (main.cpp:232-295)  -> <out>DepthMap.pgm     (main.cpp:306-310)
//     key 'b'/'g'/'h' -> --effect defocus|desaturation|haze     -> <out>ArtisticEffect.ppm (main.cpp:190-230, 312-316)
//     key 't'  -> prints "Processing Time"  (main.cpp:320-322; wall clock here, the reference uses clock())
//     --paint x,y,label,radius  = a mouse drag sample (main.cpp:46-62), repeatable
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

#include "rtdd.h"

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

#define CK(call) do { int rc_ = (call); if (rc_ != RTDD_OK) { std::printf("%s: %s (%s)\n", #call, rtdd_status_string(rc_), rtdd_last_error(ctx)); return rc_; } } while (0)

struct Paint { int x, y, label, radius; };
struct Job {
    Pnm bgr, ann;                 // bgr is BGR-interleaved like cv::imread's Mat
    bool has_ann = false;
    std::vector<Paint> paints;
    std::string effect;
    int iters = 1000;
    std::string refine;           // "" | "sor" | "mg": rtdd_refine_depth after every estimate
    float tolerance = 1e-4f;
};

// One GPU: context + stream + device staging, runs `count` estimates; keeps the last result on the host.
static int run_device(int device, const Job &job, int count, bool live, std::vector<unsigned char> *depth_u8, std::vector<unsigned char> *art, double *ms_per_estimate) {
    rtdd_ctx *ctx = nullptr;
    int rc = rtdd_ctx_create(device, &ctx);
    if (rc != RTDD_OK) { std::printf("rtdd_ctx_create(%d): %s\n", device, rtdd_status_string(rc)); return rc; }
    hipStream_t stream = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&stream) != hipSuccess) { std::printf("device %d: cannot create a stream\n", device); return RTDD_ERR_HIP; }
    rtdd_ctx_set_stream(ctx, stream);
    const int rows = job.bgr.h, cols = job.bgr.w;
    CK(rtdd_load_weights(ctx, 0.4f));                                   // main.cpp:152-155
    CK(rtdd_pyramid_create(ctx, rows, cols));                           // main.cpp:92-149
    unsigned char *d_bgr = nullptr, *d_ann = nullptr;
    if (hipMalloc((void **)&d_bgr, (size_t)rows * cols * 3) != hipSuccess || hipMalloc((void **)&d_ann, (size_t)rows * cols) != hipSuccess) { std::printf("device %d: out of memory\n", device); return RTDD_ERR_NOMEM; }
    void *p_scr, *p_ed, *p_orig, *p_gray, *p_depth, *p_art, *p_u8;
    size_t pi_scr, pi_ed, pi_orig, pi_gray, pi_depth, pi_art, pi_u8;
    CK(rtdd_pyramid_image(ctx, RTDD_IMG_SCRIBBLE, 0, &p_scr, &pi_scr, nullptr, nullptr));
    CK(rtdd_pyramid_image(ctx, RTDD_IMG_EDITED, 0, &p_ed, &pi_ed, nullptr, nullptr));
    CK(rtdd_pyramid_image(ctx, RTDD_IMG_ORIGINAL, 0, &p_orig, &pi_orig, nullptr, nullptr));
    CK(rtdd_pyramid_image(ctx, RTDD_IMG_GRAY, 0, &p_gray, &pi_gray, nullptr, nullptr));
    CK(rtdd_pyramid_image(ctx, RTDD_IMG_DEPTH, 0, &p_depth, &pi_depth, nullptr, nullptr));
    CK(rtdd_pyramid_image(ctx, RTDD_IMG_ARTISTIC, 0, &p_art, &pi_art, nullptr, nullptr));
    CK(rtdd_pyramid_image(ctx, RTDD_IMG_DEPTH_U8, 0, &p_u8, &pi_u8, nullptr, nullptr));

    auto t0 = std::chrono::steady_clock::now();
    for (int n = 0; n < count; n++) {
        if (n == 0 || !live) {                                          // a new image: upload + (re)build the pyramid inputs
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
    }
    *ms_per_estimate = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / (count > 0 ? count : 1);
    (void)hipFree(d_bgr); (void)hipFree(d_ann);
    rtdd_ctx_destroy(ctx);
    (void)hipStreamDestroy(stream);
    return RTDD_OK;
}

int main(int argc, const char *argv[]) {
    if (argc == 1) { std::printf("Usage: rtdd_harness -i image.ppm [-a annotation.pgm] [-o prefix] [--effect defocus|desaturation|haze] [--iters N] [--refine sor|mg|auto [--tolerance T]]\n"
                                 "                    [--paint x,y,label,radius]... [--live N] [--devices D --batch B]\n"); return 0; }
    Job job;
    std::string in, an, out = "";
    int devices = 1, batch = 1, live = 0;
    for (int i = 1; i < argc; i++) {
        auto next = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
        if (!std::strcmp(argv[i], "-i")) in = next();
        else if (!std::strcmp(argv[i], "-a")) an = next();
        else if (!std::strcmp(argv[i], "-o")) out = next();
        else if (!std::strcmp(argv[i], "--effect")) job.effect = next();
        else if (!std::strcmp(argv[i], "--iters")) job.iters = std::atoi(next());
        else if (!std::strcmp(argv[i], "--refine")) job.refine = next();          // sor | mg | auto
        else if (!std::strcmp(argv[i], "--tolerance")) job.tolerance = (float)std::atof(next());
        else if (!std::strcmp(argv[i], "--devices")) devices = std::atoi(next());
        else if (!std::strcmp(argv[i], "--batch")) batch = std::atoi(next());
        else if (!std::strcmp(argv[i], "--live")) live = std::atoi(next());
        else if (!std::strcmp(argv[i], "--paint")) { Paint p{0, 0, 0, 0}; if (std::sscanf(next(), "%d,%d,%d,%d", &p.x, &p.y, &p.label, &p.radius) == 4) job.paints.push_back(p); }
        else if (!std::strcmp(argv[i], "-h")) std::printf("Usage:\n -i input image (binary PPM)\n -a annotated image (binary PGM)\n");
    }
    Pnm rgb;
    if (!read_pnm(in, rgb) || rgb.ch != 3) { std::printf("cannot read %s as a binary PPM\n", in.c_str()); return 2; }
    job.bgr = rgb;
    for (size_t i = 0; i < rgb.px.size(); i += 3) { job.bgr.px[i] = rgb.px[i + 2]; job.bgr.px[i + 2] = rgb.px[i]; }   // cv::imread gives BGR
    if (!an.empty()) {
        if (!read_pnm(an, job.ann) || job.ann.ch != 1 || job.ann.w != rgb.w || job.ann.h != rgb.h) { std::printf("cannot read %s as a binary PGM of the image's size\n", an.c_str()); return 2; }
        job.has_ann = true;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { std::printf("no HIP device: %s\n", rtdd_status_string(RTDD_ERR_NO_DEVICE)); return 3; }
    if (devices > ndev) devices = ndev;

    std::vector<std::vector<unsigned char>> depth(devices), art(devices);
    std::vector<double> ms(devices, 0.0);
    std::vector<int> rcs(devices, 0), counts(devices, 0);
    for (int b = 0; b < batch; b++) counts[b % devices]++;               // image b -> device b % D
    if (live > 0) { counts.assign(devices, 0); counts[0] = live; }
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int d = 0; d < devices; d++)
        th.emplace_back([&, d]() { rcs[d] = counts[d] ? run_device(d, job, counts[d], live > 0, &depth[d], &art[d], &ms[d]) : 0; });
    for (auto &t : th) t.join();
    const double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (int d = 0; d < devices; d++) if (rcs[d]) return 4;

    const int total = live > 0 ? live : batch;
    std::printf("Processing Time: %.3f ms per estimate on device 0 (upload + estimate%s + download); %d estimate(s) on %d device(s) in %.1f ms wall incl. setup\n",
                ms[0], job.effect.empty() ? "" : " + effect", total, devices, wall);
    if (!write_pnm(out + "DepthMap.pgm", rgb.w, rgb.h, 1, depth[0].data())) { std::printf("cannot write %sDepthMap.pgm\n", out.c_str()); return 5; }
    if (!job.effect.empty()) {
        std::vector<unsigned char> o(art[0]);
        for (size_t i = 0; i < o.size(); i += 3) { o[i] = art[0][i + 2]; o[i + 2] = art[0][i]; }     // BGR -> RGB for the file
        if (!write_pnm(out + "ArtisticEffect.ppm", rgb.w, rgb.h, 3, o.data())) return 5;
    }
    std::printf("Saving images...\n");                                   // main.cpp:317
    return 0;
}
