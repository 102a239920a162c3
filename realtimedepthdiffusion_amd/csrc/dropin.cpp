// dropin.cpp -- C++-linkage shim exporting the reference's ten symbols over the C ABI
// (see include/rtdd_dropin.hpp).  One lazily created process-global context.
#include <cstdio>
#include <cstdlib>

#include "rtdd.h"
#include "rtdd_dropin.hpp"

#define RTDD_EXPORT __attribute__((visibility("default")))

namespace {

rtdd_ctx *g_ctx = nullptr;

rtdd_ctx *ctx(const char *who) {
    if (!g_ctx) {
        const char *dev = std::getenv("RTDD_DEVICE");
        const int rc = rtdd_ctx_create(dev ? std::atoi(dev) : 0, &g_ctx);
        if (rc != RTDD_OK) {
            std::printf("%s: %s\n", who, rtdd_status_string(rc));     // the reference's error style, src/GPUSolver.cu:25
            g_ctx = nullptr;
        }
    }
    return g_ctx;
}

void report(const char *who, int rc) {
    if (rc != RTDD_OK) std::printf("%s: %s (%s)\n", who, rtdd_status_string(rc), rtdd_last_error(g_ctx));
}

void sync(const char *who) { report(who, rtdd_ctx_synchronize(g_ctx)); }

}  // namespace

extern "C" RTDD_EXPORT rtdd_ctx *rtdd_dropin_context(void) { return ctx("rtdd_dropin_context"); }

RTDD_EXPORT void GPUAllocateDeviceMemory(int rows, int cols, int levels) {
    if (!ctx("GPUAllocateDeviceMemory")) return;
    report("GPUAllocateDeviceMemory", rtdd_allocate(g_ctx, rows, cols, levels));
}

RTDD_EXPORT void GPUFreeDeviceMemory(int levels) {
    (void)levels;                                   // the context knows how many levels it holds
    if (!ctx("GPUFreeDeviceMemory")) return;
    report("GPUFreeDeviceMemory", rtdd_free(g_ctx));
}

RTDD_EXPORT void GPULoadWeights(float beta) {
    if (!ctx("GPULoadWeights")) return;
    report("GPULoadWeights", rtdd_load_weights(g_ctx, beta));
}

RTDD_EXPORT void GPUMatrixFreeSolver(float *depthImage, size_t depthPitch, unsigned char *scribbleImage, size_t scribblePitch,
                                     unsigned char *grayImage, size_t grayPitch, int rows, int cols, float beta, int maxIterations,
                                     float tolerance, int level) {
    if (!ctx("GPUMatrixFreeSolver")) return;
    report("GPUMatrixFreeSolver", rtdd_matrix_free_solver(g_ctx, depthImage, depthPitch, scribbleImage, scribblePitch, grayImage,
                                                          grayPitch, rows, cols, beta, maxIterations, tolerance, level));
    // src/GPUSolver.cu:314.  A persistent launch that timed out (shared GPU) is healed in here: the solve has run again, one launch per
    // block of sweeps, by the time this returns (api.cpp check_persistent_status) -- the caller's depth map is valid either way
    sync("GPUMatrixFreeSolver");
}

RTDD_EXPORT void GPUConvertToFloat(unsigned char *src, size_t srcPitch, float *dst, size_t dstPitch, unsigned char *mask,
                                   size_t maskPitch, int rows, int cols) {
    if (!ctx("GPUConvertToFloat")) return;
    report("GPUConvertToFloat", rtdd_convert_to_float(g_ctx, src, srcPitch, dst, dstPitch, mask, maskPitch, rows, cols));
}

RTDD_EXPORT void GPUPyrDownAnnotation(unsigned char *prevScribbleImage, size_t prevScribblePitch, unsigned char *prevEditedImage,
                                      size_t prevEditedPitch, int previousRows, int previousCols, unsigned char *currScribbleImage,
                                      size_t currScribblePitch, unsigned char *currEditedImage, size_t currEditedPitch,
                                      int currentRows, int currentCols) {
    if (!ctx("GPUPyrDownAnnotation")) return;
    report("GPUPyrDownAnnotation",
           rtdd_pyrdown_annotation(g_ctx, prevScribbleImage, prevScribblePitch, prevEditedImage, prevEditedPitch, previousRows,
                                   previousCols, currScribbleImage, currScribblePitch, currEditedImage, currEditedPitch,
                                   currentRows, currentCols));
}

RTDD_EXPORT void GPUPaintImage(int x, int y, int scribbleColor, int scribbleRadius, unsigned char *editedImage, size_t editedPitch,
                               unsigned char *scribbleImage, size_t scribblePitch, int rows, int cols) {
    if (!ctx("GPUPaintImage")) return;
    report("GPUPaintImage", rtdd_paint_image(g_ctx, x, y, scribbleColor, scribbleRadius, editedImage, editedPitch, scribbleImage,
                                             scribblePitch, rows, cols));
}

RTDD_EXPORT void GPUSimulateDefocus(unsigned char *originalImage, size_t originalPitch, float *depthImage, size_t depthPitch,
                                    unsigned char *artisticImage, size_t artisticPitch, int rows, int cols) {
    if (!ctx("GPUSimulateDefocus")) return;
    report("GPUSimulateDefocus", rtdd_simulate_defocus(g_ctx, originalImage, originalPitch, depthImage, depthPitch, artisticImage,
                                                       artisticPitch, rows, cols));
}

RTDD_EXPORT void GPUSimulateDesaturation(unsigned char *originalImage, size_t originalPitch, unsigned char *grayImage,
                                         size_t grayPitch, float *depthImage, size_t depthPitch, unsigned char *artisticImage,
                                         size_t artisticPitch, int rows, int cols) {
    if (!ctx("GPUSimulateDesaturation")) return;
    report("GPUSimulateDesaturation", rtdd_simulate_desaturation(g_ctx, originalImage, originalPitch, grayImage, grayPitch,
                                                                 depthImage, depthPitch, artisticImage, artisticPitch, rows, cols));
}

RTDD_EXPORT void GPUSimulateHaze(unsigned char *originalImage, size_t originalPitch, float *depthImage, size_t depthPitch,
                                 unsigned char *artisticImage, size_t artisticPitch, int rows, int cols) {
    if (!ctx("GPUSimulateHaze")) return;
    report("GPUSimulateHaze", rtdd_simulate_haze(g_ctx, originalImage, originalPitch, depthImage, depthPitch, artisticImage,
                                                 artisticPitch, rows, cols));
}
