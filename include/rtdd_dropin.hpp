// rtdd_dropin.hpp -- the reference's ten free functions, C++ linkage, same signatures and
// therefore the same Itanium-mangled symbols as /root/reference/include/GPUSolver.h:6-10,
// GPUImageProcessing.h:4-10 and GPUDepthEffect.h:4-9.  librtdd.so exports them on top of a
// process-global default context (device = $RTDD_DEVICE or 0, null stream), so host code
// written against the reference's headers links against librtdd.so unchanged.
//
// Behavioural contract kept from the reference: void returns; errors are printed as
// "<function>: <message>" and execution continues (src/GPUSolver.cu:21-27);
// GPUAllocateDeviceMemory / GPUFreeDeviceMemory / GPULoadWeights / GPUMatrixFreeSolver return
// after a device sync, the other six are asynchronous on the null stream.
// Like the reference this shim is NOT re-entrant; multi-GPU callers use the handle-based C ABI.
#pragma once
#include <cstddef>

void GPUAllocateDeviceMemory(int rows, int cols, int levels);
void GPUFreeDeviceMemory(int levels);
void GPULoadWeights(float beta);
void GPUMatrixFreeSolver(float *depthImage, size_t depthPitch, unsigned char *scribbleImage, size_t scribblePitch,
                         unsigned char *grayImage, size_t grayPitch, int rows, int cols, float beta, int maxIterations,
                         float tolerance, int level);

void GPUConvertToFloat(unsigned char *src, size_t srcPitch, float *dst, size_t dstPitch, unsigned char *mask, size_t maskPitch,
                       int rows, int cols);
void GPUPyrDownAnnotation(unsigned char *prevScribbleImage, size_t prevScribblePitch, unsigned char *prevEditedImage,
                          size_t prevEditedPitch, int previousRows, int previousCols, unsigned char *currScribbleImage,
                          size_t currScribblePitch, unsigned char *currEditedImage, size_t currEditedPitch, int currentRows,
                          int currentCols);
void GPUPaintImage(int x, int y, int scribbleColor, int scribbleRadius, unsigned char *editedImage, size_t editedPitch,
                   unsigned char *scribbleImage, size_t scribblePitch, int rows, int cols);

void GPUSimulateDefocus(unsigned char *originalImage, size_t originalPitch, float *depthImage, size_t depthPitch,
                        unsigned char *artisticImage, size_t artisticPitch, int rows, int cols);
void GPUSimulateDesaturation(unsigned char *originalImage, size_t originalPitch, unsigned char *grayImage, size_t grayPitch,
                             float *depthImage, size_t depthPitch, unsigned char *artisticImage, size_t artisticPitch, int rows,
                             int cols);
void GPUSimulateHaze(unsigned char *originalImage, size_t originalPitch, float *depthImage, size_t depthPitch,
                     unsigned char *artisticImage, size_t artisticPitch, int rows, int cols);

// Not part of the reference: the process-global context the ten functions above run on (created on first use), for a host that
// wants to set an option (rtdd_set_option: RTDD_OPT_FP_CONTRACT ...) or read a counter (RTDD_OPT_TIMEOUT_HEALS) on it.  The
// unchanged main.cpp never needs it: a persistent launch that times out on a shared GPU is healed inside GPUMatrixFreeSolver
// (include/rtdd.h, RTDD_ERR_TIMEOUT), which -- like the reference's (src/GPUSolver.cu:311-314) -- always returns with a valid depth map.
struct rtdd_ctx;
extern "C" rtdd_ctx *rtdd_dropin_context(void);
