/* Plain C99 user of the drop-in boundary: proves include/rtdd.h is a C header (no C++ or torch types) and that a
 * C program can drive the whole estimate through it.  Built and run by tests/test_gpu_harness.py on the GPU box;
 * compiled (not run) by tests/test_abi.py on CPU.  Links only librtdd.so; device memory comes from the library's
 * own pyramid images. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rtdd.h"

#define CK(call) do { int rc_ = (call); if (rc_ != RTDD_OK) { printf("%s -> %s (%s)\n", #call, rtdd_status_string(rc_), rtdd_last_error(ctx)); return 1; } } while (0)

int main(void) {
    const int rows = 96, cols = 160;
    rtdd_ctx *ctx = NULL;
    int rc = rtdd_ctx_create(0, &ctx);
    if (rc != RTDD_OK) { printf("rtdd_ctx_create: %s\n", rtdd_status_string(rc)); return rc == RTDD_ERR_NO_DEVICE ? 77 : 1; }
    CK(rtdd_load_weights(ctx, 0.4f));
    CK(rtdd_pyramid_create(ctx, rows, cols));
    /* a grey ramp with a step edge, two scribbles (labels 0 and 254) */
    unsigned char *bgr = (unsigned char *)malloc((size_t)rows * cols * 3), *ann = (unsigned char *)malloc((size_t)rows * cols);
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            unsigned char g = (unsigned char)(x < cols / 2 ? 60 + y / 4 : 180 - y / 4);
            bgr[((size_t)y * cols + x) * 3 + 0] = g; bgr[((size_t)y * cols + x) * 3 + 1] = g; bgr[((size_t)y * cols + x) * 3 + 2] = g;
            ann[(size_t)y * cols + x] = 32;
            if (y > 40 && y < 50 && x > 10 && x < 40) ann[(size_t)y * cols + x] = 0;
            if (y > 40 && y < 50 && x > 120 && x < 150) ann[(size_t)y * cols + x] = 254;
        }
    void *p_orig, *p_u8, *p_ann_dev; size_t pi_orig, pi_u8, pi_ann;
    CK(rtdd_pyramid_image(ctx, RTDD_IMG_ORIGINAL, 0, &p_orig, &pi_orig, NULL, NULL));
    CK(rtdd_pyramid_image(ctx, RTDD_IMG_DEPTH_U8, 0, &p_u8, &pi_u8, NULL, NULL));
    CK(rtdd_pyramid_image(ctx, RTDD_IMG_ARTISTIC, 0, &p_ann_dev, &pi_ann, NULL, NULL));   /* borrowed as staging for the annotation */
    CK(rtdd_upload(ctx, p_orig, pi_orig, bgr, (size_t)cols * 3, (size_t)cols * 3, rows));
    CK(rtdd_pyramid_set_image(ctx, (const uint8_t *)p_orig, pi_orig));
    CK(rtdd_upload(ctx, p_ann_dev, pi_ann, ann, cols, cols, rows));
    CK(rtdd_pyramid_set_annotation(ctx, (const uint8_t *)p_ann_dev, pi_ann));
    CK(rtdd_estimate_depth(ctx, 1000));
    CK(rtdd_ctx_synchronize(ctx));
    unsigned char *depth = (unsigned char *)malloc((size_t)rows * cols);
    CK(rtdd_download(ctx, depth, cols, p_u8, pi_u8, cols, rows));
    const int a = depth[45 * cols + 20], b = depth[45 * cols + 135], mid = depth[45 * cols + 79];
    printf("label0 %d label254 %d left-of-edge %d\n", a, b, mid);
    if (a != 0 || b != 254 || mid > 127) { printf("unexpected depth values\n"); return 1; }   /* the edge at cols/2 keeps the left half near label 0 */
    /* the frame loop of src/main.cpp:232-295 from C: page-locked host images, two frames in flight; a frame that re-uploads the same
     * annotation is the next warm-started estimate -- the same map as calling rtdd_estimate_depth again */
    void *p_scr, *p_ed; size_t pi_scr, pi_ed;
    CK(rtdd_pyramid_image(ctx, RTDD_IMG_SCRIBBLE, 0, &p_scr, &pi_scr, NULL, NULL));
    CK(rtdd_pyramid_image(ctx, RTDD_IMG_EDITED, 0, &p_ed, &pi_ed, NULL, NULL));
    void *h_scr = NULL, *h_ed = NULL, *h_out[2] = {NULL, NULL};
    CK(rtdd_host_alloc(&h_scr, (size_t)rows * cols)); CK(rtdd_host_alloc(&h_ed, (size_t)rows * cols * 3));
    CK(rtdd_host_alloc(&h_out[0], (size_t)rows * cols)); CK(rtdd_host_alloc(&h_out[1], (size_t)rows * cols));
    CK(rtdd_download(ctx, h_scr, cols, p_scr, pi_scr, cols, rows));
    CK(rtdd_download(ctx, h_ed, (size_t)cols * 3, p_ed, pi_ed, (size_t)cols * 3, rows));
    CK(rtdd_live_submit(ctx, (const uint8_t *)h_scr, cols, (const uint8_t *)h_ed, (size_t)cols * 3, 1000, (uint8_t *)h_out[0], cols));
    CK(rtdd_live_submit(ctx, (const uint8_t *)h_scr, cols, (const uint8_t *)h_ed, (size_t)cols * 3, 1000, (uint8_t *)h_out[1], cols));
    if (rtdd_live_pending(ctx) != 2) { printf("two frames should be in flight\n"); return 1; }
    CK(rtdd_live_wait(ctx)); CK(rtdd_live_wait(ctx));
    CK(rtdd_download(ctx, depth, cols, p_u8, pi_u8, cols, rows));             /* the device's map = the second frame's */
    if (memcmp(depth, h_out[1], (size_t)rows * cols) != 0) { printf("live frame differs from the device's map\n"); return 1; }
    int heals = -1;
    CK(rtdd_get_option(ctx, RTDD_OPT_TIMEOUT_HEALS, &heals));
    if (heals != 0) { printf("unexpected heal\n"); return 1; }
    rtdd_host_free(h_scr); rtdd_host_free(h_ed); rtdd_host_free(h_out[0]); rtdd_host_free(h_out[1]);
    CK(rtdd_pyramid_destroy(ctx));
    rtdd_ctx_destroy(ctx);
    free(bgr); free(ann); free(depth);
    printf("c_abi_smoke ok\n");
    return 0;
}
