/*
 * rtdd.h -- C ABI of librtdd.so, the MI355X-native (gfx950 / HIP) implementation of the
 * RealTimeDepthDiffusion hot path: the Chebyshev-accelerated Jacobi diffusion solve plus the
 * per-pixel edge-weight, annotation and depth-effect passes.
 *
 * This is the drop-in boundary.  Every entry point below replaces one of the reference's ten
 * free functions (file:line given per function, relative to /root/reference); the reference-
 * side binding a maintainer would add is in INTEGRATION.md.  Differences from the reference
 * interface, all deliberate:
 *   - handle based (one rtdd_ctx per GPU, no file-scope globals) so N host threads or N
 *     processes can drive N GPUs; the reference is non-reentrant (src/GPUSolver.cu:13-19);
 *   - every call returns an int status (0 = ok) instead of printf-and-continue
 *     (src/GPUSolver.cu:21-27);
 *   - calls are stream-ordered and ASYNCHRONOUS on the context's stream; the C++ shim that
 *     exports the reference's mangled symbols (csrc/dropin.cpp) adds the device syncs the
 *     reference has (src/GPUSolver.cu:23,314).
 *
 * Pointer semantics are the reference's: every image pointer is a DEVICE pointer to pitched
 * row-major memory, pitch in BYTES next to it, rows/cols in pixels; u8x3 images are
 * interleaved (x*3+c); depth is f32, nominally in [0,255].  The library never allocates
 * caller-visible memory.  No torch, no C++ types in any signature.
 */
#ifndef RTDD_H
#define RTDD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rtdd_ctx rtdd_ctx;
typedef void *rtdd_stream;          /* a hipStream_t; NULL = the device's null stream */

enum rtdd_status {
    RTDD_OK = 0,
    RTDD_ERR_INVALID = 1,           /* bad argument (null pointer, negative size, level out of range ...) */
    RTDD_ERR_STATE = 2,             /* call order violated (solve before allocate / load_weights) */
    RTDD_ERR_HIP = 3,               /* a HIP runtime call failed; see rtdd_last_error() */
    RTDD_ERR_NOMEM = 4,
    RTDD_ERR_NO_DEVICE = 5,         /* no usable gfx950 device: there is NO CPU fallback */
    RTDD_ERR_TIMEOUT = 6            /* a persistent sweep launch gave up waiting for a neighbouring workgroup (the GPU was shared, so
                                       its workgroups were not all resident at once) AND running the affected calls again failed too, or a
                                       wave gave up waiting for a wave of its own workgroup (an internal error).  A first time-out is NOT
                                       an error (the reference's solver always leaves a valid depth map, src/GPUSolver.cu:311-314): the
                                       failed launch and everything queued behind it drain at once, the copy-back kernels behind them store
                                       nothing (every affected call keeps its input), and the next call that synchronises the stream anyway
                                       -- rtdd_ctx_synchronize, rtdd_download, a residual-stopped rtdd_solve_ex -- switches persistent launches
                                       off (suspended for RTDD_OPT_PERSISTENT_REARM_AFTER solves, twice as many after every further time-out,
                                       for good after four: RTDD_OPT_PERSISTENT_SUSPENDED), prints one warning on stderr, runs the solves /
                                       estimates still unconfirmed again from the failed one on and returns RTDD_OK: see RTDD_OPT_TIMEOUT_HEALS.
                                       The three depth effects queued behind an unconfirmed solve are remembered and run again with it (they
                                       may have read the solve's INPUT); the calls that read a solve's output without being remembered --
                                       rtdd_pyrup_depth, rtdd_depth_to_u8, rtdd_index_to_weight, rtdd_upload -- and the calls that change what
                                       a remembered call ran on (rtdd_ctx_set_stream, rtdd_allocate, rtdd_free, rtdd_load_weights,
                                       rtdd_pyramid_create / _destroy / _set_image) first settle the log: synchronise, look at the status
                                       word, heal.  The annotation calls (paint, pyrDown, convert) are neither: they do not depend on a solve.
                                       LIFETIME RULE: the images handed to a solve, an estimate or an effect must stay valid until an rtdd
                                       call that synchronises (rtdd_ctx_synchronize, rtdd_download, rtdd_live_wait, a settling call above) has
                                       returned RTDD_OK -- a replay reads and writes them again, and a hipDeviceSynchronize / hipStreamSynchronize
                                       of the caller's own does not look at the status word.  A call leaves the log as soon as the kernel that
                                       publishes its result has run with the status word clear (no synchronisation needed), so the log holds
                                       calls in flight, not history.  A host that cannot keep that rule sets RTDD_OPT_TIMEOUT_HEAL to 0:
                                       nothing is remembered, a time-out comes back as RTDD_ERR_TIMEOUT from the next synchronising call */
};

/* Solver variants.  RTDD_METHOD_CHEBYSHEV_JACOBI is the reference's only scheme
 * (src/GPUSolver.cu:282-309); the others are extensions with no reference behaviour
 *  and are opt-in through rtdd_solve_ex(). */
enum rtdd_method {
    RTDD_METHOD_CHEBYSHEV_JACOBI = 0,
    RTDD_METHOD_RED_BLACK_GS = 1,
    RTDD_METHOD_MULTIGRID = 2,       /* V(2,2) cycles, operator-dependent interpolation; maxIterations counts CYCLES,
                                      * checkEvery defaults to 1 cycle.  With tolerance > 0 and a check every cycle the driver
                                      * also extrapolates (x + l/(1-l) (x - x_prev)) whenever the residual ratio l of two
                                      * consecutive cycles has settled; tolerance <= 0 runs plain cycles */
    RTDD_METHOD_AUTO = 3             /* to a tolerance (required): V-cycles while they pay (until the tolerance, 60 cycles, or until
                                      * the cycles still needed at the current rate are modelled dearer than the sweeps), then red-black SOR cycles
                                      * (RTDD_RELAXATION_AUTO, started at half length: N = max(rows,cols)/2 rounded up) from there;
                                      * maxIterations caps the SOR sweeps */
};

/* Tunables, rtdd_set_option(ctx, key, value). */
enum rtdd_option {
    RTDD_OPT_FP_CONTRACT = 0,       /* 1 (default): fused multiply-adds where nvcc -fmad=true fuses; 0: none */
    RTDD_OPT_SWEEP_KERNEL = 1,      /* 0 auto (default), 1 one sweep per launch, 2 temporally blocked */
    RTDD_OPT_TEMPORAL_DEPTH = 2,    /* sweeps fused per launch by the blocked kernel (0 = auto) */
    RTDD_OPT_DEFOCUS_PATH = 3,      /* rtdd_simulate_defocus: 0 (default) automatic -- one launch with per-tile summed-area tables in LDS when the largest
                                       nominal window is <= 56 pixels wide (images up to a 2280-pixel diagonal: 1080p), a global table otherwise;
                                       1 the global table always; 2 the tile kernel wherever its region fits (0 falls back to the table for the rest of
                                       the context's life once a call has met depths far outside [0, 255]: windows beyond a tile's region are
                                       summed directly there, exactly but at a cost that grows with their area).  Same bits either way */
    RTDD_OPT_ROWS_PER_WAVE = 4,     /* one-sweep kernel: rows each wave walks (0 = auto) */
    RTDD_OPT_PERSISTENT = 6,        /* (rtdd_get_option returns what is in force: 0 while a time-out has persistent launches suspended)
                                       1 (default): levels whose tiles all fit on the chip at once run ALL sweeps in one launch,
                                       neighbouring workgroups trading halo strips in memory (no kernel boundaries); larger levels (4K, 8K) run
                                       one launch per block of sweeps.  0: one launch per block of sweeps everywhere */
    /* RTDD_METHOD_AUTO prices the V-cycles still needed against finishing with SOR cycles.  The prices are these four CONSTANTS
     * (never a clock: a solve is reproducible); they are options so that the decision can be restated from outside and re-tuned
     * without touching bits by accident.  cycle = FIXED_NS + pixels * CYCLE_FS_PER_PX; sweep = max(FLOOR_NS, pixels * SWEEP_FS_PER_PX). */
    RTDD_OPT_AUTO_CYCLE_FIXED_NS = 9,    /* default 270000 */
    RTDD_OPT_AUTO_CYCLE_FS_PER_PX = 10,  /* default 46000 (femtoseconds per level-0 pixel) */
    RTDD_OPT_AUTO_SWEEP_FS_PER_PX = 11,  /* default 1429 */
    RTDD_OPT_AUTO_SWEEP_FLOOR_NS = 12,   /* default 2500 */
    RTDD_OPT_DEBUG_WITHHOLD_TILE = 7, /* testing aid: tile number + 1 whose hand-off flag a persistent launch never publishes (0 = off),
                                       so that its neighbours run into the poll limit -> RTDD_ERR_TIMEOUT */
    RTDD_OPT_DEBUG_POLL_LIMIT_US = 8, /* testing aid: that poll limit in microseconds (0 = the default, 200 ms) */
    RTDD_OPT_DEBUG_FORCE_STATUS = 13, /* testing aid: value (1, 2) stored into the kernels' status word right behind the next temporally blocked
                                       Jacobi launch, persistent or not, as if a wave of it had given up (one shot: resets to 0); 3 = status 1
                                       now AND once more when that solve is run again, so that the replay fails too (-> RTDD_ERR_TIMEOUT) */
    RTDD_OPT_DEFOCUS_LAST_PATH = 15, /* read only: what the most recent rtdd_simulate_defocus launched -- 1 the global table, 2 the tile kernel (0: none yet).
                                       Setting RTDD_OPT_DEFOCUS_PATH to 0 also forgets an earlier fall-back of the automatic choice to the table */
    RTDD_OPT_TIMEOUT_HEALS = 14,    /* read only: how many times this context has healed a timed-out persistent launch (see RTDD_ERR_TIMEOUT) */
    RTDD_OPT_TIMEOUT_HEAL = 16,     /* 1 (default): heal as described at RTDD_ERR_TIMEOUT; 0: remember nothing, report the time-out */
    RTDD_OPT_PERSISTENT_REARM_AFTER = 17, /* solves run without persistence after the FIRST healed time-out before persistent launches are tried
                                       again (default 64; doubles with every further time-out; after four time-outs persistence stays off;
                                       0: off for good at the first).  Setting RTDD_OPT_PERSISTENT to 1 explicitly re-arms at once */
    RTDD_OPT_PERSISTENT_SUSPENDED = 18, /* read only: 0 persistent launches are armed (or RTDD_OPT_PERSISTENT is 0 by the caller's choice); n > 0: suspended
                                       for n more solves after a time-out; -1: off for the rest of the context's life */
    RTDD_OPT_PENDING_CALLS = 19,    /* read only: calls currently remembered for a replay (see RTDD_ERR_TIMEOUT): those still in flight */
    RTDD_OPT_LIVE_ZERO_COPY = 20,   /* rtdd_live_submit lets the estimate's copy-back kernel store the u8 map straight into hostDepthU8 when that buffer is
                                       page-locked (rtdd_host_alloc, hipHostMalloc, hipHostRegister) -- no staging slot, no download.  1 (default): when
                                       no other frame is in flight (one frame at a time; in a pipelined loop rtdd_live_wait downloads the staged map
                                       while the next frame computes, which is cheaper still); 2: always; 0: never */
    RTDD_OPT_DEFOCUS_STRIPS = 22,   /* the table path of rtdd_simulate_defocus, tile order of the lookup: 0 (default) automatic -- each XCD takes a COLUMN strip of the
                                       image where the table rows between a window's bottom and top edge, over the whole image width, outgrow an XCD's
                                       L2 (from ~4K on: 8K 727 -> 620 us on a smooth depth map, 2.9 -> 1.3 ms with a random depth per pixel, before the corner reuse of the same round), row bands
                                       otherwise; 1 row bands always; 2 column strips always.  Same bits */
    RTDD_OPT_DEFOCUS_LAST_SLICES = 23, /* read only: the horizontal slices the most recent table-path rtdd_simulate_defocus built a table for (1: one
                                       whole-image table: the default) */
    RTDD_OPT_DEFOCUS_SLICE_MB = 24, /* the table path of rtdd_simulate_defocus: a summed-area table (8 bytes per pixel) of more than twice this many MB is
                                       built and looked up slice by slice -- output rows + the tallest nominal window's reach above and below, each slice's
                                       table at most this large and with an origin of its own, all in one buffer.  Default 0: always one whole-image
                                       table (slices measured SLOWER at 8K -- 762 against 725 us: what the 265 MB table loses is L2 reuse, which
                                       RTDD_OPT_DEFOCUS_STRIPS restores, not the Infinity Cache); they are what lets an image beyond 2^29 pixels, whose
                                       whole table would pass 4 GiB, be processed at all.  Same bits either way; depths above 255 (windows beyond a
                                       slice) are answered exactly and send the context's later calls back to the whole-image table */
    RTDD_OPT_ANNOTATION_LDS = 21,   /* 1 (default): an estimate's annotation pyramid walks its levels in LDS (one launch, one memory round trip; pyramids of
                                       up to six levels); 0: the same launch with the levels read back from global memory (a developer's A/B knob) */
    RTDD_OPT_TILE = 5               /* blocked kernel extended tile: 0 auto, 1 = 64x64, 2 = 128x64, 3 = 128x128,
                                       4 = 128x96, 5 = 128x48, 6 = 64x96, 7 = 64x48, 8 = 128x64 (8 px/thread),
                                       9 = 64x64 (4 px/thread), 10 = 64x64 (8 px/thread), 11 = 128x32 (4 px/thread),
                                       12 = 128x96 (16 px/thread, 768 threads), 13 = 128x96 (24 px/thread, 512 threads),
                                       14 = 64x64 in the column layout (1 px x 4 rows per thread; small pyramid levels),
                                       15 = 64x32 and 16 = 64x48 in the column layout */
};

/* ---- context ------------------------------------------------------------------------------- */
int rtdd_ctx_create(int device, rtdd_ctx **out);
int rtdd_ctx_destroy(rtdd_ctx *ctx);
int rtdd_ctx_set_stream(rtdd_ctx *ctx, rtdd_stream stream);
int rtdd_ctx_synchronize(rtdd_ctx *ctx);                 /* hipStreamSynchronize on the context's stream */
int rtdd_set_option(rtdd_ctx *ctx, int key, int value);
int rtdd_get_option(rtdd_ctx *ctx, int key, int *value);
const char *rtdd_last_error(rtdd_ctx *ctx);              /* message of the last failing call on ctx */
const char *rtdd_status_string(int status);
int rtdd_version(void);                                  /* major * 100 + minor.  200: rtdd_solve_info grew from 12 to 36 bytes (kernel .. launches);
                                                          * rtdd_solve_ex / rtdd_refine_depth / rtdd_last_solve_info write the whole struct, so a
                                                          * caller compiled against a 1xx header must be rebuilt (#define RTDD_VERSION below).  210 adds
                                                          * rtdd_pyramid_annotation_changed, RTDD_OPT_TIMEOUT_HEALS and the self-healing time-out; 220: RTDD_OPT_TIMEOUT_HEAL,
                                                          * persistence re-armed after a time-out, rtdd_estimate_depth_batch, RTDD_OPT_LIVE_ZERO_COPY;
                                                          * 230: rtdd_pyramid_level_info, rtdd_live_submit_ex, RTDD_OPT_DEFOCUS_STRIPS / _SLICE_MB / _LAST_SLICES */
#define RTDD_VERSION 230

/* ---- solver (include/GPUSolver.h:6-10) ------------------------------------------------------ */

/* GPUAllocateDeviceMemory(rows, cols, levels) -- src/GPUSolver.cu:33-54.
 * Per-level private scratch for levels 0..levels-1 of size (int)(rows/2^l) x (int)(cols/2^l);
 * sets maxLevel = levels-1 (selects the un-gated weight rule, src/GPUSolver.cu:166,188). */
int rtdd_allocate(rtdd_ctx *ctx, int rows, int cols, int levels);

/* GPUFreeDeviceMemory(levels) -- src/GPUSolver.cu:56-71. */
int rtdd_free(rtdd_ctx *ctx);

/* GPULoadWeights(beta) -- src/GPUSolver.cu:264-272.  LUT w[i] = expf(-beta*i) computed on the
 * HOST with libm like the reference, w[256] = 0, f32 denormals preserved. */
int rtdd_load_weights(rtdd_ctx *ctx, float beta);

/* GPUMatrixFreeSolver(...) -- src/GPUSolver.cu:274-316.  Exactly maxIterations Chebyshev-Jacobi
 * sweeps on level `level`; depth is read (initial guess + Dirichlet values where scribble==255)
 * and overwritten with the result.  beta and tolerance are accepted and ignored, as in the
 * reference (src/GPUSolver.cu:274-275).  rows/cols must not exceed the level's allocation. */
int rtdd_matrix_free_solver(rtdd_ctx *ctx, float *depth, size_t depthPitch,
                            const uint8_t *scribble, size_t scribblePitch,
                            const uint8_t *gray, size_t grayPitch,
                            int rows, int cols, float beta, int maxIterations, float tolerance, int level);

/* Extension (no reference counterpart): same inputs, selectable method and an optional
 * residual stop.  With method = CHEBYSHEV_JACOBI, tolerance <= 0 it is bit-identical to
 * rtdd_matrix_free_solver. */
typedef struct rtdd_solve_params {
    int method;                     /* enum rtdd_method */
    int maxIterations;              /* upper bound on sweeps */
    float tolerance;                /* stop when max|J(x)-x| over free pixels <= tolerance; <= 0: never */
    int checkEvery;                 /* residual is evaluated every checkEvery sweeps (0 = 16) */
    float relaxation;               /* RED_BLACK_GS only: SOR factor in (0,2), x <- clamp(x + relaxation (gs - x));
                                     * 0 or 1 = plain Gauss-Seidel; RTDD_RELAXATION_AUTO = SOR cycles.  Cycle c (e = min(c,6)),
                                     * with gap = max(0.005, (2 - min(1.99, 2/(1+sin(4 pi/N)))) / 2^e), N = max(rows,cols):
                                     * N 2^e sweeps at omega = 2 - gap, a quarter as many at max(1, 2 - 10 gap), then <= 100
                                     * plain sweeps with the residual checked every 20 (checkEvery is ignored); cycles repeat
                                     * until tolerance or maxIterations */
} rtdd_solve_params;
#define RTDD_RELAXATION_AUTO (-1.0f)

typedef struct rtdd_solve_info {
    int iterations;                 /* sweeps actually executed (RTDD_METHOD_MULTIGRID: cycles) */
    float residual;                 /* last evaluated max|J(x)-x| (NaN if never evaluated) */
    int cycles;                     /* V-cycles executed (RTDD_METHOD_MULTIGRID, RTDD_METHOD_AUTO), else 0 */
    /* what actually ran, so that a log line identifies the code path (the automatic choices depend on the image size): */
    int kernel;                     /* sweep kernel of the LAST sweep launch: 1 one Jacobi sweep per launch, 2 temporally blocked Jacobi,
                                     * 3 one red-black colour per launch, 4 register-blocked red-black; 0 = no sweep launch */
    int tile;                       /* blocked kernels: tile id (RTDD_OPT_TILE numbering; red-black: 1 = 128x64, 2 = 128x128) */
    int temporal_depth;             /* blocked kernels: sweeps per launch (persistent: per exchange) */
    int persistent;                 /* 1: that launch was persistent (all its sweeps in one launch) */
    int fp_contract;                /* RTDD_OPT_FP_CONTRACT in force */
    int launches;                   /* kernel launches of the solve, k_prepare / k_finish excluded */
} rtdd_solve_info;

int rtdd_solve_ex(rtdd_ctx *ctx, float *depth, size_t depthPitch,
                  const uint8_t *scribble, size_t scribblePitch,
                  const uint8_t *gray, size_t grayPitch,
                  int rows, int cols, int level,
                  const rtdd_solve_params *params, rtdd_solve_info *info);

/* The rtdd_solve_info of the most recent rtdd_solve_ex / rtdd_matrix_free_solver call on this context (also of the per-level
 * solves inside rtdd_estimate_depth: the finest level's). */
int rtdd_last_solve_info(rtdd_ctx *ctx, rtdd_solve_info *info);

/* Diagnostic for the parity tests: after a RTDD_METHOD_MULTIGRID solve, copy plane `which` (0-4: couplings E,S,SE,SW and
 * diagonal D; 5-8: interpolation weights; 9-11: e, b, r) of hierarchy level `level` to host memory, dense rows x cols
 * floats.  host == NULL only reports the size.  Synchronises. */
int rtdd_multigrid_level(rtdd_ctx *ctx, int level, int which, float *host, int *rows, int *cols);

/* The edge-weight index pass on its own (loadIndexToWeight, src/GPUSolver.cu:136-224), exposed
 * for parity tests: writes the reference's int2 {left*1000+right, up*1000+down} per pixel,
 * dense (y*cols+x), into a device buffer of rows*cols*2 int32. */
int rtdd_index_to_weight(rtdd_ctx *ctx, const uint8_t *gray, size_t grayPitch,
                         const float *depth, size_t depthPitch,
                         int32_t *index2, int level, int rows, int cols);

/* ---- image processing (include/GPUImageProcessing.h:4-10) ---------------------------------- */

/* GPUConvertToFloat -- src/GPUImageProcessing.cu:8-21,72-79: dst[y][x] = src[y][3x] where mask==255. */
int rtdd_convert_to_float(rtdd_ctx *ctx, const uint8_t *src, size_t srcPitch, float *dst, size_t dstPitch,
                          const uint8_t *mask, size_t maskPitch, int rows, int cols);

/* GPUPyrDownAnnotation -- src/GPUImageProcessing.cu:23-49,81-91. */
int rtdd_pyrdown_annotation(rtdd_ctx *ctx, const uint8_t *prevScribble, size_t prevScribblePitch,
                            const uint8_t *prevEdited, size_t prevEditedPitch, int previousRows, int previousCols,
                            uint8_t *currScribble, size_t currScribblePitch,
                            uint8_t *currEdited, size_t currEditedPitch, int currentRows, int currentCols);

/* GPUPaintImage -- src/GPUImageProcessing.cu:51-70,93-101 (square brush, integer radius/2). */
int rtdd_paint_image(rtdd_ctx *ctx, int x, int y, int scribbleColor, int scribbleRadius,
                     uint8_t *edited, size_t editedPitch, uint8_t *scribble, size_t scribblePitch, int rows, int cols);

/* ---- depth effects (include/GPUDepthEffect.h:4-9) ------------------------------------------ */

/* GPUSimulateDefocus -- src/GPUDepthEffect.cu:29-72,105-113 (exact, via an integer summed-area table). */
int rtdd_simulate_defocus(rtdd_ctx *ctx, const uint8_t *original, size_t originalPitch,
                          const float *depth, size_t depthPitch, uint8_t *artistic, size_t artisticPitch,
                          int rows, int cols);

/* GPUSimulateDesaturation -- src/GPUDepthEffect.cu:8-27,95-103. */
int rtdd_simulate_desaturation(rtdd_ctx *ctx, const uint8_t *original, size_t originalPitch,
                               const uint8_t *gray, size_t grayPitch, const float *depth, size_t depthPitch,
                               uint8_t *artistic, size_t artisticPitch, int rows, int cols);

/* GPUSimulateHaze -- src/GPUDepthEffect.cu:74-93,115-123. */
int rtdd_simulate_haze(rtdd_ctx *ctx, const uint8_t *original, size_t originalPitch,
                       const float *depth, size_t depthPitch, uint8_t *artistic, size_t artisticPitch,
                       int rows, int cols);

/* ---- whole-estimate driver ------------------------------------------------
 * One depth estimate = the loop body of src/main.cpp:232-295, run as a single stream-ordered launch
 * sequence with the gray pyramid, f32 pyrUp and u8 conversion ON THE DEVICE (the reference round-trips
 * through the host for cv::pyrDown / cv::pyrUp).  The context owns the pyramid images, like main.cpp
 * owns its GpuMats (src/main.cpp:117-137).  The OpenCV ops are third party: the formulas used are
 * stated in oracle/rtdd_cascade_oracle.c. */
enum rtdd_pyramid_image_kind {
    RTDD_IMG_ORIGINAL = 0,          /* u8x3, level 0 only */
    RTDD_IMG_GRAY = 1,              /* u8, ceil-sized chain (cv::pyrDown's default size) */
    RTDD_IMG_SCRIBBLE = 2,          /* u8, 255 = Dirichlet */
    RTDD_IMG_EDITED = 3,            /* u8x3 */
    RTDD_IMG_DEPTH = 4,             /* f32 */
    RTDD_IMG_DEPTH_U8 = 5,          /* u8, level 0 only */
    RTDD_IMG_ARTISTIC = 6           /* u8x3, level 0 only: output of the effect calls */
};
int rtdd_pyramid_levels(int rows, int cols);             /* src/main.cpp:95 */
int rtdd_pyramid_create(rtdd_ctx *ctx, int rows, int cols);   /* main.cpp:92-155 minus I/O: images, depth := 255, rtdd_allocate */
int rtdd_pyramid_destroy(rtdd_ctx *ctx);
/* Batched estimates (BASELINE configs[3]: independent images, several per GPU).  The coarse levels of one image occupy a fraction of the
 * chip -- 120 x 67 and 240 x 135 are 0.7 of a 1080p estimate's time on ~135 of 512 workgroup slots -- so `images` pyramids of ONE size on one
 * context run every level of all images in the same launches (blockIdx.z = image): rtdd_estimate_depth_batch.  rtdd_pyramid_select
 * says which image rtdd_pyramid_set_image / _set_annotation / _image (and rtdd_download of what it returns), rtdd_estimate_depth and
 * rtdd_refine_depth address (0 after creation).  Every image's maps are bit for bit those of a single-image pyramid.  Live frames
 * (rtdd_live_submit) need a single-image pyramid. */
int rtdd_pyramid_create_batch(rtdd_ctx *ctx, int rows, int cols, int images);
int rtdd_pyramid_select(rtdd_ctx *ctx, int index);
int rtdd_pyramid_batch(rtdd_ctx *ctx);                   /* the number of images of the context's pyramid (0: none) */
int rtdd_estimate_depth_batch(rtdd_ctx *ctx, int maxIterations);   /* rtdd_estimate_depth for every image of the batch; asynchronous */
/* What the most recent estimate ran on pyramid level `level` (0 = finest): the rtdd_solve_info of that level's solve, and how many images
 * each of its sweep launches covered (a batch: all of them in the same launches, or 1 = image after image, each with the launch a single
 * solve gets).  The choice is a function of the level's size AND of the batch size, so that a log line / a test can name the kernel
 * configuration a timed batch actually ran (tests/test_gpu_batch.py pins the ones bench.py times).  info->temporal_depth is the NOMINAL
 * number of sweeps per launch / exchange here (rtdd_last_solve_info reports the last launch's, which may be the short tail block; a level
 * that is one tile runs all its sweeps in one launch).  imagesPerLaunch may be NULL. */
int rtdd_pyramid_level_info(rtdd_ctx *ctx, int level, rtdd_solve_info *info, int *imagesPerLaunch);
/* image: DEVICE pointer to an interleaved BGR u8 image; builds the gray pyramid, edited[0] := image, scribble[0] := 0 */
int rtdd_pyramid_set_image(rtdd_ctx *ctx, const uint8_t *bgr, size_t pitch);
/* annotation: DEVICE pointer to a 1-channel u8 map; decode rule of src/main.cpp:160-168 (value != 32 -> label, mask 255) */
int rtdd_pyramid_set_annotation(rtdd_ctx *ctx, const uint8_t *annotation, size_t pitch);
int rtdd_pyramid_image(rtdd_ctx *ctx, int kind, int level, void **ptr, size_t *pitch, int *rows, int *cols);
/* The coarse annotation levels (GPUPyrDownAnnotation, src/main.cpp:249-253) and the coarsest level's injection (:257-259) are brought up
 * to date by the first rtdd_estimate_depth after the annotation changed, not by every estimate (they depend on nothing else, the
 * down-sampling only ever adds and the solver never moves a Dirichlet pixel: same images, same bits).  Every entry point of this
 * library that writes RTDD_IMG_SCRIBBLE / RTDD_IMG_EDITED (set_image, set_annotation, rtdd_paint_image, rtdd_pyrdown_annotation,
 * rtdd_upload) notes the change itself; a caller that writes those images, or the coarsest RTDD_IMG_DEPTH, through the raw
 * pointers by other means says so with this call. */
int rtdd_pyramid_annotation_changed(rtdd_ctx *ctx);
/* src/main.cpp:239-291; asynchronous; results in RTDD_IMG_DEPTH (all levels) and RTDD_IMG_DEPTH_U8 */
int rtdd_estimate_depth(rtdd_ctx *ctx, int maxIterations);
/* Live mode: one frame of src/main.cpp:232-295 as the reference clocks it -- upload of the host's scribble and edited images
 * (:236-237), the estimate, download of the u8 map (:290-291) -- pipelined two frames deep: the copies run on a second stream of the
 * context's, so frame N+1's upload and frame N's download overlap the other frame's arithmetic and a frame costs about
 * max(compute, copies).  rtdd_live_submit returns at once; the host buffers must stay valid (and unchanged) until the frame has been
 * waited for, and should be page-locked (rtdd_host_alloc) -- pageable memory makes the copies synchronous, and with no other frame in
 * flight a page-locked hostDepthU8 is written by the estimate's last kernel itself instead of being downloaded (RTDD_OPT_LIVE_ZERO_COPY);
 * otherwise rtdd_live_wait downloads the map.  hostScribble / hostEdited
 * NULL: no upload, the annotation is the one already on the device.  A third submit waits for the oldest frame itself.
 * rtdd_live_wait blocks until the OLDEST frame in flight has landed in its host buffer (a timed-out persistent launch is healed
 * there like in rtdd_ctx_synchronize: every frame in flight is run again on its own uploaded annotation; the COARSE annotation levels,
 * which only ever accumulate, may by then hold the newer frame's strokes too).  Results: hostDepthU8, and RTDD_IMG_DEPTH /
 * RTDD_IMG_DEPTH_U8 on the device as after rtdd_estimate_depth.  No staging copies: an uploaded annotation pair BECOMES the pyramid's
 * level-0 RTDD_IMG_SCRIBBLE / RTDD_IMG_EDITED -- pointers obtained from rtdd_pyramid_image for those two images are good until the
 * next rtdd_live_submit that uploads: ask again after it (every other image keeps its address; a library call handed such a retired
 * pointer -- rtdd_paint_image, rtdd_upload, rtdd_convert_to_float, rtdd_pyrdown_annotation -- fails with RTDD_ERR_STATE instead of
 * writing a buffer no estimate reads). */
int rtdd_live_submit(rtdd_ctx *ctx, const uint8_t *hostScribble, size_t scribblePitch, const uint8_t *hostEdited, size_t editedPitch,
                     int maxIterations, uint8_t *hostDepthU8, size_t depthPitch);
/* The reference's frame WITH a sticky depth effect (src/main.cpp:190-230: once 'b' / 'g' / 'h' has been pressed the effect is rendered
 * and its image downloaded in every iteration of the loop at :180, next to the estimate at :232).  rtdd_live_submit plus: `effect` is
 * rendered from the frame's own depth map (RTDD_IMG_ORIGINAL, RTDD_IMG_GRAY, RTDD_IMG_DEPTH level 0) by the effect's kernel queued
 * right behind the estimate's copy-back, and the artistic image lands in hostArtistic (u8 x 3, page-locked for an asynchronous copy)
 * by the time rtdd_live_wait returns for the frame: downloaded by rtdd_live_wait while the next frame computes when frames are
 * pipelined, queued on the compute stream when no other frame is in flight.  RTDD_IMG_ARTISTIC then names the newest frame's image
 * on the device (like the annotation pair: ask rtdd_pyramid_image again after a submit with an effect).  The reference renders the
 * effect at the TOP of the next loop iteration from the same depth map -- the same sequence of artistic images, shown one iteration
 * later.  RTDD_EFFECT_NONE: exactly rtdd_live_submit (hostArtistic ignored).  A healed time-out renders the effect again too. */
enum rtdd_effect { RTDD_EFFECT_NONE = 0, RTDD_EFFECT_DEFOCUS = 1, RTDD_EFFECT_DESATURATION = 2, RTDD_EFFECT_HAZE = 3 };
int rtdd_live_submit_ex(rtdd_ctx *ctx, const uint8_t *hostScribble, size_t scribblePitch, const uint8_t *hostEdited, size_t editedPitch,
                        int maxIterations, uint8_t *hostDepthU8, size_t depthPitch, int effect, uint8_t *hostArtistic, size_t artisticPitch);
int rtdd_live_wait(rtdd_ctx *ctx);
int rtdd_live_pending(rtdd_ctx *ctx);                    /* frames submitted and not yet waited for: 0..2 */
int rtdd_host_alloc(void **ptr, size_t bytes);           /* page-locked host memory (hipHostMalloc) / its release */
int rtdd_host_free(void *ptr);
/* Extension: one more solve of the finest level, in place on RTDD_IMG_DEPTH level 0, by rtdd_solve_ex with `params`
 * (e.g. RTDD_METHOD_RED_BLACK_GS + RTDD_RELAXATION_AUTO, or RTDD_METHOD_MULTIGRID, tolerance 1e-4), then RTDD_IMG_DEPTH_U8
 * again: "estimate, then converge".  The level-0 edge weights are rebuilt from the current depth, as every solve does
 * (src/GPUSolver.cu:136-224).  Synchronises when params->tolerance > 0. */
int rtdd_refine_depth(rtdd_ctx *ctx, const rtdd_solve_params *params, rtdd_solve_info *info);
/* the standalone third-party pieces, exposed for parity tests against the oracle's restatement */
int rtdd_bgr2gray(rtdd_ctx *ctx, const uint8_t *bgr, size_t bgrPitch, uint8_t *gray, size_t grayPitch, int rows, int cols);
int rtdd_pyrdown_gray(rtdd_ctx *ctx, const uint8_t *src, size_t srcPitch, int rows, int cols, uint8_t *dst, size_t dstPitch);
int rtdd_pyrup_depth(rtdd_ctx *ctx, const float *src, size_t srcPitch, int rows, int cols,
                     float *dst, size_t dstPitch, int dstRows, int dstCols);
int rtdd_depth_to_u8(rtdd_ctx *ctx, const float *src, size_t srcPitch, uint8_t *dst, size_t dstPitch, int rows, int cols);
/* pitched host<->device copies on the context's stream, then a stream sync (harness / binding convenience).  Any host pitch: a
 * contiguous host image whose pitch is no multiple of four (an odd-width cv::Mat) goes as one linear copy through a device buffer --
 * the runtime's own 2-D copy takes ~9 us per row for such a pitch.  rtdd_live_submit's copies do the same. */
int rtdd_upload(rtdd_ctx *ctx, void *dev, size_t devPitch, const void *host, size_t hostPitch, size_t widthBytes, int rows);
int rtdd_download(rtdd_ctx *ctx, void *host, size_t hostPitch, const void *dev, size_t devPitch, size_t widthBytes, int rows);

/* ---- instrumentation ------------------------------------------------------------------------ */

/* Device time of the solver's phases, measured with HIP events recorded on the context's stream around
 * them (enable with rtdd_profile_enable(ctx, 1)).  Recording does not synchronise; rtdd_profile_get waits
 * for the recorded calls and returns TOTALS over the solve calls made since the previous rtdd_profile_get
 * (at most the last 64).  launches = sweep-kernel launches, sweeps = Jacobi sweeps they performed. */
typedef struct rtdd_profile {
    double sweep_ms;
    int launches;
    int sweeps;
    double prepare_ms;              /* edge-weight + staging pass */
    double finish_ms;               /* copy back to the caller's pitched buffer */
} rtdd_profile;
int rtdd_profile_enable(rtdd_ctx *ctx, int on);
int rtdd_profile_get(rtdd_ctx *ctx, rtdd_profile *out);

#ifdef __cplusplus
}
#endif
#endif /* RTDD_H */
