// rtdd_internal.hpp -- private declarations shared by the translation units of librtdd.so.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "rtdd.h"

namespace rtdd {

// ---- internal plane geometry ------------------------------------------------------------------
// Solver state lives in dense private planes (x_k, x_{k-1}/x_{k+1}, packed per-pixel metadata),
// NOT in the caller's pitched buffers.  A plane has `ip` elements per row with at least
// kGuardCols unused elements after the last image column (they double as the left guard of the
// next row) and kGuardRows unused rows before row 0 and after the last row, so stencil/halo
// loads never need bounds checks on the ADDRESS -- only on whether the value is used.
constexpr int kGuardRows = 16;
constexpr int kGuardCols = 64;
constexpr int kRowAlign = 64;       // ip is a multiple of 64 elements = 256 B

inline size_t plane_pitch(int cols) { return (size_t)((cols + kRowAlign - 1) / kRowAlign) * kRowAlign + kGuardCols; }
inline size_t plane_elems(int rows, int cols) { return plane_pitch(cols) * (size_t)(rows + 2 * kGuardRows) + 1024; }

// packed per-pixel metadata: bits 0-7 edge-weight index to the RIGHT neighbour, 8-15 index to the
// neighbour BELOW, bit 16 = Dirichlet (scribble == 255).  Left/up indices are the right/down
// indices of the left/upper neighbour (|a-b| and the depth gate are symmetric,
// /root/reference/src/GPUSolver.cu:188-218); "no neighbour" (256) follows from coordinates.
constexpr uint32_t kMetaDirichlet = 1u << 16;

struct Level {
    int rows = 0, cols = 0;         // allocation size (upper bound for solve calls)
    size_t elems = 0;               // elements per plane
    float *plane[4] = {nullptr, nullptr, nullptr, nullptr};   // raw allocations (4 f32 planes)
    uint32_t *meta = nullptr;
    // views at row 0 (after the guard rows)
    float *P(int i, size_t ip) const { return plane[i] + (size_t)kGuardRows * ip; }
    uint32_t *M(size_t ip) const { return meta + (size_t)kGuardRows * ip; }
    // A context created for a batch of images (rtdd_pyramid_create_batch) allocates every plane `images` times over, image b's copy
    // `elems` elements behind image b - 1's: view(b) is the level as image b sees it.
    Level view(int b) const {
        Level v = *this;
        for (auto &q : v.plane) if (q) q += (size_t)b * elems;
        if (v.meta) v.meta += (size_t)b * elems;
        return v;
    }
};

// The images a batched launch covers (rtdd_estimate_depth_batch; BASELINE configs[3]: independent images on one GPU): blockIdx.z = image
// index - first.  Every kernel on the estimate's path takes, next to each image pointer, the BYTE stride between consecutive images of
// that argument (0 and gridDim.z = 1 for everything else).  Planes of the solver: Level::elems * 4.
struct Batch {
    int n = 1;                            // images per launch
    int first = 0;                        // the first one's index in the context's batched allocations
    size_t depth = 0, scribble = 0, gray = 0, u8 = 0;     // byte strides of the caller-side arguments of the solve in progress
};
#define RTDD_Z(ptr, stride) ptr = (decltype(ptr))((const char *)(ptr) + (size_t)blockIdx.z * (size_t)(stride))

// What a solve is told BESIDES the reference's own arguments (GPUMatrixFreeSolver's, src/GPUSolver.cu:274-275).  Passed down the call
// chain by value or const reference -- estimate_levels -> solve_with -> the launchers -- and stored by value in the pending-call log
// (PendingOp): nothing is parked in the context around a call, so a replay restores nothing by hand.
struct SolveTargets {
    Batch batch;                                    // the images the launches cover (n = 1, first = 0: one image)
    bool defer_finish = false;                      // leave the result in its plane: the next level's pyrUp kernel reads it there
    uint8_t *u8 = nullptr; size_t u8_pitch = 0;     // the copy-back kernel also writes the u8 map here (src/main.cpp:290) ...
    uint8_t *u8b = nullptr; size_t u8b_pitch = 0;   // ... and a second copy here (a live frame's staging slot or the host's buffer)
    bool logged = true;                             // false: a step of a larger logged call (an estimate's per-level solves)
};
// What a solve hands back to a caller inside the library: its sequence number (what the guarded copy-back kernels report) and the
// plane its result is in (defer_finish).
struct SolveOutcome { int seq = 0, plane = -1; };
// A live frame's own images (cascade_api.cpp): the level-0 annotation pair its estimate ran on, the second target of its u8 map
// (u8_pitch 0: a staging slot with the pyramid's pitch), and the sticky depth effect behind it (src/main.cpp:190-230) with its target.
struct LiveTargets {
    void *scribble = nullptr, *edited = nullptr;
    uint8_t *u8 = nullptr; size_t u8_pitch = 0;
    int effect = 0;                                 // RTDD_EFFECT_*
    uint8_t *artistic = nullptr; size_t artistic_pitch = 0;     // device image the effect writes (the frame's staging slot)
};

struct Options {
    int fp_contract = 1;
    int sweep_kernel = 0;
    int temporal_depth = 0;
    int rows_per_wave = 0;
    int tile = 0;
    int persistent = 1;
    int defocus_path = 0;                // RTDD_OPT_DEFOCUS_PATH: 0 automatic, 1 global summed-area table, 2 per-tile tables in LDS where they fit
    // RTDD_METHOD_AUTO's cost model (constants, not clocks: a solve is reproducible).  Readable and settable as options so that
    // what decided a solve can be restated from outside (tests/test_gpu_multigrid.py).
    int auto_cycle_fixed_ns = 270000;    // a V-cycle with its residual check: launch-bound part ...
    int auto_cycle_fs_per_px = 46000;    // ... plus streaming part per level-0 pixel (femtoseconds)
    int auto_sweep_fs_per_px = 1429;     // one red-black sweep per pixel (1 / 700 Gpx-sweeps/s) ...
    int auto_sweep_floor_ns = 2500;      // ... but never below one launch-bound sweep
    int debug_withhold_tile = 0;     // RTDD_OPT_DEBUG_WITHHOLD_TILE: tile number + 1 whose exchange flag is never published (0 = off)
    int debug_poll_limit_us = 0;     // RTDD_OPT_DEBUG_POLL_LIMIT_US: exchange poll limit (0 = default, 200 ms)
    int debug_force_status = 0;      // RTDD_OPT_DEBUG_FORCE_STATUS: one-shot value for the status word behind the next blocked launch
    int timeout_heal = 1;            // RTDD_OPT_TIMEOUT_HEAL: 1 a timed-out persistent launch is healed (calls logged, run again); 0 it is reported
    int annotation_lds = 1;          // RTDD_OPT_ANNOTATION_LDS: the annotation pyramid's chain of levels in LDS (pyramids of up to 6 levels); 0: through global memory
    int live_zero_copy = 1;          // RTDD_OPT_LIVE_ZERO_COPY: a live frame's u8 map is stored by the copy-back kernel straight into the host's page-locked buffer (0 never, 1 when no other frame is in flight, 2 always)
    int defocus_slice_mb = 0;        // RTDD_OPT_DEFOCUS_SLICE_MB: > 0: a summed-area table of more than twice this is built and looked up in slices of at most this size
    int defocus_strips = 0;          // RTDD_OPT_DEFOCUS_STRIPS: the table lookup's tile order -- 0 automatic, 1 row bands per XCD, 2 column strips per XCD
    int rearm_after = 64;            // RTDD_OPT_PERSISTENT_REARM_AFTER: solves without persistence after the first heal, doubling with every further one
};

// One asynchronous call whose results the caller has not yet seen confirmed by a synchronising call: what check_persistent_status
// needs to run it again when a persistent launch gave up (api.cpp, "self-healing").  A solve is one sequence number (handed to the
// kernel that publishes its result: k_finish, or k_pyrup_inject inside an estimate); an estimate is one per pyramid level.
struct PendingOp {
    enum Kind { kSolve = 0, kEstimate = 1, kDefocus = 2, kDesaturate = 3, kHaze = 4 } kind = kSolve;
    Options opt;                          // the options in force when the call was made
    // kSolve: the arguments of rtdd_solve_ex (+ the optional u8 copy of the result, rtdd_refine_depth)
    float *depth = nullptr; size_t depthPitch = 0;
    const uint8_t *scribble = nullptr; size_t scribblePitch = 0;
    const uint8_t *gray = nullptr; size_t grayPitch = 0;
    int rows = 0, cols = 0, level = 0, seq = 0;
    rtdd_solve_params params{};
    SolveTargets targets;                 // (rtdd_refine_depth: the selected image of a batch, the u8 copy of the result)
    // kEstimate
    int maxIterations = 0;
    int level_seq[32] = {};               // sequence number of level l's solve (0: the level is empty)
    int batch_first = 0, batch_n = 1;     // the images of the context's batched pyramid the estimate covers
    LiveTargets live;                     // a live frame: its annotation pair, its map's second target, its effect (scribble == nullptr: not one)
    unsigned long long id = 0;            // position in the context's call order (live mode drops the confirmed prefix of the log)
    // kDefocus / kDesaturate / kHaze: a depth effect queued BEHIND an unconfirmed solve (it may have read that solve's input instead of
    // its result); `depth` / `depthPitch` / `gray` / `grayPitch` / `rows` / `cols` above, and:
    const uint8_t *original = nullptr; size_t originalPitch = 0;
    uint8_t *artistic = nullptr; size_t artisticPitch = 0;
};
constexpr int kRestartSolve = -1000;      // internal status: the pending calls were healed inside a solve's residual check; that solve starts over
constexpr size_t kMaxPendingOps = 4096;
constexpr int kMaxRearms = 4;             // heals after which persistence stays off for the context's life

}  // namespace rtdd

namespace rtdd { struct Pyramid; struct MgState; }

namespace rtdd {
// A contiguous device buffer a host image with an unaligned pitch goes through (one per stream that copies: uses of one buffer are
// ordered by that stream).
struct Bounce { void *ptr = nullptr; size_t bytes = 0; };
}

struct rtdd_ctx {
    int device = 0;
    rtdd::Pyramid *pyr = nullptr;       // whole-estimate driver state (cascade_api.cpp)
    rtdd::MgState *mg = nullptr;        // multigrid hierarchy buffers (multigrid.hip), kept between solves of one size
    hipStream_t stream = nullptr;
    std::vector<rtdd::Level> levels;
    int alloc_images = 1;               // how many images' planes the NEXT rtdd_allocate makes every level hold (rtdd_pyramid_create_batch; one shot)
    int levels_images = 1;              // ... and how many the levels hold now
    int maxLevel = -1;
    bool weights_loaded = false;
    float lut_host[257];
    float *lut_dev = nullptr;
    float *omega_dev = nullptr;     // device copy of the omega schedule (temporally blocked kernel)
    int omega_cap = 0;
    float *residual_dev = nullptr;  // extension: residual reduction target
    int *sync_words = nullptr;      // control words of the persistent kernels (persist_sync.hpp); allocated with the context
    int defocus_last_path = 0;           // RTDD_OPT_DEFOCUS_LAST_PATH: what the most recent rtdd_simulate_defocus launched (1 table, 2 tile kernel)
    int defocus_last_slices = 0;         // ... and, on the table path, how many horizontal slices it built a table for (1: one whole-image table)
    bool defocus_band_sticky = false;    // a banded-table defocus met windows beyond a slice (depths above 255): one whole-image table from then on
    bool defocus_table_sticky = false;   // a tile-kernel defocus met out-of-range depths (seen at a synchronisation): automatic choice = the table from then on
    int flag_epoch = 0;             // the per-tile flags of the persistent kernels only ever grow: base value of the next persistent launch (api.cpp)
    int sync_header[2] = {0, 0};    // what sync_words[kSyncWithhold], [kSyncLimit] currently hold on the device
    int last_nominal_depth = 0;     // sweeps per launch / exchange the most recent blocked solve was configured with (its last launch may be shorter)
    int last_launch_images = 1;     // images each sweep launch of the most recent solve covered (sweep_blocked.hip: a batch in the same launches, or image after image)
    rtdd_solve_info last_info{};    // of the most recent solve; kernel/tile/temporal_depth/persistent are filled in by the sweep launchers
    // Self-healing after RTDD_ERR_TIMEOUT (api.cpp heal_pending): every solve / estimate since the last status check, in call order
    int solve_seq = 0;              // sequence number of the most recent rtdd_solve_ex (1 .. 2^30, never 0)
    std::vector<rtdd::PendingOp> pending;
    bool pending_overflow = false;  // more than kMaxPendingOps calls without a synchronisation: a timeout among them is reported, not healed
    bool healing = false;           // a replay is running: nothing is logged, a second timeout is final
    bool heal_warned = false;
    int heals = 0;                  // RTDD_OPT_TIMEOUT_HEALS
    // opt.persistent is what the launchers read; persistent_wanted is what the caller asked for.  A heal switches opt.persistent off and
    // suspends it for persist_suspend more solves (rearm_after, doubling with every heal; -1 after kMaxRearms heals: off for good)
    int persistent_wanted = 1;
    int persist_suspend = 0;        // RTDD_OPT_PERSISTENT_SUSPENDED
    int *confirm_host = nullptr;    // page-locked: sequence number of the latest solve whose copy-back kernel published its result (persist_sync.hpp)
    unsigned long long op_counter = 0;
    bool persistent_used = false;   // a launch that can set the status word (sync_words[kSyncStatus]) happened since the last status check
    // The copy-back kernels report in page-locked memory the sequence number of a solve they published WITH THE STATUS WORD CLEAR
    // (confirm_host).  If the newest such kernel launched (publish_seq) has reported, and nothing that can set a control word has been
    // launched behind it (status_writer_behind), a synchronised stream proves the words clear without reading them from the device: the
    // drop-in shim synchronises behind every pyramid level's solve, and the 32-byte blocking hipMemcpy was most of what that cost it.
    int publish_seq = 0;
    bool status_writer_behind = false;
    signed char persist_fit[17][2];  // per (tile id, contraction): does one workgroup of the persistent kernel fit a CU of THIS device (-1 = not asked yet)
    rtdd::Bounce bounce;            // host <-> device 2-D copies with an unaligned host pitch, on ctx->stream (copy_h2d / copy_d2h, cascade_api.cpp)
    uint32_t *sat = nullptr;        // defocus summed-area table scratch
    size_t sat_elems = 0;
    int sat_rows = 0, sat_cols = 0;     // the geometry the table's zero padding was laid out for (effect_kernels.hip)
    int num_cus = 256;
    rtdd::Options opt;
    bool profile_on = false;
    rtdd_profile prof{};
    // profiling: 4 events per solve call, a ring of kProfSlots calls; resolved lazily by rtdd_profile_get (no sync per call)
    static constexpr int kProfSlots = 64;
    hipEvent_t ev[4 * kProfSlots] = {};
    int prof_launches[kProfSlots] = {}, prof_sweeps[kProfSlots] = {};
    int prof_pending = 0;           // calls recorded since the last rtdd_profile_get
    std::string last_error;
};

namespace rtdd {

#ifdef __HIPCC__
// A thread's wave number within its workgroup as a SCALAR (every lane of a wave holds the same value; v_readfirstlane tells the compiler
// so): row indices and row pointers derived from it are then computed once per wave on the scalar unit.  Left as `threadIdx.x >> 6` they
// are per-lane values, and `(size_t)y * pitch` becomes two v_mul_lo_u32 and a v_mad_u64_u32 per row pointer -- quarter-rate
// instructions: 17 of them in the desaturation kernel, 16 in k_prepare4, 31 in k_pyrup_inject4 (round 4: all but one gone; the streaming
// kernels are memory-bound, so it bought them only 1-9 %: EXPERIMENTS.md).
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
#endif

int fail(rtdd_ctx *ctx, int status, const char *what, hipError_t e = hipSuccess);
// a launch that can set one of the control words (sync_words) has been queued: the next synchronising call must look at them
inline void note_status_writer(rtdd_ctx *ctx) { ctx->persistent_used = true; ctx->status_writer_behind = true; }
// a guarded copy-back kernel (k_finish, k_pyrup_inject) for solve `seq` has been queued: it reports `seq` if it finds the words clear
inline void note_publisher(rtdd_ctx *ctx, int seq) { ctx->persistent_used = true; ctx->publish_seq = seq; ctx->status_writer_behind = false; }

#define RTDD_HIP(ctx, call)                                                        \
    do {                                                                           \
        hipError_t e_ = (call);                                                    \
        if (e_ != hipSuccess) return ::rtdd::fail((ctx), RTDD_ERR_HIP, #call, e_); \
    } while (0)

#define RTDD_LAUNCH_CHECK(ctx, name)                                                     \
    do {                                                                                 \
        hipError_t e_ = hipGetLastError();                                               \
        if (e_ != hipSuccess) return ::rtdd::fail((ctx), RTDD_ERR_HIP, "launch " name, e_); \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
        else prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// ---- solver_kernels.hip -------------------------------------------------------------------------
// (B: the images a batched launch covers; L is image B.first's view of the level)
int launch_prepare(rtdd_ctx *ctx, const Level &L, size_t ip, const float *depth, size_t depthPitch,
                   const uint8_t *scribble, size_t scribblePitch, const uint8_t *gray, size_t grayPitch,
                   int rows, int cols, int level, const Batch &B);
// Both sweep launchers advance n sweeps from (plane *pk = x_k, plane *pm = x_{k-1}) and update *pk / *pm to
// the planes holding x_{k+n} / x_{k+n-1}.
int launch_sweeps(rtdd_ctx *ctx, const Level &L, size_t ip, int rows, int cols, const float *omegas_host, int n,
                  int *pk, int *pm, int *launches);
// sweep_blocked.hip
int launch_sweeps_blocked(rtdd_ctx *ctx, const Level &L, size_t ip, int rows, int cols, const float *omegas_dev, int n,
                          int *pk, int *pm, int *launches, int images = 1);
// (k_finish and k_pyrup_inject store NOTHING when the status word is set: a timed-out solve leaves the caller's buffers as they were;
// the first of them to find it set records ctx->guard_seq in sync_words[kSyncFailedSeq])
// (seq: the solve's sequence number, reported by the kernel either as confirmed or as the first failed one; t: the batch and the u8 targets)
int launch_finish(rtdd_ctx *ctx, const Level &L, size_t ip, int src_plane, float *depth, size_t depthPitch, int rows, int cols,
                  const SolveTargets &t, int seq);
int launch_index_to_weight(rtdd_ctx *ctx, const uint8_t *gray, size_t grayPitch, const float *depth, size_t depthPitch,
                           int32_t *index2, int level, int rows, int cols);
int launch_residual(rtdd_ctx *ctx, const Level &L, size_t ip, int plane, int rows, int cols, float *host_out);
int launch_rbgs(rtdd_ctx *ctx, const Level &L, size_t ip, int plane, int rows, int cols, int nsweeps, float omega);
// ---- multigrid.hip ------------------------------------------------------------------------------
int launch_multigrid(rtdd_ctx *ctx, const Level &L0, size_t ip, int rows, int cols, int max_cycles, float tolerance, int check_every, double alternative_seconds,
                     double cycle_seconds, int *plane, int *cycles_done, float *residual, int *launches);
void mg_release(rtdd_ctx *ctx);
int mg_download(rtdd_ctx *ctx, int level, int which, float *host, int *rows, int *cols);

int launch_rbgs_blocked(rtdd_ctx *ctx, const Level &L, size_t ip, int rows, int cols, int n, float omega, int *plane, int *launches, int keep = -1);

// ---- image_kernels.hip --------------------------------------------------------------------------
// (images, z*: a batched launch over `images` images whose arguments lie z* bytes apart)
int launch_convert(rtdd_ctx *ctx, const uint8_t *src, size_t srcPitch, float *dst, size_t dstPitch,
                   const uint8_t *mask, size_t maskPitch, int rows, int cols, int images = 1, size_t zSrc = 0, size_t zDst = 0, size_t zMask = 0);
int launch_pyrdown_annotation(rtdd_ctx *ctx, const uint8_t *ps, size_t psp, const uint8_t *pe, size_t pep, int prows, int pcols,
                              uint8_t *cs, size_t csp, uint8_t *ce, size_t cep, int crows, int ccols,
                              int images = 1, size_t zPs = 0, size_t zPe = 0, size_t zCs = 0, size_t zCe = 0);
// the annotation pyramid of an estimate (levels 1 .. levels-1 from level 0) and the coarsest level's injection, one launch (image_kernels.hip)
int launch_annotation_pyramid(rtdd_ctx *ctx, int levels, uint8_t *const *scribble, const size_t *sp, const size_t *zs, uint8_t *const *edited, const size_t *ep, const size_t *ze,
                              const int *rows, const int *cols, float *depth, size_t dp, size_t zd, int images);
int launch_repitch(rtdd_ctx *ctx, hipStream_t stream, const void *src, size_t srcPitch, void *dst, size_t dstPitch, size_t widthBytes, int rows);
int copy_h2d(rtdd_ctx *ctx, Bounce &b, void *dev, size_t devPitch, const void *host, size_t hostPitch, size_t widthBytes, int rows, hipStream_t stream);
int copy_d2h(rtdd_ctx *ctx, Bounce &b, void *host, size_t hostPitch, const void *dev, size_t devPitch, size_t widthBytes, int rows, hipStream_t stream);
int launch_paint(rtdd_ctx *ctx, int x, int y, int color, int radius, uint8_t *edited, size_t editedPitch,
                 uint8_t *scribble, size_t scribblePitch, int rows, int cols);

// ---- effect_kernels.hip -------------------------------------------------------------------------
int launch_desaturate(rtdd_ctx *ctx, const uint8_t *orig, size_t op, const uint8_t *gray, size_t gp, const float *depth, size_t dp,
                      uint8_t *art, size_t ap, int rows, int cols);
int launch_haze(rtdd_ctx *ctx, const uint8_t *orig, size_t op, const float *depth, size_t dp, uint8_t *art, size_t ap, int rows, int cols);
int launch_defocus(rtdd_ctx *ctx, const uint8_t *orig, size_t op, const float *depth, size_t dp, uint8_t *art, size_t ap, int rows, int cols);

// ---- cascade.hip -------------------------------------------------------------------------------
int launch_bgr2gray(rtdd_ctx *ctx, const uint8_t *bgr, size_t bp, uint8_t *gray, size_t gp, int rows, int cols);
int launch_pyrdown_u8(rtdd_ctx *ctx, const uint8_t *src, size_t sp, int rows, int cols, uint8_t *dst, size_t dp);
struct PyrupBatch { int n = 1; size_t src = 0, dst = 0, edited = 0, mask = 0, coarse = 0; };      // images and byte strides of a batched pyrUp
int launch_pyrup_inject(rtdd_ctx *ctx, const float *src, size_t sp, int rows, int cols, float *dst, size_t dp, int drows, int dcols,
                        const uint8_t *edited, size_t ep, const uint8_t *mask, size_t mp, float *coarse_out = nullptr, size_t cp = 0,
                        int guard_seq = 0 /* != 0: guarded like k_finish, reporting this sequence number */, const PyrupBatch *batch = nullptr);
int launch_depth_to_u8(rtdd_ctx *ctx, const float *src, size_t sp, uint8_t *dst, size_t dp, int rows, int cols);
int launch_decode_annotation(rtdd_ctx *ctx, const uint8_t *bgr, size_t bp, const uint8_t *ann, size_t ap, uint8_t *edited, size_t ep,
                             uint8_t *scribble, size_t sp, int rows, int cols);
int launch_fill_f32(rtdd_ctx *ctx, float *dst, size_t dp, int rows, int cols, float v);
void pyramid_free(rtdd_ctx *ctx);
// an annotation image is about to be written: one of the pyramid's?  RTDD_ERR_STATE for a level-0 pointer a live frame has retired
int pyramid_note_write(rtdd_ctx *ctx, const void *scribble, const void *edited);
int pyramid_check_read(rtdd_ctx *ctx, const void *a, const void *b);     // ... about to be read

// persistent kernels (persist_sync.hpp): reserve the launch's flag values and refresh the debug words before a persistent launch;
// read the status word where the stream has just been synchronised (-> RTDD_ERR_TIMEOUT, status cleared)
int prepare_persistent_launch(rtdd_ctx *ctx, int nblocks, int *flag_base);
// in_solve: called from a residual check inside rtdd_solve_ex -- after a successful heal of the calls before it that solve starts over (kRestartSolve)
int check_persistent_status(rtdd_ctx *ctx, bool in_solve = false);
void prune_confirmed(rtdd_ctx *ctx);    // drop the logged calls a copy-back kernel has confirmed (no synchronisation)
int settle_pending(rtdd_ctx *ctx);      // before a call changes what the logged calls ran on: synchronise + check (+ heal) while that state still exists
// cascade_api.cpp: levels from_level .. 0 of an estimate (src/main.cpp:261-291); level_seq (optional) receives each level's solve sequence number
int estimate_levels(rtdd_ctx *ctx, int maxIterations, int from_level, int *level_seq, int first, int n, const LiveTargets &live);
// api.cpp: rtdd_solve_ex with everything the library's own callers add to it
int solve_with(rtdd_ctx *ctx, float *depth, size_t depthPitch, const uint8_t *scribble, size_t scribblePitch, const uint8_t *gray, size_t grayPitch,
               int rows, int cols, int level, const rtdd_solve_params *params, rtdd_solve_info *info, const SolveTargets &t, SolveOutcome *out);
// cascade_api.cpp: a live frame's effect (again, on a replay): RTDD_EFFECT_* on the pyramid's level-0 images into live.artistic
int live_effect(rtdd_ctx *ctx, const LiveTargets &live);
// an estimate of the pending log again, from the level whose solve has sequence number failed_seq (0: every level)
int estimate_replay(rtdd_ctx *ctx, const PendingOp &op, int failed_seq);

// the reference's host-side omega recurrence (src/GPUSolver.cu:282-299)
void omega_schedule(int n, std::vector<float> &out);

}  // namespace rtdd
