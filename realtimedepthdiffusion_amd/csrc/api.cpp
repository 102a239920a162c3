// api.cpp -- the extern "C" surface of librtdd.so (include/rtdd.h): context management, argument
// validation, the per-level solve driver (GPUMatrixFreeSolver, /root/reference/src/GPUSolver.cu:274-316)
// and thin forwards to the kernel launchers.  Host code only; kernels live in the *.hip files.
#include <cmath>
#include <cstring>
#include <new>

#include "rtdd_internal.hpp"
#include "persist_sync.hpp"

namespace rtdd {

int fail(rtdd_ctx *ctx, int status, const char *what, hipError_t e) {
    if (ctx) {
        ctx->last_error = what ? what : "";
        if (e != hipSuccess) {
            ctx->last_error += ": ";
            ctx->last_error += hipGetErrorString(e);
        }
    }
    return status;
}

// omega recurrence of the reference driver (src/GPUSolver.cu:282-299): float state, double
// intermediates, S = 10, rho = 0.99f.
void omega_schedule(int n, std::vector<float> &out) {
    out.resize(n > 0 ? n : 0);
    const int S = 10;
    float omega = 0.0f;
    const float rho = 0.99;
    for (int it = 0; it < n; it++) {
        if (it < S) omega = 1;
        else if (it == S) omega = 2.0 / (2.0 - rho * rho);
        else omega = 4.0 / (4.0 - rho * rho * omega);
        out[it] = omega;
    }
}

// The per-tile flags are never reset between launches: a persistent launch with `nblocks` blocks is handed the base value
// *flag_base = the context's running epoch, its workgroups publish and wait for flag_base + 1 .. flag_base + nblocks - 1, and the epoch
// advances past them.  Launches of one context are stream-ordered, so every flag a launch finds is below its base.  (Round 2 zeroed
// the 1024 flags with a hipMemsetAsync in front of every persistent launch: a ~5 us fill kernel per pyramid level and per solve.)
int prepare_persistent_launch(rtdd_ctx *ctx, int nblocks, int *flag_base) {
    if (ctx->flag_epoch > (1 << 30) - nblocks - 2) {                  // (once in ~10^7 solves) start over
        RTDD_HIP(ctx, hipMemsetAsync(ctx->sync_words + kSyncFlags, 0, (size_t)kSyncMaxTiles * kSyncFlagStride * sizeof(int), ctx->stream));
        ctx->flag_epoch = 0;
    }
    // (a launch's workgroups announce themselves with its base value: never the zero the flags start from)
    if (ctx->flag_epoch == 0) ctx->flag_epoch = 1;
    *flag_base = ctx->flag_epoch;
    ctx->flag_epoch += nblocks + 1;
    const int limit = ctx->opt.debug_poll_limit_us > 0 ? ctx->opt.debug_poll_limit_us * 100 : 0;        // 10 ns ticks
    if (ctx->sync_header[0] != ctx->opt.debug_withhold_tile || ctx->sync_header[1] != limit) {
        RTDD_HIP(ctx, hipMemsetD32Async((hipDeviceptr_t)(ctx->sync_words + kSyncWithhold), ctx->opt.debug_withhold_tile, 1, ctx->stream));
        RTDD_HIP(ctx, hipMemsetD32Async((hipDeviceptr_t)(ctx->sync_words + kSyncLimit), limit, 1, ctx->stream));
        ctx->sync_header[0] = ctx->opt.debug_withhold_tile; ctx->sync_header[1] = limit;
    }
    note_status_writer(ctx);
    return RTDD_OK;
}

// ---- self-healing ---------------------------------------------------------------------------------------------------------------
// The reference's GPUMatrixFreeSolver always leaves a valid depth map behind (src/GPUSolver.cu:311-314), and an unchanged main.cpp can
// neither set options nor upload its input again.  A persistent launch that is not fully co-resident (a shared GPU) gives up after its
// poll limit and sets the status word; from then on k_finish / k_pyrup_inject store nothing (persist_sync.hpp solve_is_dead), so every
// call made since keeps its INPUT, and the first of them has left its sequence number in sync_words[kSyncFailedSeq].  The next call that
// synchronises finds the word, switches persistence off for the rest of the context's life (one warning on stderr), runs the logged
// calls again from the failed one on, one launch per block of sweeps, and only then returns -- RTDD_OK, with the results the calls
// would have produced.  Status 2 (a wave waiting for a wave of its own workgroup: a protocol bug, not a scheduling accident) and a
// second failure during the replay are reported as RTDD_ERR_TIMEOUT as before.
static int replay(rtdd_ctx *ctx, const PendingOp &op, int failed_seq) {
    const Options now = ctx->opt;
    ctx->opt = op.opt;
    ctx->opt.persistent = 0;
    ctx->opt.debug_force_status = op.opt.debug_force_status == 3 ? 1 : 0;      // (3: the testing aid that makes the REPLAY fail as well)
    int rc = RTDD_OK;
    if (op.kind == PendingOp::kSolve) {
        rc = solve_with(ctx, op.depth, op.depthPitch, op.scribble, op.scribblePitch, op.gray, op.grayPitch, op.rows, op.cols, op.level,
                        &op.params, nullptr, op.targets, nullptr);
    } else if (op.kind == PendingOp::kEstimate) {
        rc = estimate_replay(ctx, op, failed_seq);
    } else if (op.kind == PendingOp::kDefocus) {
        rc = launch_defocus(ctx, op.original, op.originalPitch, op.depth, op.depthPitch, op.artistic, op.artisticPitch, op.rows, op.cols);
    } else if (op.kind == PendingOp::kDesaturate) {
        rc = launch_desaturate(ctx, op.original, op.originalPitch, op.gray, op.grayPitch, op.depth, op.depthPitch, op.artistic,
            op.artisticPitch, op.rows, op.cols);
    } else {
        rc = launch_haze(ctx, op.original, op.originalPitch, op.depth, op.depthPitch, op.artistic, op.artisticPitch, op.rows, op.cols);
    }
    ctx->opt = now;
    return rc;
}

// A depth effect queued behind solves that no synchronising call has confirmed yet is logged with them: should one of those solves turn
// out to have timed out, the effect ran on its INPUT and is run again behind the replayed solve.  (Nothing unconfirmed: nothing to log.)
static void log_effect(rtdd_ctx *ctx, PendingOp::Kind kind, const uint8_t *original, size_t originalPitch, const uint8_t *gray,
    size_t grayPitch,
                       const float *depth, size_t depthPitch, uint8_t *artistic, size_t artisticPitch, int rows, int cols) {
    if (ctx->healing || ctx->pending.empty()) return;
    prune_confirmed(ctx);
    if (ctx->pending.empty() || ctx->pending.size() >= kMaxPendingOps) return;
    PendingOp op;
    op.kind = kind; op.opt = ctx->opt; op.id = ++ctx->op_counter;
    op.original = original; op.originalPitch = originalPitch; op.gray = gray; op.grayPitch = grayPitch;
    op.depth = const_cast<float *>(depth); op.depthPitch = depthPitch; op.artistic = artistic; op.artisticPitch = artisticPitch;
    op.rows = rows; op.cols = cols;
    ctx->pending.push_back(op);
}

// Sequence number of the kernel that publishes the LAST result of a logged call (0: the call publishes no solve).
static int last_seq(const PendingOp &op) {
    if (op.kind == PendingOp::kSolve) return op.seq;
    if (op.kind != PendingOp::kEstimate) return 0;
    int m = 0;
    for (int l = 0; l < 32; l++) if (op.level_seq[l] > m) m = op.level_seq[l];
    return m;
}

// The copy-back kernels report, in page-locked memory, the sequence number of the latest solve whose result they published while the
// status word was clear (persist_sync.hpp solve_is_dead).  Every logged call up to and including that solve -- the effects queued in
// front of it too: they ran behind solves that had succeeded -- can never be asked for again, so it leaves the log here, without any
// synchronisation: the log holds the calls still in flight (plus the effects behind the last solve), not everything since the last
// rtdd_ctx_synchronize, and the caller's pointers are kept no longer than any asynchronous call keeps them.
void prune_confirmed(rtdd_ctx *ctx) {
    if (!ctx->confirm_host || ctx->pending.empty() || ctx->healing) return;
    const int confirmed = *(volatile int *)ctx->confirm_host;
    size_t n = 0;
    for (size_t i = 0; i < ctx->pending.size(); i++) {
        const int s = last_seq(ctx->pending[i]);
        if (s != 0 && s <= confirmed) n = i + 1;
    }
    if (n) ctx->pending.erase(ctx->pending.begin(), ctx->pending.begin() + n);
}

static bool op_holds(const PendingOp &op, int seq) {
    if (op.kind == PendingOp::kSolve) return op.seq == seq;
    if (op.kind != PendingOp::kEstimate) return false;
    for (int l = 0; l < 32; l++) if (op.level_seq[l] != 0 && op.level_seq[l] == seq) return true;
    return false;
}

static const char *kTimeoutText =
    "persistent sweep kernel: a workgroup timed out waiting for a neighbouring tile (its workgroups were not all "
                                  "co-resident: is the GPU shared?)";

// The stream has just been synchronised by the caller.  A blocked-sweep launch since the last check may have given up (persist_sync.hpp).
int check_persistent_status(rtdd_ctx *ctx, bool in_solve) {
    if (!ctx->persistent_used || !ctx->sync_words) {
        if (!ctx->healing) { ctx->pending.clear(); ctx->pending_overflow = false; }
        return RTDD_OK;
    }
    // The newest guarded copy-back kernel has reported its solve published with the status word clear, and nothing that could set a
    // control word was queued behind it: the words are clear (they are sticky, and that kernel ran behind every launch that could have
    // set them) -- no need to read them back.
    if (!ctx->healing && !ctx->status_writer_behind && ctx->publish_seq != 0 && ctx->confirm_host &&
        *(volatile int *)ctx->confirm_host == ctx->publish_seq) {
        ctx->persistent_used = false;
        ctx->pending.clear(); ctx->pending_overflow = false;
        return RTDD_OK;
    }
    int words[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    RTDD_HIP(ctx, hipMemcpy(words, ctx->sync_words, sizeof(words), hipMemcpyDeviceToHost));
    ctx->persistent_used = false;
    const int status = words[kSyncStatus], failed_seq = words[kSyncFailedSeq];
    // a defocus kernel summed windows by hand: not a depth map -- bit 0: the tile kernel, the table path
    if (words[kSyncNonLocal] != 0) {
                                                   // from now on; bit 1: a banded table, one whole-image table from now on
        if (words[kSyncNonLocal] & 1) ctx->defocus_table_sticky = true;
        if (words[kSyncNonLocal] & 2) ctx->defocus_band_sticky = true;
        RTDD_HIP(ctx, hipMemset(ctx->sync_words + kSyncNonLocal, 0, sizeof(int)));
    }
    if (status == 0) { if (!ctx->healing) { ctx->pending.clear(); ctx->pending_overflow = false; } return RTDD_OK; }
    RTDD_HIP(ctx, hipMemset(ctx->sync_words + kSyncStatus, 0, sizeof(int)));
    RTDD_HIP(ctx, hipMemset(ctx->sync_words + kSyncFailedSeq, 0, sizeof(int)));
    if (status != 1) {
        ctx->pending.clear(); ctx->pending_overflow = false;
        return fail(ctx, RTDD_ERR_TIMEOUT,
            "blocked sweep kernel: a wave timed out waiting for a neighbouring wave of its own workgroup (internal error); "
                                           "the results since the last synchronisation are invalid");
    }
    // Persistence off, and suspended: rearm_after solves after the first heal, twice as many after every further one, for good after
    // kMaxRearms heals (an unchanged main.cpp on the drop-in shim can set no option: one scheduling accident on a shared GPU must not
    // cost it the persistent kernel until exit, and a GPU that stays shared must not cost it a 200 ms stall every few frames).
    if (!ctx->healing) {                            // (a second time-out while the calls are being run again is part of the same event)
        ctx->opt.persistent = 0;
        ctx->heals++;
        if (ctx->heals > kMaxRearms || ctx->opt.rearm_after <= 0) ctx->persist_suspend = -1;
        else {
            const long long n = (long long)ctx->opt.rearm_after << (ctx->heals - 1);
            ctx->persist_suspend = n > (1 << 30) ? (1 << 30) : (int)n;
        }
    }
    if (ctx->healing || ctx->pending_overflow || !ctx->opt.timeout_heal) {
        std::string msg = kTimeoutText;
        msg += ctx->healing ? "; it happened again while the calls were being run again without persistence"
             : !ctx->opt.timeout_heal ? "; RTDD_OPT_TIMEOUT_HEAL is 0, so nothing was run again"
                 : "; too many calls were queued without a synchronisation to run them again";
        msg += "; the results since the last synchronisation are invalid";
        ctx->pending.clear(); ctx->pending_overflow = false;
        return fail(ctx, RTDD_ERR_TIMEOUT, msg.c_str());
    }
    // heal: the logged calls again from the first failed one
    if (!ctx->heal_warned) {
        ctx->heal_warned = true;
        std::fprintf(stderr,
                     "rtdd: %s; running the affected calls again one launch per block of sweeps -- persistent launches are suspended "
                     "for this context's next %d solves (twice as long after every further time-out, for good after %d)\n",
                     kTimeoutText, ctx->persist_suspend, kMaxRearms);
    }
    std::vector<PendingOp> ops;
    ops.swap(ctx->pending);
    size_t first = ops.size();                      // failed_seq == 0: every logged call had published its result before the word was set
    if (failed_seq != 0) {
        for (size_t i = 0; i < ops.size(); i++) if (op_holds(ops[i], failed_seq)) { first = i; break; }
        if (first == ops.size()) return fail(ctx, RTDD_ERR_TIMEOUT,
            "persistent sweep kernel timed out and the failed call is not among the logged ones; the results since the last "
            "synchronisation are invalid");
    }
    ctx->healing = true;
    int rc = RTDD_OK;
    for (size_t i = first; i < ops.size() && rc == RTDD_OK; i++) rc = replay(ctx, ops[i], i == first ? failed_seq : 0);
    if (rc == RTDD_OK) {
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = fail(ctx, RTDD_ERR_HIP, "hipStreamSynchronize (replay)", e);
        else { note_status_writer(ctx); rc = check_persistent_status(ctx); }
    }
    ctx->healing = false;
    if (rc != RTDD_OK) return rc;
    return in_solve ? kRestartSolve : RTDD_OK;
}

// Calls that change what a logged solve / estimate would run on (the level planes, the weight table, the pyramid's images) first
// settle the log: synchronise and look at the status word while the state the logged calls were made against still exists.
int settle_pending(rtdd_ctx *ctx) {
    if (ctx->pending.empty() || ctx->healing) return RTDD_OK;
    RTDD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return check_persistent_status(ctx);
}

static void free_levels(rtdd_ctx *ctx) {
    for (auto &L : ctx->levels) {
        for (auto &p : L.plane)
            if (p) { (void)hipFree(p); p = nullptr; }
        if (L.meta) { (void)hipFree(L.meta); L.meta = nullptr; }
    }
    ctx->levels.clear();
    ctx->maxLevel = -1;
}

}  // namespace rtdd

using namespace rtdd;

#define REQUIRE(ctx, cond, msg) \
    do { if (!(cond)) return fail((ctx), RTDD_ERR_INVALID, msg); } while (0)

#pragma GCC visibility push(default)
extern "C" {

int rtdd_version(void) { return RTDD_VERSION; }

const char *rtdd_status_string(int s) {
    switch (s) {
        case RTDD_OK: return "ok";
        case RTDD_ERR_INVALID: return "invalid argument";
        case RTDD_ERR_STATE: return "call order violated";
        case RTDD_ERR_HIP: return "HIP runtime error";
        case RTDD_ERR_NOMEM: return "out of memory";
        case RTDD_ERR_NO_DEVICE: return "no usable HIP device (there is no CPU fallback)";
        case RTDD_ERR_TIMEOUT: return "persistent kernel timed out (workgroups not co-resident)";
        default: return "unknown status";
    }
}

const char *rtdd_last_error(rtdd_ctx *ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

int rtdd_ctx_create(int device, rtdd_ctx **out) {
    if (!out) return RTDD_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return RTDD_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return RTDD_ERR_INVALID;
    rtdd_ctx *ctx = new (std::nothrow) rtdd_ctx();
    if (!ctx) return RTDD_ERR_NOMEM;
    ctx->device = device;
    for (auto &t : ctx->persist_fit) t[0] = t[1] = -1;
    DeviceGuard g(device);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    bool ok = hipMalloc((void **)&ctx->lut_dev, 257 * sizeof(float)) == hipSuccess &&
              hipMalloc((void **)&ctx->residual_dev, 64) == hipSuccess &&
              hipMalloc((void **)&ctx->sync_words, kSyncWords * sizeof(int)) == hipSuccess &&
              hipMemset(ctx->sync_words, 0, kSyncWords * sizeof(int)) == hipSuccess;
    for (auto &e : ctx->ev) ok = ok && hipEventCreate(&e) == hipSuccess;
    // the word the copy-back kernels report confirmed solves in: page-locked host memory the device writes directly (persist_sync.hpp)
    if (ok && hipHostMalloc((void **)&ctx->confirm_host, 64, hipHostMallocMapped) == hipSuccess) {
        *ctx->confirm_host = 0;
        void *dev_view = nullptr;
        ok = hipHostGetDevicePointer(&dev_view, ctx->confirm_host, 0) == hipSuccess &&
             hipMemcpy(ctx->sync_words + kSyncConfirmPtr, &dev_view, sizeof(dev_view), hipMemcpyHostToDevice) == hipSuccess;
    } else ok = false;
    if (!ok) { rtdd_ctx_destroy(ctx); return RTDD_ERR_HIP; }
    *out = ctx;
    return RTDD_OK;
}

int rtdd_ctx_destroy(rtdd_ctx *ctx) {
    if (!ctx) return RTDD_OK;
    DeviceGuard g(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    pyramid_free(ctx);
    mg_release(ctx);
    free_levels(ctx);
    if (ctx->lut_dev) (void)hipFree(ctx->lut_dev);
    if (ctx->omega_dev) (void)hipFree(ctx->omega_dev);
    if (ctx->residual_dev) (void)hipFree(ctx->residual_dev);
    if (ctx->sync_words) (void)hipFree(ctx->sync_words);
    if (ctx->sat) (void)hipFree(ctx->sat);
    if (ctx->bounce.ptr) (void)hipFree(ctx->bounce.ptr);
    if (ctx->confirm_host) (void)hipHostFree(ctx->confirm_host);
    for (auto &e : ctx->ev) if (e) (void)hipEventDestroy(e);
    delete ctx;
    return RTDD_OK;
}

int rtdd_ctx_set_stream(rtdd_ctx *ctx, rtdd_stream stream) {
    if (!ctx) return RTDD_ERR_INVALID;
    DeviceGuard g(ctx->device);
    // the logged calls were queued on the OLD stream: confirm (or heal) them there before anything is queued on the new one
    { const int rc_ = settle_pending(ctx); if (rc_ != RTDD_OK) return rc_; }
    ctx->stream = (hipStream_t)stream;
    return RTDD_OK;
}

int rtdd_ctx_synchronize(rtdd_ctx *ctx) {
    if (!ctx) return RTDD_ERR_INVALID;
    DeviceGuard g(ctx->device);
    RTDD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // (a timed-out persistent launch is healed here: the affected calls run again, api.cpp above)
    return check_persistent_status(ctx);
}

int rtdd_set_option(rtdd_ctx *ctx, int key, int value) {
    if (!ctx) return RTDD_ERR_INVALID;
    switch (key) {
        case RTDD_OPT_FP_CONTRACT: ctx->opt.fp_contract = value ? 1 : 0; break;
        case RTDD_OPT_SWEEP_KERNEL: REQUIRE(ctx, value >= 0 && value <= 2, "sweep kernel must be 0..2"); ctx->opt.sweep_kernel = value;
        break;
        case RTDD_OPT_TEMPORAL_DEPTH: REQUIRE(ctx, value >= 0 && value <= 28, "temporal depth must be 0..28");
        ctx->opt.temporal_depth = value; break;
        case RTDD_OPT_ROWS_PER_WAVE: REQUIRE(ctx, value >= 0 && value <= 1024, "rows per wave must be 0..1024");
        ctx->opt.rows_per_wave = value; break;
        // (setting the automatic choice again forgets what earlier depths made it choose)
        case RTDD_OPT_DEFOCUS_PATH: REQUIRE(ctx, value >= 0 && value <= 2, "defocus path must be 0..2"); ctx->opt.defocus_path = value;
        if (value == 0) ctx->defocus_table_sticky = ctx->defocus_band_sticky = false; break;
        case RTDD_OPT_TILE: REQUIRE(ctx, value >= 0 && value <= 16, "tile must be 0..16"); ctx->opt.tile = value; break;
        // (said explicitly: armed at once, whatever a heal suspended)
        case RTDD_OPT_PERSISTENT: ctx->opt.persistent = ctx->persistent_wanted = value ? 1 : 0; ctx->persist_suspend = 0; break;
        case RTDD_OPT_ANNOTATION_LDS: ctx->opt.annotation_lds = value ? 1 : 0; break;
        case RTDD_OPT_LIVE_ZERO_COPY: REQUIRE(ctx, value >= 0 && value <= 2, "RTDD_OPT_LIVE_ZERO_COPY is 0, 1 or 2");
        ctx->opt.live_zero_copy = value; break;
        case RTDD_OPT_TIMEOUT_HEAL: ctx->opt.timeout_heal = value ? 1 : 0;
        if (!value && !ctx->healing) { ctx->pending.clear(); ctx->pending_overflow = false; } break;
        case RTDD_OPT_PERSISTENT_REARM_AFTER: REQUIRE(ctx, value >= 0 && value <= (1 << 20), "must be 0..2^20");
        ctx->opt.rearm_after = value; break;
        case RTDD_OPT_DEFOCUS_SLICE_MB: REQUIRE(ctx, value >= 0 && value <= 4095, "must be 0..4095 MB"); ctx->opt.defocus_slice_mb = value;
        break;
        case RTDD_OPT_DEFOCUS_STRIPS: REQUIRE(ctx, value >= 0 && value <= 2, "must be 0, 1 or 2"); ctx->opt.defocus_strips = value; break;
        case RTDD_OPT_AUTO_CYCLE_FIXED_NS: REQUIRE(ctx, value >= 0, "must be >= 0"); ctx->opt.auto_cycle_fixed_ns = value; break;
        case RTDD_OPT_AUTO_CYCLE_FS_PER_PX: REQUIRE(ctx, value >= 0, "must be >= 0"); ctx->opt.auto_cycle_fs_per_px = value; break;
        case RTDD_OPT_AUTO_SWEEP_FS_PER_PX: REQUIRE(ctx, value >= 0, "must be >= 0"); ctx->opt.auto_sweep_fs_per_px = value; break;
        case RTDD_OPT_AUTO_SWEEP_FLOOR_NS: REQUIRE(ctx, value >= 0, "must be >= 0"); ctx->opt.auto_sweep_floor_ns = value; break;
        case RTDD_OPT_DEBUG_WITHHOLD_TILE: REQUIRE(ctx, value >= 0 && value <= kSyncMaxTiles, "tile number + 1 out of range");
        ctx->opt.debug_withhold_tile = value; break;
        case RTDD_OPT_DEBUG_POLL_LIMIT_US: REQUIRE(ctx, value >= 0 && value <= 10000000, "poll limit must be 0..1e7 us");
        ctx->opt.debug_poll_limit_us = value; break;
        case RTDD_OPT_DEBUG_FORCE_STATUS: REQUIRE(ctx, value >= 0 && value <= 3, "status must be 0..3");
        ctx->opt.debug_force_status = value; break;
        default: return fail(ctx, RTDD_ERR_INVALID, "unknown option");
    }
    return RTDD_OK;
}

int rtdd_get_option(rtdd_ctx *ctx, int key, int *value) {
    if (!ctx || !value) return RTDD_ERR_INVALID;
    switch (key) {
        case RTDD_OPT_FP_CONTRACT: *value = ctx->opt.fp_contract; break;
        case RTDD_OPT_SWEEP_KERNEL: *value = ctx->opt.sweep_kernel; break;
        case RTDD_OPT_TEMPORAL_DEPTH: *value = ctx->opt.temporal_depth; break;
        case RTDD_OPT_ROWS_PER_WAVE: *value = ctx->opt.rows_per_wave; break;
        case RTDD_OPT_DEFOCUS_PATH: *value = ctx->opt.defocus_path; break;
        case RTDD_OPT_TILE: *value = ctx->opt.tile; break;
        case RTDD_OPT_PERSISTENT: *value = ctx->opt.persistent; break;
        case RTDD_OPT_AUTO_CYCLE_FIXED_NS: *value = ctx->opt.auto_cycle_fixed_ns; break;
        case RTDD_OPT_AUTO_CYCLE_FS_PER_PX: *value = ctx->opt.auto_cycle_fs_per_px; break;
        case RTDD_OPT_AUTO_SWEEP_FS_PER_PX: *value = ctx->opt.auto_sweep_fs_per_px; break;
        case RTDD_OPT_AUTO_SWEEP_FLOOR_NS: *value = ctx->opt.auto_sweep_floor_ns; break;
        case RTDD_OPT_DEBUG_WITHHOLD_TILE: *value = ctx->opt.debug_withhold_tile; break;
        case RTDD_OPT_DEBUG_POLL_LIMIT_US: *value = ctx->opt.debug_poll_limit_us; break;
        case RTDD_OPT_DEBUG_FORCE_STATUS: *value = ctx->opt.debug_force_status; break;
        case RTDD_OPT_TIMEOUT_HEALS: *value = ctx->heals; break;
        case RTDD_OPT_TIMEOUT_HEAL: *value = ctx->opt.timeout_heal; break;
        case RTDD_OPT_LIVE_ZERO_COPY: *value = ctx->opt.live_zero_copy; break;
        case RTDD_OPT_ANNOTATION_LDS: *value = ctx->opt.annotation_lds; break;
        case RTDD_OPT_PERSISTENT_REARM_AFTER: *value = ctx->opt.rearm_after; break;
        case RTDD_OPT_DEFOCUS_STRIPS: *value = ctx->opt.defocus_strips; break;
        case RTDD_OPT_DEFOCUS_SLICE_MB: *value = ctx->opt.defocus_slice_mb; break;
        case RTDD_OPT_DEFOCUS_LAST_SLICES: *value = ctx->defocus_last_slices; break;
        case RTDD_OPT_PERSISTENT_SUSPENDED: *value = ctx->persist_suspend; break;
        case RTDD_OPT_PENDING_CALLS: prune_confirmed(ctx); *value = (int)ctx->pending.size(); break;
        case RTDD_OPT_DEFOCUS_LAST_PATH: *value = ctx->defocus_last_path; break;
        default: return fail(ctx, RTDD_ERR_INVALID, "unknown option");
    }
    return RTDD_OK;
}

int rtdd_profile_enable(rtdd_ctx *ctx, int on) {
    if (!ctx) return RTDD_ERR_INVALID;
    ctx->profile_on = on != 0;
    ctx->prof_pending = 0;
    return RTDD_OK;
}

int rtdd_profile_get(rtdd_ctx *ctx, rtdd_profile *out) {
    if (!ctx || !out) return RTDD_ERR_INVALID;
    DeviceGuard g(ctx->device);
    // totals over the solve calls made since the previous rtdd_profile_get (at most the last kProfSlots of them)
    rtdd_profile p{};
    const int n = ctx->prof_pending < rtdd_ctx::kProfSlots ? ctx->prof_pending : rtdd_ctx::kProfSlots;
    for (int i = 0; i < n; i++) {
        const int slot = (ctx->prof_pending - 1 - i) % rtdd_ctx::kProfSlots;
        hipEvent_t *ev = ctx->ev + 4 * slot;
        RTDD_HIP(ctx, hipEventSynchronize(ev[3]));
        float a = 0, b = 0, c = 0;
        RTDD_HIP(ctx, hipEventElapsedTime(&a, ev[0], ev[1]));
        RTDD_HIP(ctx, hipEventElapsedTime(&b, ev[1], ev[2]));
        RTDD_HIP(ctx, hipEventElapsedTime(&c, ev[2], ev[3]));
        p.prepare_ms += a; p.sweep_ms += b; p.finish_ms += c;
        p.launches += ctx->prof_launches[slot]; p.sweeps += ctx->prof_sweeps[slot];
    }
    ctx->prof_pending = 0;
    ctx->prof = p;
    *out = p;
    return RTDD_OK;
}

// ---- solver ------------------------------------------------------------------------------------

int rtdd_allocate(rtdd_ctx *ctx, int rows, int cols, int levels) {
    if (!ctx) return RTDD_ERR_INVALID;
    REQUIRE(ctx, rows > 0 && cols > 0 && levels > 0 && levels <= 30, "rows, cols, levels must be positive");
    DeviceGuard g(ctx->device);
    { const int rc_ = settle_pending(ctx); if (rc_ != RTDD_OK) return rc_; }
    RTDD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    free_levels(ctx);
    const int images = ctx->alloc_images > 0 ? ctx->alloc_images : 1;
    ctx->alloc_images = 1;                         // (one shot: rtdd_pyramid_create_batch sets it for the call it makes)
    ctx->levels.resize(levels);
    for (int l = 0; l < levels; l++) {
        Level &L = ctx->levels[l];
        L.rows = (int)(rows / powf(2, l));          // src/GPUSolver.cu:42-43 (float divide, truncation)
        L.cols = (int)(cols / powf(2, l));
        L.elems = plane_elems(L.rows > 0 ? L.rows : 1, L.cols > 0 ? L.cols : 1);
        const size_t all = L.elems * (size_t)images;           // (a batched pyramid: every plane once per image, Level::view)
        for (auto &p : L.plane) {
            hipError_t e = hipMalloc((void **)&p, all * sizeof(float));
            if (e != hipSuccess) { free_levels(ctx); return fail(ctx, e == hipErrorOutOfMemory ? RTDD_ERR_NOMEM : RTDD_ERR_HIP,
                "hipMalloc(plane)", e); }
        }
        hipError_t e = hipMalloc((void **)&L.meta, all * sizeof(uint32_t));
        if (e != hipSuccess) { free_levels(ctx); return fail(ctx, e == hipErrorOutOfMemory ? RTDD_ERR_NOMEM : RTDD_ERR_HIP,
            "hipMalloc(meta)", e); }
        // guard cells are read (never used); give them a defined value once
        for (auto &p : L.plane) RTDD_HIP(ctx, hipMemsetAsync(p, 0, all * sizeof(float), ctx->stream));
        RTDD_HIP(ctx, hipMemsetAsync(L.meta, 0, all * sizeof(uint32_t), ctx->stream));
    }
    ctx->levels_images = images;
    ctx->maxLevel = levels - 1;                    // :51
    RTDD_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the reference syncs here (:52)
    return RTDD_OK;
}

int rtdd_free(rtdd_ctx *ctx) {
    if (!ctx) return RTDD_ERR_INVALID;
    DeviceGuard g(ctx->device);
    { const int rc_ = settle_pending(ctx); if (rc_ != RTDD_OK) return rc_; }
    RTDD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    mg_release(ctx);
    free_levels(ctx);
    return RTDD_OK;
}

int rtdd_load_weights(rtdd_ctx *ctx, float beta) {
    if (!ctx) return RTDD_ERR_INVALID;
    DeviceGuard g(ctx->device);
    { const int rc_ = settle_pending(ctx); if (rc_ != RTDD_OK) return rc_; }
    for (int w = 0; w < 256; w++) ctx->lut_host[w] = expf(-beta * w);      // src/GPUSolver.cu:267, host libm
    ctx->lut_host[256] = 0;
    RTDD_HIP(ctx, hipMemcpyAsync(ctx->lut_dev, ctx->lut_host, sizeof(ctx->lut_host), hipMemcpyHostToDevice, ctx->stream));
    RTDD_HIP(ctx, hipStreamSynchronize(ctx->stream));                      // lut_host may be rewritten by the next call
    ctx->weights_loaded = true;
    return RTDD_OK;
}

// The omega schedule depends only on the iteration index, so one device copy serves every call;
// it is re-uploaded only when a longer schedule is requested.
static int ensure_omegas(rtdd_ctx *ctx, int n) {
    if (n <= ctx->omega_cap) return RTDD_OK;
    int cap = n < 1024 ? 1024 : n;
    std::vector<float> om;
    omega_schedule(cap, om);
    RTDD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->omega_dev) { RTDD_HIP(ctx, hipFree(ctx->omega_dev)); ctx->omega_dev = nullptr; ctx->omega_cap = 0; }
    RTDD_HIP(ctx, hipMalloc((void **)&ctx->omega_dev, (size_t)cap * sizeof(float)));
    RTDD_HIP(ctx, hipMemcpy(ctx->omega_dev, om.data(), (size_t)cap * sizeof(float), hipMemcpyHostToDevice));
    ctx->omega_cap = cap;
    return RTDD_OK;
}

static constexpr int kAutoMaxCycles = 60;

static int check_solve_args(rtdd_ctx *ctx, const float *depth, size_t depthPitch, const uint8_t *scribble, size_t scribblePitch,
                            const uint8_t *gray, size_t grayPitch, int rows, int cols, int level) {
    REQUIRE(ctx, depth && scribble && gray, "null image pointer");
    REQUIRE(ctx, rows > 0 && cols > 0, "rows and cols must be positive");
    REQUIRE(ctx, depthPitch >= (size_t)cols * sizeof(float) && depthPitch % sizeof(float) == 0,
        "depth pitch too small or not a multiple of 4");
    REQUIRE(ctx, scribblePitch >= (size_t)cols && grayPitch >= (size_t)cols, "u8 pitch smaller than a row");
    if (ctx->levels.empty()) return fail(ctx, RTDD_ERR_STATE, "rtdd_allocate has not been called");
    if (!ctx->weights_loaded) return fail(ctx, RTDD_ERR_STATE, "rtdd_load_weights has not been called");
    REQUIRE(ctx, level >= 0 && level < (int)ctx->levels.size(), "level out of range");
    const Level &L = ctx->levels[level];
    REQUIRE(ctx, plane_elems(rows, cols) <= L.elems, "rows x cols exceeds the level's allocation");
    return RTDD_OK;
}

namespace {

// One rtdd_solve_ex call between k_prepare and k_finish: which planes hold the iterate, what has run, the last residual.
struct Solve {
    rtdd_ctx *ctx;
    const Level &L;
    size_t ip;
    int rows, cols;
    const rtdd_solve_params *p;
    int images;                                    // a batched solve: the images every launch covers (blockIdx.z)
    int done = 0, launches = 0, cycles = 0;
    int pk = 0, pm = 1;                            // planes holding x_k and x_{k-1}
    float residual = NAN;

    bool stop_on_residual() const { return p->tolerance > 0.0f; }
    bool reached() const { return residual <= p->tolerance; }
    int check() { return launch_residual(ctx, L, ip, pk, rows, cols, &residual); }

    // the reference's scheme, optionally in chunks with a residual check after each
    int chebyshev_jacobi() {
        std::vector<float> omegas;
        omega_schedule(p->maxIterations, omegas);
        const bool blocked = ctx->opt.sweep_kernel != 1;       // 0 (auto) and 2 -> temporally blocked kernel
        const float *omegas_dev = nullptr;
        int rc;
        if (blocked && p->maxIterations > 0) {
            if ((rc = ensure_omegas(ctx, p->maxIterations)) != RTDD_OK) return rc;
            omegas_dev = ctx->omega_dev;
        }
        const int chunk = stop_on_residual() ? (p->checkEvery > 0 ? p->checkEvery : 16) : p->maxIterations;
        while (done < p->maxIterations) {
            const int n = p->maxIterations - done < chunk ? p->maxIterations - done : chunk;
            int ln = 0;
            rc = blocked ? launch_sweeps_blocked(ctx, L, ip, rows, cols, omegas_dev + done, n, &pk, &pm, &ln, images)
                         : launch_sweeps(ctx, L, ip, rows, cols, omegas.data() + done, n, &pk, &pm, &ln);
            if (rc != RTDD_OK) return rc;
            done += n; launches += ln;
            if (stop_on_residual()) {
                if ((rc = check()) != RTDD_OK) return rc;
                if (reached()) break;
            }
        }
        return RTDD_OK;
    }

    // n red-black sweeps (capped by maxIterations) at one relaxation factor
    int red_black_sweeps(int n, float omega) {
        if (n > p->maxIterations - done) n = p->maxIterations - done;
        if (n <= 0) return RTDD_OK;
        int ln = 2 * n, rc;
        if (ctx->opt.sweep_kernel == 1) rc = launch_rbgs(ctx, L, ip, pk, rows, cols, n, omega);     // one launch per colour, in place
        else rc = launch_rbgs_blocked(ctx, L, ip, rows, cols, n, omega, &pk, &ln);                  // register-blocked, ping-pong planes
        done += n; launches += ln;
        return rc;
    }

    // Gauss-Seidel / SOR at a fixed factor, optionally in chunks with a residual check after each
    int red_black() {
        const int chunk = stop_on_residual() ? (p->checkEvery > 0 ? p->checkEvery : 16) : (p->maxIterations > 0 ? p->maxIterations : 1);
        const float omega = p->relaxation == 0.0f ? 1.0f : p->relaxation;
        int rc;
        while (done < p->maxIterations) {
            if ((rc = red_black_sweeps(chunk, omega)) != RTDD_OK) return rc;
            if (stop_on_residual()) {
                if ((rc = check()) != RTDD_OK) return rc;
                if (reached()) break;
            }
        }
        return RTDD_OK;
    }

    // RTDD_RELAXATION_AUTO: SOR cycles.  Over-relaxation removes the smooth error a plain sweep hardly touches, but in f32 it idles
    // at a residual ~ ulp(x)/(2 - omega); plain Gauss-Seidel has an exact f32 fixed point but is slow on smooth error.  So:
    // n_hi sweeps at omega_hi, n_hi/4 at omega_mid, then a Gauss-Seidel polish of at most 100 sweeps with the residual checked
    // every 20; a cycle that does not get there is followed by one twice as long and twice as close to omega = 2, until the
    // tolerance or maxIterations.  After V-cycles the smooth error is gone and half the length does
    // (scripts/auto_probe.py).
    int sor_cycles(bool after_vcycles) {
        const int longest = rows > cols ? rows : cols;
        const int base = after_vcycles ? (longest + 1) / 2 : longest;
        double w0 = 2.0 / (1.0 + sin(4.0 * 3.14159265358979323846 / (double)longest));
        if (w0 > 1.99) w0 = 1.99;
        if (w0 < 1.0) w0 = 1.0;
        bool ok = false;
        int rc;
        for (int cycle = 0; done < p->maxIterations && !ok; cycle++) {
            const int e = cycle < 6 ? cycle : 6;
            double gap = (2.0 - w0) / (double)(1 << e);
            if (gap < 0.005) gap = 0.005;
            const float w_hi = (float)(2.0 - gap);
            float w_mid = (float)(2.0 - 10.0 * gap);
            if (w_mid < 1.0f) w_mid = 1.0f;
            const int n_hi = base << e;
            if ((rc = red_black_sweeps(n_hi, w_hi)) != RTDD_OK) return rc;
            if ((rc = red_black_sweeps(n_hi / 4, w_mid)) != RTDD_OK) return rc;
            for (int k = 0; k < 5 && done < p->maxIterations && !ok; k++) {
                if ((rc = red_black_sweeps(20, 1.0f)) != RTDD_OK) return rc;
                if (stop_on_residual()) {
                    if ((rc = check()) != RTDD_OK) return rc;
                    ok = reached();
                }
            }
        }
        return RTDD_OK;
    }

    // V-cycles; alternative_seconds > 0: leave when the cycles still needed are modelled dearer than that (RTDD_METHOD_AUTO)
    int vcycles(int max_cycles, int check_every, double alternative_seconds) {
        const double px = (double)rows * cols;
        const double cycle_seconds = ctx->opt.auto_cycle_fixed_ns * 1e-9 + px * ctx->opt.auto_cycle_fs_per_px * 1e-15;
        return launch_multigrid(ctx, L, ip, rows, cols, max_cycles, p->tolerance, check_every, alternative_seconds, cycle_seconds, &pk,
            &cycles, &residual, &launches);
    }

    // V-cycles while they pay: they stop at the tolerance, after kAutoMaxCycles, or when the cycles still needed (at the rate of the
    // last two) are modelled to cost more than finishing with SOR cycles of half length -- thin high-contrast structures stall them
    // (DESIGN.md section 7), and below ~4K a cycle is launch-bound and dear.  Then those SOR cycles.
    int automatic() {
        const int longest = rows > cols ? rows : cols;
        const double px = (double)rows * cols;
        const double per_px = px * ctx->opt.auto_sweep_fs_per_px * 1e-15, floor_s = ctx->opt.auto_sweep_floor_ns * 1e-9;
        const double sweep_seconds = per_px > floor_s ? per_px : floor_s;                   // k_rbgs_blocked (constants: RTDD_OPT_AUTO_*)
        const double sor_seconds = ((double)((longest + 1) / 2) * 1.25 + 20.0) * sweep_seconds;
        const int rc = vcycles(kAutoMaxCycles, 1, sor_seconds);
        if (rc != RTDD_OK || reached()) return rc;
        return sor_cycles(true);
    }
};

}  // namespace

// one attempt: stage + edge weights, the sweeps, the (guarded) copy-back
static int solve_once(rtdd_ctx *ctx, float *depth, size_t depthPitch, const uint8_t *scribble, size_t scribblePitch,
                      const uint8_t *gray, size_t grayPitch, int rows, int cols, int level, const rtdd_solve_params *params, int seq,
                      const SolveTargets &t, SolveOutcome *out) {
    // (the first image the launches cover: image 0 of 1 unless the caller says otherwise)
    const Level L = ctx->levels[level].view(t.batch.first);
    const size_t ip = plane_pitch(cols);
    const bool prof = ctx->profile_on;
    hipEvent_t *ev = ctx->ev + 4 * (ctx->prof_pending % rtdd_ctx::kProfSlots);

    if (prof) RTDD_HIP(ctx, hipEventRecord(ev[0], ctx->stream));
    int rc = launch_prepare(ctx, L, ip, depth, depthPitch, scribble, scribblePitch, gray, grayPitch, rows, cols, level, t.batch);
    if (rc != RTDD_OK) return rc;
    if (prof) RTDD_HIP(ctx, hipEventRecord(ev[1], ctx->stream));

    ctx->last_info = rtdd_solve_info{};
    ctx->last_info.residual = NAN;
    Solve s{ctx, L, ip, rows, cols, params, t.batch.n};
    switch (params->method) {
        case RTDD_METHOD_CHEBYSHEV_JACOBI: rc = s.chebyshev_jacobi(); break;
        case RTDD_METHOD_MULTIGRID:
            rc = s.vcycles(params->maxIterations, params->checkEvery > 0 ? params->checkEvery : 1, 0.0);
            s.done = s.cycles;
            break;
        case RTDD_METHOD_AUTO: rc = s.automatic(); break;
        default: rc = params->relaxation < 0.0f ? s.sor_cycles(false) : s.red_black(); break;
    }
    if (rc != RTDD_OK) return rc;

    if (prof) RTDD_HIP(ctx, hipEventRecord(ev[2], ctx->stream));
    // (a deferred copy-back is k_pyrup_inject's: estimate_levels hands it the same number)
    if (out) { out->plane = s.pk; out->seq = seq; }
    if (!t.defer_finish) {
        rc = launch_finish(ctx, L, ip, s.pk, depth, depthPitch, rows, cols, t, seq);
        if (rc != RTDD_OK) return rc;
    }
    if (prof) {
        RTDD_HIP(ctx, hipEventRecord(ev[3], ctx->stream));
        const int slot = ctx->prof_pending % rtdd_ctx::kProfSlots;
        ctx->prof_launches[slot] = s.launches; ctx->prof_sweeps[slot] = s.done;
        ctx->prof_pending++;                        // resolved (and synchronised) by rtdd_profile_get, not here
    }
    ctx->last_info.iterations = s.done; ctx->last_info.residual = s.residual; ctx->last_info.cycles = s.cycles;
    ctx->last_info.fp_contract = ctx->opt.fp_contract; ctx->last_info.launches = s.launches;
    return RTDD_OK;
}

int rtdd_solve_ex(rtdd_ctx *ctx, float *depth, size_t depthPitch, const uint8_t *scribble, size_t scribblePitch,
                  const uint8_t *gray, size_t grayPitch, int rows, int cols, int level,
                  const rtdd_solve_params *params, rtdd_solve_info *info) {
    return solve_with(ctx, depth, depthPitch, scribble, scribblePitch, gray, grayPitch, rows, cols, level, params, info, SolveTargets(),
        nullptr);
}

}  // extern "C"
#pragma GCC visibility pop

// rtdd_solve_ex, plus what the library's own callers add to it (SolveTargets: a batch of images in the same launches, a deferred
// copy-back, the u8 copies of the result, whether the call is logged on its own).
int rtdd::solve_with(rtdd_ctx *ctx, float *depth, size_t depthPitch, const uint8_t *scribble, size_t scribblePitch,
                     const uint8_t *gray, size_t grayPitch, int rows, int cols, int level,
                     const rtdd_solve_params *params, rtdd_solve_info *info, const SolveTargets &t, SolveOutcome *out) {
    if (!ctx) return RTDD_ERR_INVALID;
    REQUIRE(ctx, params != nullptr, "null params");
    REQUIRE(ctx, params->maxIterations >= 0, "maxIterations must be >= 0");
    REQUIRE(ctx, params->method == RTDD_METHOD_CHEBYSHEV_JACOBI || params->method == RTDD_METHOD_RED_BLACK_GS
        || params->method == RTDD_METHOD_MULTIGRID ||
                 params->method == RTDD_METHOD_AUTO, "unknown method");
    REQUIRE(ctx, params->method != RTDD_METHOD_AUTO || params->tolerance > 0.0f, "RTDD_METHOD_AUTO needs a tolerance");
    REQUIRE(ctx, params->method != RTDD_METHOD_RED_BLACK_GS || params->relaxation == RTDD_RELAXATION_AUTO
        || (params->relaxation >= 0.0f && params->relaxation < 2.0f),
            "relaxation must be in [0,2) or RTDD_RELAXATION_AUTO");
    int rc = check_solve_args(ctx, depth, depthPitch, scribble, scribblePitch, gray, grayPitch, rows, cols, level);
    if (rc != RTDD_OK) return rc;
    REQUIRE(ctx, t.batch.first >= 0 && t.batch.n >= 1 && t.batch.first + t.batch.n <= ctx->levels_images,
            "the batch exceeds what the context's levels were allocated for");
    REQUIRE(ctx, t.batch.n == 1
        || (params->method == RTDD_METHOD_CHEBYSHEV_JACOBI && params->tolerance <= 0.0f && ctx->opt.sweep_kernel != 1),
            "a batched solve runs the reference's scheme with the temporally blocked kernel only");
    DeviceGuard g(ctx->device);
    // (once in 10^9 solves) the sequence numbers start over: nothing may be left that compares against them
    if (ctx->solve_seq >= (1 << 30)) {
        if (!ctx->healing) { const int rc_ = settle_pending(ctx); if (rc_ != RTDD_OK) return rc_; }
        RTDD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        *(volatile int *)ctx->confirm_host = 0;
        ctx->solve_seq = 0; ctx->publish_seq = 0;
    }
    const int seq = ++ctx->solve_seq;
    const Options asked = ctx->opt;
    rc = solve_once(ctx, depth, depthPitch, scribble, scribblePitch, gray, grayPitch, rows, cols, level, params, seq, t, out);
    // A residual check inside the solve found the status word set, and the calls before this one have been healed
    // (check_persistent_status):
    // nothing of this solve has reached the caller's buffers (its copy-back is the last thing it does), so it simply starts over --
    // persistence is off by now.
    if (rc == kRestartSolve) rc =
        solve_once(ctx, depth, depthPitch, scribble, scribblePitch, gray, grayPitch, rows, cols, level, params, seq, t, out);
    if (rc == kRestartSolve) rc =
        fail(ctx, RTDD_ERR_TIMEOUT, "the solve was restarted after a timed-out persistent launch and failed again");
    if (rc != RTDD_OK) return rc;
    // re-armed (check_persistent_status)
    if (!ctx->healing && ctx->persist_suspend > 0 && --ctx->persist_suspend == 0 && ctx->persistent_wanted) ctx->opt.persistent = 1;
    // remembered until a copy-back kernel or a synchronising call has confirmed it
    if (!ctx->healing && t.logged && ctx->opt.timeout_heal) {
        prune_confirmed(ctx);
        if (ctx->pending.size() >= kMaxPendingOps) { ctx->pending.clear(); ctx->pending_overflow = true; }
        PendingOp op;
        op.kind = PendingOp::kSolve; op.opt = asked; op.seq = seq;
        op.depth = depth; op.depthPitch = depthPitch; op.scribble = scribble; op.scribblePitch = scribblePitch; op.gray = gray;
        op.grayPitch = grayPitch;
        op.rows = rows; op.cols = cols; op.level = level; op.params = *params;
        op.targets = t;
        op.id = ++ctx->op_counter;
        ctx->pending.push_back(op);
    }
    if (info) *info = ctx->last_info;
    return RTDD_OK;
}

#pragma GCC visibility push(default)
extern "C" {

int rtdd_last_solve_info(rtdd_ctx *ctx, rtdd_solve_info *info) {
    if (!ctx || !info) return RTDD_ERR_INVALID;
    *info = ctx->last_info;
    return RTDD_OK;
}

int rtdd_multigrid_level(rtdd_ctx *ctx, int level, int which, float *host, int *rows, int *cols) {
    if (!ctx) return RTDD_ERR_INVALID;
    REQUIRE(ctx, rows && cols, "null size pointers");
    DeviceGuard g(ctx->device);
    const int rc = mg_download(ctx, level, which, host, rows, cols);
    if (rc == RTDD_ERR_INVALID) return fail(ctx, rc, "no multigrid hierarchy, or level/plane out of range");
    return rc;
}

int rtdd_matrix_free_solver(rtdd_ctx *ctx, float *depth, size_t depthPitch, const uint8_t *scribble, size_t scribblePitch,
                            const uint8_t *gray, size_t grayPitch, int rows, int cols, float beta, int maxIterations,
                            float tolerance, int level) {
    (void)beta; (void)tolerance;                   // ignored by the reference too (src/GPUSolver.cu:274-275)
    if (!ctx) return RTDD_ERR_INVALID;
    if (maxIterations < 0) maxIterations = 0;      // the reference's loop simply does not run (:295)
    rtdd_solve_params p;
    p.method = RTDD_METHOD_CHEBYSHEV_JACOBI; p.maxIterations = maxIterations; p.tolerance = 0.0f; p.checkEvery = 0; p.relaxation = 0.0f;
    return rtdd_solve_ex(ctx, depth, depthPitch, scribble, scribblePitch, gray, grayPitch, rows, cols, level, &p, nullptr);
}

int rtdd_index_to_weight(rtdd_ctx *ctx, const uint8_t *gray, size_t grayPitch, const float *depth, size_t depthPitch,
                         int32_t *index2, int level, int rows, int cols) {
    if (!ctx) return RTDD_ERR_INVALID;
    REQUIRE(ctx, gray && depth && index2, "null pointer");
    REQUIRE(ctx, rows > 0 && cols > 0 && grayPitch >= (size_t)cols && depthPitch >= (size_t)cols * 4, "bad size or pitch");
    if (ctx->maxLevel < 0) return fail(ctx, RTDD_ERR_STATE, "rtdd_allocate has not been called (maxLevel unknown)");
    DeviceGuard g(ctx->device);
    // reads a depth image a logged, unconfirmed solve may not have written (its copy-back stores nothing after a time-out) and is not
    // logged itself: confirm or heal first.  Nothing logged: nothing to wait for.
    { const int rc_ = settle_pending(ctx); if (rc_ != RTDD_OK) return rc_; }
    return launch_index_to_weight(ctx, gray, grayPitch, depth, depthPitch, index2, level, rows, cols);
}

// ---- image processing ---------------------------------------------------------------------------

int rtdd_convert_to_float(rtdd_ctx *ctx, const uint8_t *src, size_t srcPitch, float *dst, size_t dstPitch,
                          const uint8_t *mask, size_t maskPitch, int rows, int cols) {
    if (!ctx) return RTDD_ERR_INVALID;
    REQUIRE(ctx, src && dst && mask, "null image pointer");
    REQUIRE(ctx, rows >= 0 && cols >= 0, "negative size");
    if (rows == 0 || cols == 0) return RTDD_OK;
    REQUIRE(ctx, srcPitch >= (size_t)cols * 3 && dstPitch >= (size_t)cols * 4 && maskPitch >= (size_t)cols, "pitch smaller than a row");
    DeviceGuard g(ctx->device);
    // (the coarsest depth image of the context's pyramid? then the next estimate injects again; stale annotation pointers: RTDD_ERR_STATE)
    { const int rc_ = pyramid_check_read(ctx, src, mask); if (rc_ != RTDD_OK) return rc_; }
    { const int rc_ = pyramid_note_write(ctx, dst, dst); if (rc_ != RTDD_OK) return rc_; }
    return launch_convert(ctx, src, srcPitch, dst, dstPitch, mask, maskPitch, rows, cols);
}

int rtdd_pyrdown_annotation(rtdd_ctx *ctx, const uint8_t *prevScribble, size_t prevScribblePitch, const uint8_t *prevEdited,
                            size_t prevEditedPitch, int previousRows, int previousCols, uint8_t *currScribble,
                            size_t currScribblePitch, uint8_t *currEdited, size_t currEditedPitch, int currentRows, int currentCols) {
    if (!ctx) return RTDD_ERR_INVALID;
    REQUIRE(ctx, prevScribble && prevEdited && currScribble && currEdited, "null image pointer");
    REQUIRE(ctx, previousRows >= 0 && previousCols >= 0 && currentRows >= 0 && currentCols >= 0, "negative size");
    if (currentRows == 0 || currentCols == 0) return RTDD_OK;
    REQUIRE(ctx, prevScribblePitch >= (size_t)previousCols && prevEditedPitch >= (size_t)previousCols * 3 &&
                 currScribblePitch >= (size_t)currentCols && currEditedPitch >= (size_t)currentCols * 3, "pitch smaller than a row");
    DeviceGuard g(ctx->device);
    { const int rc_ = pyramid_check_read(ctx, prevScribble, prevEdited); if (rc_ != RTDD_OK) return rc_; }
    { const int rc_ = pyramid_note_write(ctx, currScribble, currEdited); if (rc_ != RTDD_OK) return rc_; }
    return launch_pyrdown_annotation(ctx, prevScribble, prevScribblePitch, prevEdited, prevEditedPitch, previousRows, previousCols,
                                     currScribble, currScribblePitch, currEdited, currEditedPitch, currentRows, currentCols);
}

int rtdd_paint_image(rtdd_ctx *ctx, int x, int y, int scribbleColor, int scribbleRadius, uint8_t *edited, size_t editedPitch,
                     uint8_t *scribble, size_t scribblePitch, int rows, int cols) {
    if (!ctx) return RTDD_ERR_INVALID;
    REQUIRE(ctx, edited && scribble, "null image pointer");
    REQUIRE(ctx, rows >= 0 && cols >= 0, "negative size");
    if (rows == 0 || cols == 0) return RTDD_OK;
    REQUIRE(ctx, editedPitch >= (size_t)cols * 3 && scribblePitch >= (size_t)cols, "pitch smaller than a row");
    DeviceGuard g(ctx->device);
    { const int rc_ = pyramid_note_write(ctx, scribble, edited); if (rc_ != RTDD_OK) return rc_; }
    return launch_paint(ctx, x, y, scribbleColor, scribbleRadius, edited, editedPitch, scribble, scribblePitch, rows, cols);
}

// ---- depth effects -------------------------------------------------------------------------------

static int check_effect(rtdd_ctx *ctx, const void *a, const void *b, const void *c, size_t op, size_t dp, size_t ap, int rows, int cols) {
    REQUIRE(ctx, a && b && c, "null image pointer");
    REQUIRE(ctx, rows >= 0 && cols >= 0, "negative size");
    REQUIRE(ctx, (long long)rows * rows + (long long)cols * cols < 2147483647LL, "image too large");
    REQUIRE(ctx, op >= (size_t)cols * 3 && ap >= (size_t)cols * 3 && dp >= (size_t)cols * 4, "pitch smaller than a row");
    return RTDD_OK;
}

int rtdd_simulate_defocus(rtdd_ctx *ctx, const uint8_t *original, size_t originalPitch, const float *depth, size_t depthPitch,
                          uint8_t *artistic, size_t artisticPitch, int rows, int cols) {
    if (!ctx) return RTDD_ERR_INVALID;
    int rc = check_effect(ctx, original, depth, artistic, originalPitch, depthPitch, artisticPitch, rows, cols);
    if (rc != RTDD_OK || rows == 0 || cols == 0) return rc;
    REQUIRE(ctx, original != artistic, "defocus cannot run in place");
    DeviceGuard g(ctx->device);
    rc = launch_defocus(ctx, original, originalPitch, depth, depthPitch, artistic, artisticPitch, rows, cols);
    if (rc == RTDD_OK) log_effect(ctx, PendingOp::kDefocus, original, originalPitch, nullptr, 0, depth, depthPitch, artistic, artisticPitch,
        rows, cols);
    return rc;
}

int rtdd_simulate_desaturation(rtdd_ctx *ctx, const uint8_t *original, size_t originalPitch, const uint8_t *gray, size_t grayPitch,
                               const float *depth, size_t depthPitch, uint8_t *artistic, size_t artisticPitch, int rows, int cols) {
    if (!ctx) return RTDD_ERR_INVALID;
    int rc = check_effect(ctx, original, depth, artistic, originalPitch, depthPitch, artisticPitch, rows, cols);
    if (rc != RTDD_OK || rows == 0 || cols == 0) return rc;
    REQUIRE(ctx, gray && grayPitch >= (size_t)cols, "bad gray image");
    DeviceGuard g(ctx->device);
    rc = launch_desaturate(ctx, original, originalPitch, gray, grayPitch, depth, depthPitch, artistic, artisticPitch, rows, cols);
    if (rc == RTDD_OK) log_effect(ctx, PendingOp::kDesaturate, original, originalPitch, gray, grayPitch, depth, depthPitch, artistic,
        artisticPitch, rows, cols);
    return rc;
}

int rtdd_simulate_haze(rtdd_ctx *ctx, const uint8_t *original, size_t originalPitch, const float *depth, size_t depthPitch,
                       uint8_t *artistic, size_t artisticPitch, int rows, int cols) {
    if (!ctx) return RTDD_ERR_INVALID;
    int rc = check_effect(ctx, original, depth, artistic, originalPitch, depthPitch, artisticPitch, rows, cols);
    if (rc != RTDD_OK || rows == 0 || cols == 0) return rc;
    DeviceGuard g(ctx->device);
    rc = launch_haze(ctx, original, originalPitch, depth, depthPitch, artistic, artisticPitch, rows, cols);
    if (rc == RTDD_OK) log_effect(ctx, PendingOp::kHaze, original, originalPitch, nullptr, 0, depth, depthPitch, artistic, artisticPitch,
        rows, cols);
    return rc;
}

}  // extern "C"
#pragma GCC visibility pop
