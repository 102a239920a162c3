// Diagnostic: per-wave, per-sweep timeline of ONE workgroup of the persistent 1080p launch (s_memtime stamps, shader cycles).
// Shows who waits for whom inside a block of sweeps.  Timing only: planes hold random data.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off ... -DRTDD_TIMELINE scripts/ubench/sweep_timeline.hip
#define RTDD_TIMELINE 1
#include "../../realtimedepthdiffusion_amd/csrc/sweep_blocked.hip"
#include <algorithm>
#include <cstdlib>
namespace rtdd {
int fail(rtdd_ctx *, int s, const char *, hipError_t) { return s; }
int prepare_persistent_launch(rtdd_ctx *ctx, int nblocks, int *flag_base) {
    if (!ctx->sync_words) { (void)hipMalloc((void **)&ctx->sync_words, kSyncWords * sizeof(int)); (void)hipMemset(ctx->sync_words, 0, kSyncWords * sizeof(int)); }
    if (!ctx->flag_epoch) ctx->flag_epoch = 1;
    *flag_base = ctx->flag_epoch; ctx->flag_epoch += nblocks + 1;
    return 0;
}
}
using namespace rtdd;
int main(int argc, char **argv) {
    int rows = argc > 1 ? atoi(argv[1]) : 1080, cols = argc > 2 ? atoi(argv[2]) : 1920, tile = argc > 3 ? atoi(argv[3]) : 4, T = argc > 4 ? atoi(argv[4]) : 8;
    int nsweeps = argc > 5 ? atoi(argv[5]) : 32;
    rtdd_ctx ctx; ctx.opt.tile = tile; ctx.opt.temporal_depth = T; ctx.opt.persistent = argc > 6 ? atoi(argv[6]) : 1; ctx.num_cus = 256;
    { int t = argc > 7 ? atoi(argv[7]) : 100; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tl_tile), &t, sizeof(t)); }
    { int fb; rtdd::prepare_persistent_launch(&ctx, 0, &fb); for (auto &t : ctx.persist_fit) t[0] = t[1] = -1; }
    Level L; size_t ip = plane_pitch(cols); L.elems = plane_elems(rows, cols);
    std::vector<float> h(L.elems); for (auto &v : h) v = (float)(rand() % 25500) / 100.0f;
    std::vector<uint32_t> hm(L.elems); for (auto &v : hm) v = (rand() % 12) | ((rand() % 12) << 8) | ((rand() % 10 == 0) ? kMetaDirichlet : 0);
    for (auto &p : L.plane) { (void)hipMalloc((void **)&p, L.elems * 4); (void)hipMemcpy(p, h.data(), L.elems * 4, hipMemcpyHostToDevice); }
    (void)hipMalloc((void **)&L.meta, L.elems * 4); (void)hipMemcpy(L.meta, hm.data(), L.elems * 4, hipMemcpyHostToDevice);
    float lut[257]; for (int i = 0; i < 256; i++) lut[i] = expf(-0.4f * i); lut[256] = 0;
    (void)hipMalloc((void **)&ctx.lut_dev, sizeof(lut)); (void)hipMemcpy(ctx.lut_dev, lut, sizeof(lut), hipMemcpyHostToDevice);
    std::vector<float> om(1024, 1.75f); float *om_d; (void)hipMalloc((void **)&om_d, 4096); (void)hipMemcpy(om_d, om.data(), 4096, hipMemcpyHostToDevice);
    int pk = 0, pm = 1, ln = 0;
    for (int rep = 0; rep < 3; rep++) launch_sweeps_blocked(&ctx, L, ip, rows, cols, om_d, nsweeps, &pk, &pm, &ln);
    (void)hipDeviceSynchronize();
    static unsigned long long tl[16][64][4];
    (void)hipMemcpyFromSymbol(tl, HIP_SYMBOL(g_tl), sizeof(tl));
    unsigned long long t0 = ~0ull;
    for (int w = 0; w < 16; w++) if (tl[w][0][0]) t0 = std::min(t0, tl[w][0][0]);
    printf("%dx%d tile %d T %d, %d sweeps; per wave and sweep: top-of-sweep / rows-in-hand / published / end, in shader cycles from the first stamp\n", cols, rows, tile, T, nsweeps);
    for (int s = 0; s < std::min(nsweeps, 30); s++) {
        printf("sweep %2d:", s);
        for (int w = 0; w < 16; w++) if (tl[w][s][0]) printf(" w%-2d %6llu+%4llu+%4llu+%4llu |", w, tl[w][s][0] - t0, tl[w][s][1] - tl[w][s][0], tl[w][s][2] - tl[w][s][1], tl[w][s][3] - tl[w][s][2]);
        printf("\n");
    }
    return 0;
}
