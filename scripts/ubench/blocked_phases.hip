// Diagnostic: where does a k_sweep_blocked launch spend its time?  Per-workgroup s_memrealtime stamps
// (100 MHz) at kernel start / after the tile load + weight setup / after the sweeps / after the stores.
// Timing only: planes hold random data.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -DRTDD_STAMPS ...
#define RTDD_STAMPS 1
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
    rtdd_ctx ctx; ctx.opt.tile = tile; ctx.opt.temporal_depth = T;
    { int fb; rtdd::prepare_persistent_launch(&ctx, 0, &fb); for (auto &t : ctx.persist_fit) t[0] = t[1] = -1; }
    if (getenv("RTDD_STREAM")) { hipStreamCreateWithFlags(&ctx.stream, hipStreamNonBlocking); printf("using a created stream\n"); }
    Level L; size_t ip = plane_pitch(cols); L.elems = plane_elems(rows, cols);
    std::vector<float> h(L.elems); for (auto &v : h) v = (float)(rand() % 25500) / 100.0f;
    std::vector<uint32_t> hm(L.elems); for (auto &v : hm) v = (rand() % 12) | ((rand() % 12) << 8) | ((rand() % 10 == 0) ? kMetaDirichlet : 0);
    for (auto &p : L.plane) { hipMalloc((void **)&p, L.elems * 4); hipMemcpy(p, h.data(), L.elems * 4, hipMemcpyHostToDevice); }
    hipMalloc((void **)&L.meta, L.elems * 4); hipMemcpy(L.meta, hm.data(), L.elems * 4, hipMemcpyHostToDevice);
    float lut[257]; for (int i = 0; i < 256; i++) lut[i] = expf(-0.4f * i); lut[256] = 0;
    hipMalloc((void **)&ctx.lut_dev, sizeof(lut)); hipMemcpy(ctx.lut_dev, lut, sizeof(lut), hipMemcpyHostToDevice);
    std::vector<float> om(1024, 1.75f); float *om_d; hipMalloc((void **)&om_d, 4096); hipMemcpy(om_d, om.data(), 4096, hipMemcpyHostToDevice);
    int pk = 0, pm = 1, ln = 0;
    for (int rep = 0; rep < 5; rep++) launch_sweeps_blocked(&ctx, L, ip, rows, cols, om_d, T * 20, &pk, &pm, &ln);
    hipDeviceSynchronize();
    // stamps of the LAST launch of a back-to-back train (every launch overwrites them): steady state, not an isolated launch
    { static unsigned long long z[4096][5]; hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, ctx.stream);
    launch_sweeps_blocked(&ctx, L, ip, rows, cols, om_d, T * 20, &pk, &pm, &ln);
    hipEventRecord(e1, ctx.stream);
    hipDeviceSynchronize();
    float evms = 0; hipEventElapsedTime(&evms, e0, e1);
    printf("launch interval (events / launches): %.2f us over %d launches\n", evms * 1e3 / ln, ln);
    { static unsigned long long z[4096][5]; hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)); }
    launch_sweeps_blocked(&ctx, L, ip, rows, cols, om_d, T, &pk, &pm, &ln);      // ONE stamped launch (stamps are max-accumulated)
    hipDeviceSynchronize();
    static unsigned long long st[4096][5];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st));
    // count WGs: find grid size as in launch
    unsigned long long t0 = ~0ull, tend = 0; int n = 0;
    for (int i = 0; i < 4096; i++) if (st[i][3] > st[i][0] && st[i][0] != 0) { n++; t0 = std::min(t0, st[i][0]); tend = std::max(tend, st[i][3]); }
    double a = 0, b = 0, c = 0, start = 0, ld = 0;
    for (int i = 0; i < 4096; i++) if (st[i][3] > st[i][0] && st[i][0] != 0) { ld += st[i][4] - st[i][0]; a += st[i][1] - st[i][0]; b += st[i][2] - st[i][1]; c += st[i][3] - st[i][2]; start += st[i][0] - t0; }
    if (getenv("RTDD_PERSIST")) {
        static unsigned long long z6[4096][6]; hipMemcpyToSymbol(HIP_SYMBOL(g_xphase), z6, sizeof(z6));
        ctx.opt.persistent = 1; ctx.num_cus = 256;
        hipEvent_t p0, p1; hipEventCreate(&p0); hipEventCreate(&p1);
        const int nb = 25;
        hipEventRecord(p0, ctx.stream);
        launch_sweeps_blocked(&ctx, L, ip, rows, cols, om_d, T * nb, &pk, &pm, &ln);
        hipEventRecord(p1, ctx.stream); hipDeviceSynchronize();
        float pms; hipEventElapsedTime(&pms, p0, p1);
        static unsigned long long xp[4096][6]; hipMemcpyFromSymbol(xp, HIP_SYMBOL(g_xphase), sizeof(xp));
        double ph[5] = {0, 0, 0, 0, 0}; int nw = 0;
        for (int i = 0; i < 4096; i++) if (xp[i][0]) { nw++; for (int k = 0; k < 5; k++) ph[k] += xp[i][k]; }
        printf("PERSISTENT %d launches, %d blocks of %d sweeps: %.2f us per block; wave-0 means per block: sweeps %.2f, publish+drain+barrier %.2f, flag+poll %.2f, acquire+barrier %.2f, halo load %.2f us\n",
               ln, nb, T, pms * 1e3 / nb, ph[0] / nw / 100 / (nb - 1), ph[1] / nw / 100 / (nb - 1), ph[2] / nw / 100 / (nb - 1), ph[3] / nw / 100 / (nb - 1), ph[4] / nw / 100 / (nb - 1));
    }
    printf("%dx%d tile %d T %d: %d workgroups; mean per WG: start skew %.2f us, load+setup %.2f us (of it: until the slowest wave's loads have landed %.2f, setup behind them %.2f), %d sweeps %.2f us (%.3f us/sweep), store %.2f us; first start -> last end %.2f us\n",
           cols, rows, tile, T, n, start / n / 100, a / n / 100, ld / n / 100, (a - ld) / n / 100, T, b / n / 100, b / n / 100 / T, c / n / 100, (tend - t0) / 100.0);
    return 0;
}
