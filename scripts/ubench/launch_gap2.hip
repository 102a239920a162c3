// Kernel that mimics the blocked sweep's memory behaviour only: each 1024-thread workgroup loads a 128x96 tile of 3 planes
// (12 dwordx4 per thread... here 9), spins for `spin_us`, stores 2 planes of the 112x80 centre.  What is the launch interval?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ unsigned long long g_st[1024][2];
__global__ __launch_bounds__(1024) void k(const float4 *a, const float4 *b, const float4 *c, float4 *ya, float4 *yb, int ip4, int spin_ticks, int lds_kb) {
    extern __shared__ float4 sm[];
    if (threadIdx.x == 0) g_st[blockIdx.x][0] = __builtin_amdgcn_s_memrealtime();
    const int tx = blockIdx.x % 18, ty = blockIdx.x / 18;
    const int lx = threadIdx.x & 31, tr = threadIdx.x >> 5;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int g = 0; g < 3; g++) {
        int y = ty * 80 - 8 + tr * 3 + g, x4 = tx * 28 - 2 + lx;
        y = y < 0 ? 0 : (y > 1079 ? 1079 : y); x4 = x4 < 0 ? 0 : (x4 > 479 ? 479 : x4);
        const size_t o = (size_t)y * ip4 + x4;
        const float4 v1 = a[o], v2 = b[o], v3 = c[o];
        acc.x += v1.x + v2.x + v3.x; acc.y += v1.y + v2.y + v3.y; acc.z += v1.z + v2.z + v3.z; acc.w += v1.w + v2.w + v3.w;
    }
    if (lds_kb) sm[threadIdx.x] = acc;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < spin_ticks) __builtin_amdgcn_s_sleep(8);
    for (int g = 0; g < 3; g++) {
        const int ry = tr * 3 + g, y = ty * 80 - 8 + ry, x4 = tx * 28 - 2 + lx;
        if (ry >= 8 && ry < 88 && lx >= 2 && lx < 30 && y < 1080 && x4 < 480 && y >= 0) { ya[(size_t)y * ip4 + x4] = acc; yb[(size_t)y * ip4 + x4] = acc; }
    }
    __builtin_amdgcn_s_waitcnt(0);
    if (threadIdx.x == 0) g_st[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
}
int main(int argc, char **argv) {
    const int spin = argc > 1 ? atoi(argv[1]) : 1000;   // 100 MHz ticks: 1000 = 10 us
    const int lds_kb = argc > 2 ? atoi(argv[2]) : 0;
    const int ip4 = 496; const size_t n = (size_t)ip4 * 1100;
    float4 *p[5]; for (auto &q : p) { hipMalloc(&q, n * 16); hipMemset(q, 0, n * 16); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        for (int i = 0; i < 100; i++) { k<<<252, 1024, lds_kb * 1024>>>(p[i & 1 ? 3 : 0], p[i & 1 ? 4 : 1], p[2], p[i & 1 ? 0 : 3], p[i & 1 ? 1 : 4], ip4, spin, lds_kb); }
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long st[1024][2]; hipMemcpyFromSymbol(st, HIP_SYMBOL(g_st), sizeof(st));
    unsigned long long a = ~0ull, b = 0; for (int i = 0; i < 252; i++) { if (st[i][0] < a) a = st[i][0]; if (st[i][1] > b) b = st[i][1]; }
    printf("spin %.1f us, dyn LDS %d KB: launch interval %.2f us, in-kernel span %.2f us\n", spin / 100.0, lds_kb, ms * 1e3 / 100, (b - a) / 100.0);
    return 0;
}
