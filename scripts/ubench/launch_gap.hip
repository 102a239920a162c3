// What does a dependent kernel boundary cost on gfx950 as a function of workgroup size, LDS per workgroup and bytes left dirty?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS_BYTES>
__global__ void k(float *p, int nstore) {
    __shared__ float s[LDS_BYTES / 4 > 0 ? LDS_BYTES / 4 : 1];
    if (LDS_BYTES > 0) { s[threadIdx.x] = threadIdx.x; __syncthreads(); }
    float v = LDS_BYTES > 0 ? s[(threadIdx.x + 1) % blockDim.x] : 1.0f;
    for (int i = 0; i < nstore; i++) ((float4 *)p)[((size_t)blockIdx.x * nstore + i) * blockDim.x + threadIdx.x] = make_float4(v, v, v, v);
}
template <int L> void run(const char *name, int blocks, int threads, int nstore, float *buf) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; i++) k<L><<<blocks, threads>>>(buf, nstore);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 200; i++) k<L><<<blocks, threads>>>(buf, nstore);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s blocks %4d threads %4d  stores %6.1f MB/launch : %.2f us per launch\n", name, blocks, threads, (double)blocks * threads * nstore * 16 / 1e6, ms * 1e3 / 200);
}
int main() {
    float *buf; hipMalloc(&buf, 256u << 20);
    run<0>("no LDS", 252, 1024, 0, buf);
    run<0>("no LDS", 252, 256, 0, buf);
    run<66 * 1024>("66 KB LDS", 252, 1024, 0, buf);
    run<16 * 1024>("16 KB LDS", 252, 1024, 0, buf);
    run<0>("no LDS, dirty stores", 252, 1024, 4, buf);     // 16.5 MB
    run<66 * 1024>("66 KB LDS, dirty stores", 252, 1024, 4, buf);
    run<0>("no LDS, dirty stores", 1008, 256, 4, buf);
    run<0>("no LDS, dirty stores 66 MB", 252, 1024, 16, buf);
    return 0;
}
