// micro-benchmark: issue cost of the VALU ops the sweep kernel is made of (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define N_ITER 2000
template <int OP>
__global__ void k(float *out, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    v2f p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, pa = {a, a}, pb = {b, b};
    for (int i = 0; i < N_ITER; i++) {
        if (OP == 0) {  // 8 independent v_fma_f32
            x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
            x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
        } else if (OP == 1) {  // 4 independent v_pk_fma_f32 (= 8 fma)
            p0 = __builtin_elementwise_fma(p0, pa, pb); p1 = __builtin_elementwise_fma(p1, pa, pb);
            p2 = __builtin_elementwise_fma(p2, pa, pb); p3 = __builtin_elementwise_fma(p3, pa, pb);
        } else if (OP == 2) {  // 8 IEEE divisions
            x0 = b / x0; x1 = b / x1; x2 = b / x2; x3 = b / x3; x4 = b / x4; x5 = b / x5; x6 = b / x6; x7 = b / x7;
        } else if (OP == 3) {  // 8 v_rcp_f32
            x0 = __builtin_amdgcn_rcpf(x0); x1 = __builtin_amdgcn_rcpf(x1); x2 = __builtin_amdgcn_rcpf(x2); x3 = __builtin_amdgcn_rcpf(x3);
            x4 = __builtin_amdgcn_rcpf(x4); x5 = __builtin_amdgcn_rcpf(x5); x6 = __builtin_amdgcn_rcpf(x6); x7 = __builtin_amdgcn_rcpf(x7);
        } else if (OP == 4) {  // 8 v_div_fixup_f32
            x0 = __builtin_amdgcn_div_fixupf(x0, a, b); x1 = __builtin_amdgcn_div_fixupf(x1, a, b); x2 = __builtin_amdgcn_div_fixupf(x2, a, b); x3 = __builtin_amdgcn_div_fixupf(x3, a, b);
            x4 = __builtin_amdgcn_div_fixupf(x4, a, b); x5 = __builtin_amdgcn_div_fixupf(x5, a, b); x6 = __builtin_amdgcn_div_fixupf(x6, a, b); x7 = __builtin_amdgcn_div_fixupf(x7, a, b);
        } else if (OP == 5) {  // 8 v_med3
            x0 = __builtin_amdgcn_fmed3f(x0, a, b); x1 = __builtin_amdgcn_fmed3f(x1, a, b); x2 = __builtin_amdgcn_fmed3f(x2, a, b); x3 = __builtin_amdgcn_fmed3f(x3, a, b);
            x4 = __builtin_amdgcn_fmed3f(x4, a, b); x5 = __builtin_amdgcn_fmed3f(x5, a, b); x6 = __builtin_amdgcn_fmed3f(x6, a, b); x7 = __builtin_amdgcn_fmed3f(x7, a, b);
            asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        }
        if (OP == 1) asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int OP>
void run(const char *name, int ops_per_iter, float *out) {
    for (int wpc : {4, 8, 16, 32}) {   // waves per CU (256 CUs)
        int threads = 256, blocks = 256 * wpc / 4;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<OP><<<blocks, threads>>>(out, 1.0001f, 0.5f);
        hipEventRecord(e0);
        k<OP><<<blocks, threads>>>(out, 1.0001f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double wave_instr = (double)blocks * 4 * N_ITER * ops_per_iter;
        // cycles per wave-instruction per SIMD at 2.4 GHz nominal
        double cyc = ms * 1e-3 * 2.4e9 * 1024 / wave_instr;
        printf("%-14s waves/CU %2d  %.3f ms  %.2f cyc/wave-instr/SIMD (@2.4GHz)\n", name, wpc, ms, cyc);
    }
}
int main() {
    float *out; hipMalloc(&out, 256 * 32 * 64 * 4 * 4);
    run<0>("v_fma_f32", 8, out);
    run<1>("v_pk_fma_f32", 4, out);
    run<2>("fdiv(ieee)", 8, out);
    run<3>("v_rcp_f32", 8, out);
    run<4>("v_div_fixup", 8, out);
    run<5>("v_med3", 8, out);
    return 0;
}
