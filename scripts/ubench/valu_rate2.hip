// Does a 3-VGPR-operand v_fma_f32 issue at the same rate as one with scalar operands?  (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 2000
template <int OP>
__global__ void k(float *out, const float *in) {
    float x[8], y[8], z[8];
    for (int i = 0; i < 8; i++) { x[i] = in[threadIdx.x + i * 64]; y[i] = in[threadIdx.x + 512 + i * 64]; z[i] = in[threadIdx.x + 1024 + i * 64]; }
    float a = in[2000], b = in[2001];
    for (int it = 0; it < N_ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) x[i] = __builtin_fmaf(x[i], a, b);                    // 1 VGPR + 2 SGPR
            if (OP == 1) x[i] = __builtin_fmaf(x[i], y[i], z[i]);              // 3 VGPR (v_fma_f32)
            if (OP == 2) x[i] = __builtin_fmaf(y[i], z[i], x[i]);              // v_fmac_f32: dst accumulates
            if (OP == 3) x[i] = __builtin_fmaf(-y[(i + 1) & 7], z[(i + 3) & 7], x[i]);   // mixed registers
            if (OP == 4) x[i] = x[i] * y[i];                                   // v_mul 2 VGPR
            if (OP == 5) x[i] = fminf(fmaxf(x[i], a), b);                      // max+min
            if (OP == 6) x[i] = x[i] - y[i];
        }
        asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
    }
    float s = 0; for (int i = 0; i < 8; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char *name, int per, float *out, float *in) {
    for (int wpc : {4, 8, 16}) {
        int blocks = 256 * wpc / 4;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<OP><<<blocks, 256>>>(out, in); hipEventRecord(e0); k<OP><<<blocks, 256>>>(out, in); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-22s waves/SIMD %d  %.3f ms  %.2f nominal cyc/wave-instr/SIMD\n", name, wpc / 4, ms, ms * 1e-3 * 2.4e9 * 1024 / ((double)blocks * 4 * N_ITER * 8 * per));
    }
}
int main() {
    float *out, *in; hipMalloc(&out, 1 << 22); hipMalloc(&in, 1 << 16); hipMemset(in, 0x3f, 1 << 16);
    run<0>("fma 1vgpr+2sgpr", 1, out, in); run<1>("fma 3 vgpr", 1, out, in); run<2>("fmac (acc)", 1, out, in); run<3>("fma mixed regs", 1, out, in);
    run<4>("mul 2 vgpr", 1, out, in); run<5>("max+min", 2, out, in); run<6>("sub 2 vgpr", 1, out, in);
    return 0;
}
