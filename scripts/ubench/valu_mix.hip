// micro-benchmark: issue cost per SIMD of the instruction FORMS the sweep kernels are made of (gfx950), 4 and 8 waves per SIMD.
// Each variant is a loop of 8 independent instructions of one form (inline asm, so the form is exactly what is written).
// Build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/valu_mix.hip -o scripts/ubench/valu_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 4000
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP>
__global__ void k(float *out, float a, float b, unsigned long long mask) {
    float x[8], y[8];
    for (int i = 0; i < 8; i++) { x[i] = threadIdx.x + i; y[i] = threadIdx.x * 0.5f + i; }
    for (int it = 0; it < N_ITER; it++) {
#define A_FMA(i)   asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(y[i]), "v"(y[(i + 1) & 7]));
#define A_FMAC(i)  asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(y[(i + 1) & 7]));
#define A_MED3(i)  asm volatile("v_med3_f32 %0, %0, 0, %1" : "+v"(x[i]) : "s"(b));
#define A_CND(i)   asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "s"(mask));
#define A_DPP(i)   asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(x[i]) : "v"(y[i]));
#define A_FDPP(i)  asm volatile("v_fmac_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x[i]) : "v"(y[i]), "v"(y[(i + 1) & 7]));
#define A_NFDPP(i) asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x[i]) : "v"(y[i]), "v"(y[(i + 1) & 7]));
#define A_SUB(i)   asm volatile("v_sub_f32_e32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
#define A_LSHLA(i) asm volatile("v_lshl_add_u32 %0, %0, 1, -1" : "+v"(x[i]));
#define A_MIN(i)   asm volatile("v_min_u32_e32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
#define A_FMAS(i)  asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(x[i]) : "s"(a), "v"(y[i]));
#define A_NOP(i)   asm volatile("s_nop 1\n\tv_fmac_f32_e32 %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(y[(i + 1) & 7]));
#define A_CMPBR(i) asm volatile("v_cmp_gt_u32_e32 vcc, %1, %0\n\ts_cbranch_vccnz 1f\n1:" : : "v"(x[i]), "s"(0) : "vcc");
#define A_MUL(i)   asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
#define A_FMAN(i)  asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(x[i]) : "v"(y[i]), "v"(y[(i + 1) & 7]));
#define A_MED3V(i) asm volatile("v_med3_f32 %0, %0, 0, %1" : "+v"(x[i]) : "v"(y[7]));
#define A_MED3VV(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y[6]), "v"(y[7]));
#define A_CNDVCC(i) asm volatile("s_mov_b64 vcc, %2\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(y[i]), "s"(mask) : "vcc");
#define A_CNDVCC1(i) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(y[i]));
#define A_CMP(i)   asm volatile("v_cmp_gt_u32_e32 vcc, %1, %0" : : "v"(x[i]), "s"(0) : "vcc");
#define A_CMPS(i)  asm volatile("v_cmp_gt_u32_e64 s[40:41], %1, %0" : : "v"(x[i]), "s"(0) : "s40", "s41");
#define A_MIN3(i)  asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(y[(i + 1) & 7]));
#define A_MINF(i)  asm volatile("v_min_f32_e32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
#define A_MAX3F(i) asm volatile("v_max3_f32 %0, |%0|, |%1|, |%2|" : "+v"(x[i]) : "v"(y[i]), "v"(y[(i + 1) & 7]));
#define A_ADDU(i)  asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
#define A_MULLEG(i) asm volatile("v_mul_legacy_f32_e32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
        if (OP == 20) { REP8(A_MED3V) } else if (OP == 21) { REP8(A_MED3VV) } else if (OP == 22) { REP8(A_CNDVCC) } else if (OP == 23) { REP8(A_CNDVCC1) }
        else if (OP == 24) { REP8(A_CMP) } else if (OP == 25) { REP8(A_CMPS) } else if (OP == 26) { REP8(A_MIN3) } else if (OP == 27) { REP8(A_MINF) }
        else if (OP == 28) { REP8(A_MAX3F) } else if (OP == 29) { REP8(A_ADDU) }
        else if (OP == 30) {      // the row with the cheaper forms: VGPR gamma / omega / 255, VCC select, no s_nop
#define ROW2(i, j) A_DPP(i) A_FMA(j) A_FDPP(j) A_FMAC(j) A_FMAC(j) A_LSHLA(i) A_MUL(j) A_FMAN(j) A_FMAC(j) A_MED3V(j) A_SUB(j) A_FMA(j) A_SUB(j) A_FMA(j) A_CNDVCC(j) A_MIN(i)
            ROW2(0, 1) ROW2(2, 3) ROW2(4, 5) ROW2(6, 7)
        }
        else if (OP == 0) { REP8(A_FMA) } else if (OP == 1) { REP8(A_FMAC) } else if (OP == 2) { REP8(A_MED3) } else if (OP == 3) { REP8(A_CND) }
        else if (OP == 4) { REP8(A_DPP) } else if (OP == 5) { REP8(A_FDPP) } else if (OP == 6) { REP8(A_NFDPP) } else if (OP == 7) { REP8(A_SUB) }
        else if (OP == 8) { REP8(A_LSHLA) } else if (OP == 9) { REP8(A_MIN) } else if (OP == 10) { REP8(A_FMAS) } else if (OP == 11) { REP8(A_NOP) }
        else if (OP == 12) { REP8(A_CMPBR) } else if (OP == 13) { REP8(A_MUL) } else if (OP == 14) { REP8(A_FMAN) }
        else if (OP == 15) {      // one pixel-row of the sweep: the forms in their proportions (16 instructions), twice, on independent registers
#define ROW(i, j) A_DPP(i) A_FMA(j) A_NFDPP(j) A_FMAC(j) A_FMAC(j) A_LSHLA(i) A_MUL(j) A_FMAN(j) A_FMAC(j) A_MED3(j) A_SUB(j) A_FMAS(j) A_SUB(j) A_FMAS(j) A_CND(j) A_MIN(i)
            ROW(0, 1) ROW(2, 3) ROW(4, 5) ROW(6, 7)
        }
    }
    float s = 0; for (int i = 0; i < 8; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP>
void run(const char *name, int ops_per_iter, float *out) {
    printf("%-34s", name);
    for (int wpc : {4, 8, 16, 32}) {   // waves per CU (256 CUs)
        int threads = 256, blocks = 256 * wpc / 4;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<OP><<<blocks, threads>>>(out, 1.0001f, 255.0f, 0x5555aaaa3333ccccull);
        hipEventRecord(e0);
        k<OP><<<blocks, threads>>>(out, 1.0001f, 255.0f, 0x5555aaaa3333ccccull);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double wave_instr = (double)blocks * 4 * N_ITER * ops_per_iter;
        printf("  %2d w/CU: %5.2f", wpc, ms * 1e-3 * 2.4e9 * 1024 / wave_instr);
    }
    printf("   cycles per wave-instruction per SIMD (@2.4 GHz nominal)\n");
}
int main() {
    float *out; hipMalloc(&out, 256 * 32 * 64 * 4 * 4);
    run<0>("v_fma_f32 (3 VGPR)", 8, out); run<1>("v_fmac_f32_e32", 8, out); run<13>("v_mul_f32_e32", 8, out); run<7>("v_sub_f32_e32", 8, out);
    run<14>("v_fma_f32 -v (neg modifier)", 8, out); run<10>("v_fma_f32 with an SGPR operand", 8, out);
    run<2>("v_med3_f32 v, v, 0, s", 8, out); run<3>("v_cndmask_b32_e64 (SGPR mask)", 8, out);
    run<8>("v_lshl_add_u32", 8, out); run<9>("v_min_u32_e32", 8, out);
    run<4>("v_mov_b32_dpp wave_shr:1", 8, out); run<5>("v_fmac_f32_dpp wave_shl:1", 8, out); run<6>("s_nop 1 + v_fmac_f32_dpp", 8, out);
    run<11>("s_nop 1 + v_fmac_f32_e32", 8, out); run<12>("v_cmp + s_cbranch_vccnz", 8, out);
    run<15>("sweep row mix (16 VALU + 1 s_nop)", 64, out);
    run<30>("sweep row mix, cheaper forms", 64, out);
    run<20>("v_med3_f32 v, v, 0, v", 8, out); run<21>("v_med3_f32 v, v, v, v", 8, out);
    run<22>("s_mov vcc + v_cndmask_b32_e32", 8, out); run<23>("v_cndmask_b32_e32 (vcc as is)", 8, out);
    run<24>("v_cmp_gt_u32_e32 vcc", 8, out); run<25>("v_cmp_gt_u32_e64 s[..]", 8, out);
    run<26>("v_min3_u32", 8, out); run<27>("v_min_f32_e32", 8, out); run<28>("v_max3_f32 |.|", 8, out); run<29>("v_add_u32_e32", 8, out);
    return 0;
}
