// micro-benchmark: issue cost per SIMD of the PACKED f32 forms (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, gfx950) beside the scalar
// forms of valu_mix.hip, and of one "row pair" of a sweep written with them (two vertically adjacent pixels per packed instruction).
// Build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/pk_mix.hip -o scripts/ubench/pk_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 4000
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
typedef float f2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ void k(float *out, float a, float b, unsigned long long mask) {
    f2 x[8], y[8];
    float u[8];
    f2 sg = {a, b};
    for (int i = 0; i < 8; i++) { x[i] = f2{(float)threadIdx.x + i, 1.0f + i}; y[i] = f2{threadIdx.x * 0.5f + i, 0.25f * i}; u[i] = i; }
    for (int it = 0; it < N_ITER; it++) {
#define P_FMA(i)    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(y[i]), "v"(y[(i + 1) & 7]));
#define P_MUL(i)    asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
#define P_ADD(i)    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
#define P_SUB(i)    asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(x[i]) : "v"(y[i]));
#define P_FMASW(i)  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,0,1]" : "+v"(x[i]) : "v"(y[i]), "v"(y[(i + 1) & 7]));
#define P_FMAN(i)   asm volatile("v_pk_fma_f32 %0, %1, %2, %0 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "+v"(x[i]) : "v"(y[i]), "v"(y[(i + 1) & 7]));
#define P_FMAS(i)   asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel_hi:[0,1,1]" : "+v"(x[i]) : "s"(sg), "v"(y[i]));
#define S_FMA(i)    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(u[i]) : "v"(y[i].x), "v"(y[(i + 1) & 7].y));
#define S_FMAH(i)   asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i].y) : "v"(y[i].x), "v"(y[(i + 1) & 7].y));
#define S_FMAL(i)   asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i].x) : "v"(y[i].y), "v"(y[(i + 1) & 7].x));
#define S_MED3L(i)  asm volatile("v_med3_f32 %0, %0, 0, %1" : "+v"(x[i].x) : "s"(b));
#define S_MED3H(i)  asm volatile("v_med3_f32 %0, %0, 0, %1" : "+v"(x[i].y) : "s"(b));
#define S_CNDL(i)   asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[i].x) : "v"(y[i].x), "s"(mask));
#define S_CNDH(i)   asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[i].y) : "v"(y[i].y), "s"(mask));
#define S_LSHLL(i)  asm volatile("v_lshl_add_u32 %0, %1, 1, -1" : "=v"(u[i]) : "v"(x[i].x));
#define S_LSHLH(i)  asm volatile("v_lshl_add_u32 %0, %1, 1, -1" : "=v"(u[(i + 1) & 7]) : "v"(x[i].y));
#define S_MIN3(i)   asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(u[(i + 2) & 7]) : "v"(u[i]), "v"(u[(i + 1) & 7]));
#define S_DPP(i)    asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(u[(i + 3) & 7]) : "v"(y[i].x));
#define P_MOV(i)    asm volatile("v_pk_mov_b32 %0, %1, %2" : "=v"(x[i]) : "v"(y[i]), "v"(y[(i + 1) & 7]));
        if (OP == 0) { REP8(P_FMA) } else if (OP == 1) { REP8(P_MUL) } else if (OP == 2) { REP8(P_ADD) } else if (OP == 3) { REP8(P_SUB) }
        else if (OP == 4) { REP8(P_FMASW) } else if (OP == 5) { REP8(P_FMAN) } else if (OP == 6) { REP8(P_FMAS) } else if (OP == 7) { REP8(S_FMA) }
        else if (OP == 8) { REP8(S_FMAH) } else if (OP == 11) { REP8(P_MOV) }
        else if (OP == 9) {        // two vertically adjacent pixels of a sweep in packed forms: 20 instructions
#define PAIR(i) S_DPP(i) P_MUL(i) P_FMA(i) S_FMAL(i) P_FMASW(i) S_FMAH(i) S_LSHLL(i) S_LSHLH(i) S_MIN3(i) P_MUL(i) P_FMAN(i) P_FMA(i) S_MED3L(i) S_MED3H(i) P_SUB(i) P_FMAS(i) P_SUB(i) P_FMAS(i) S_CNDL(i) S_CNDH(i)
            REP8(PAIR)
        }
        else if (OP == 10) {       // the same with VGPR factors in place of the SGPR pair
#define PAIRV(i) S_DPP(i) P_MUL(i) P_FMA(i) S_FMAL(i) P_FMASW(i) S_FMAH(i) S_LSHLL(i) S_LSHLH(i) S_MIN3(i) P_MUL(i) P_FMAN(i) P_FMA(i) S_MED3L(i) S_MED3H(i) P_SUB(i) P_FMA(i) P_SUB(i) P_FMA(i) S_CNDL(i) S_CNDH(i)
            REP8(PAIRV)
        }
    }
    float s = 0; for (int i = 0; i < 8; i++) s += x[i].x + x[i].y + u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP>
void run(const char *name, int ops_per_iter, float *out) {
    printf("%-44s", name);
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
    run<7>("v_fma_f32 (3 VGPR)", 8, out); run<8>("v_fma_f32 on the high half of a pair", 8, out);
    run<0>("v_pk_fma_f32 (3 VGPR pairs)", 8, out); run<1>("v_pk_mul_f32", 8, out); run<2>("v_pk_add_f32", 8, out); run<3>("v_pk_add_f32 neg", 8, out);
    run<4>("v_pk_fma_f32 op_sel (halves swapped / bcast)", 8, out); run<5>("v_pk_fma_f32 neg", 8, out); run<6>("v_pk_fma_f32 SGPR pair, bcast lo", 8, out);
    run<11>("v_pk_mov_b32", 8, out);
    run<9>("row PAIR in packed forms (20 instr = 2 px)", 160, out);
    run<10>("row PAIR, VGPR factors", 160, out);
    return 0;
}
