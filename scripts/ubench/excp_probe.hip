// micro-benchmark / probe: can the wave's sticky floating-point exception bits (TRAPSTS.EXCP, gfx950) stand in for the per-pixel "tiny
// numerator" test of the 3-operation divide (csrc/sweep_common.hpp div_tail)?
//   part 1  semantics: which bits a v_mul_f32 / v_fma_f32 sets for (normal, inexact), (underflow, inexact), (denormal result, exact),
//           whether they accumulate with EXCP_EN = 0 (no trap handler), whether s_setreg clears them, and whether an s_getreg issued
//           DIRECTLY behind the VALU instruction (no wait states) already sees them;
//   part 2  soundness: for tiny / borderline / normal numerators n and divisors d (normal, <= 4), y = RN(1/d): whenever the 3-operation
//           quotient differs from the IEEE quotient n / d, the wave's UNDERFLOW bit was set by those three instructions (per wave: one
//           lane carries the hard case, the other 63 carry harmless ones); and how often the bit is set when it need not be;
//   part 3  cost: s_getreg + s_cmp + s_cbranch per 16 VALU instructions, s_mov exec + v_fmac against v_cndmask, VGPR against SGPR omega.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/ubench/excp_probe.hip -o /tmp/excp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <cstring>

#define CLEAR_EXCP() asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_TRAPSTS, 0, 9), 0" ::: "memory")

__device__ __forceinline__ uint32_t read_excp() {
    uint32_t r;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 9)" : "=s"(r)::"memory");
    return r;
}

// ---- part 1 --------------------------------------------------------------------------------------------------------------------
__global__ void k_semantics(uint32_t *out, float tiny, float one_third, float three, float den_exact, float onef) {
    uint32_t r[12];
    float v;
    CLEAR_EXCP();
    r[0] = read_excp();
    // normal * normal, inexact: expect INEXACT only (bit 5)
    asm volatile("v_mul_f32 %0, %2, %3\n\ts_getreg_b32 %1, hwreg(HW_REG_TRAPSTS, 0, 9)" : "=v"(v), "=s"(r[1]) : "v"(one_third), "v"(three) : "memory");
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 9)" : "=s"(r[2])::"memory");
    CLEAR_EXCP();
    r[3] = read_excp();
    // tiny * tiny -> 0, inexact underflow: expect UNDERFLOW (bit 4) + INEXACT, read with no wait state at all, then late
    asm volatile("v_mul_f32 %0, %2, %2\n\ts_getreg_b32 %1, hwreg(HW_REG_TRAPSTS, 0, 9)" : "=v"(v), "=s"(r[4]) : "v"(tiny) : "memory");
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 9)" : "=s"(r[5])::"memory");
    // sticky: a clean instruction behind it must not clear it
    asm volatile("v_mul_f32 %0, %2, %2\n\ts_nop 7\n\ts_getreg_b32 %1, hwreg(HW_REG_TRAPSTS, 0, 9)" : "=v"(v), "=s"(r[6]) : "v"(onef) : "memory");
    CLEAR_EXCP();
    // denormal result, EXACT (2^-130 * 1.0): IEEE raises no underflow flag when the tiny result is exact
    asm volatile("v_mul_f32 %0, %2, %3\n\ts_nop 7\n\ts_getreg_b32 %1, hwreg(HW_REG_TRAPSTS, 0, 9)" : "=v"(v), "=s"(r[7]) : "v"(den_exact), "v"(onef) : "memory");
    CLEAR_EXCP();
    // only ONE lane underflows: the bit is per wave
    float t = (threadIdx.x & 63) == 17 ? tiny : onef;
    asm volatile("v_mul_f32 %0, %2, %2\n\ts_nop 7\n\ts_getreg_b32 %1, hwreg(HW_REG_TRAPSTS, 0, 9)" : "=v"(v), "=s"(r[8]) : "v"(t) : "memory");
    CLEAR_EXCP();
    // fma whose exact result is tiny and inexact
    asm volatile("v_fma_f32 %0, %2, %2, %3\n\ts_nop 7\n\ts_getreg_b32 %1, hwreg(HW_REG_TRAPSTS, 0, 9)" : "=v"(v), "=s"(r[9]) : "v"(tiny), "v"(den_exact) : "memory");
    CLEAR_EXCP();
    // an EXEC-masked lane must not contribute: the underflowing lane is switched off
    {
        uint64_t m = ~(1ull << 17);
        asm volatile("s_mov_b64 exec, %3\n\tv_mul_f32 %0, %2, %2\n\ts_mov_b64 exec, -1\n\ts_nop 7\n\ts_getreg_b32 %1, hwreg(HW_REG_TRAPSTS, 0, 9)" : "=v"(v), "=s"(r[10]) : "v"(t), "s"(m) : "memory");
    }
    CLEAR_EXCP();
    r[11] = read_excp();
    if (threadIdx.x == 0) for (int i = 0; i < 12; i++) out[i] = r[i];
}

// ---- part 2 --------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rng(uint32_t &s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
__device__ __forceinline__ float rcp_rn(float d) { const float y0 = __builtin_amdgcn_rcpf(d); return __builtin_fmaf(__builtin_fmaf(-d, y0, 1.0f), y0, y0); }

// mode 0: every lane a tiny numerator (exponent field 0..40, i.e. denormal .. 2^-87)
// mode 1: ONE lane per wave tiny, the others ordinary (|n| in [2^-20, 2^10], or exactly 0)
// mode 2: every lane ordinary -- the bit must never be set (spurious slow paths)
// mode 3: numerators around the 2^-100 threshold (exponent field 20..34)
__global__ void k_soundness(unsigned long long *counts, int mode, uint32_t seed, int iters) {
    uint32_t s = seed ^ (blockIdx.x * 9781u + threadIdx.x * 6271u + 1u);
    for (int i = 0; i < 8; i++) rng(s);
    unsigned long long mismatch_unflagged = 0, mismatch_flagged = 0, flagged = 0, waves = 0;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; it++) {
        const uint32_t r0 = rng(s), r1 = rng(s), r2 = rng(s);
        const int hot = __builtin_amdgcn_readfirstlane(r2) & 63;
        // divisor: a sum of up to four weights in (0, 1]: normal, <= 4; exponent field 1..129
        const uint32_t de = 1u + (r1 >> 23) % 129u;
        const float d = __uint_as_float((de << 23) | (r1 & 0x7FFFFFu));
        uint32_t ne;
        const bool tiny_lane = mode == 0 || mode == 3 || (mode == 1 && lane == hot);
        if (mode == 3) ne = 20u + (r0 >> 23) % 15u;
        else if (tiny_lane) ne = (r0 >> 23) % 41u;
        else ne = 107u + (r0 >> 23) % 31u;
        float n = __uint_as_float((r0 & 0x80000000u) | (ne << 23) | (r0 & 0x7FFFFFu));
        if (!tiny_lane && (r2 & 0xF00u) == 0) n = 0.0f;
        // the quotient must stay bounded like a weighted mean does: skip (make harmless) pairs whose quotient would overflow
        if (fabsf(n) > 1024.0f * d) n = 0.0f;
        const float y = rcp_rn(d);
        const float ref = n / d;                          // hipcc's IEEE divide (-fhip-fp32-correctly-rounded-divide-sqrt is the default for HIP)
        float q0, rr, q;
        uint32_t ex;
        asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_TRAPSTS, 0, 9), 0\n\t"
                     "v_mul_f32 %0, %4, %6\n\t"
                     "v_fma_f32 %1, -%5, %0, %4\n\t"
                     "v_fma_f32 %2, %1, %6, %0\n\t"
                     "v_med3_f32 %2, %2, %2, %2\n\t"          // three dependent instructions, as in the sweep (clamp, r - x, gamma)
                     "v_sub_f32 %1, %2, %1\n\t"
                     "v_add_f32 %1, %1, %1\n\t"
                     "s_getreg_b32 %3, hwreg(HW_REG_TRAPSTS, 0, 9)"
                     : "=&v"(q0), "=&v"(rr), "=&v"(q), "=s"(ex) : "v"(n), "v"(d), "v"(y) : "memory");
        // (the three trailing instructions can raise flags of their own -- only ever MORE bits: conservative)
        const bool mis = __float_as_uint(q) != __float_as_uint(ref);
        const bool any_mis = __builtin_amdgcn_ballot_w64(mis) != 0;
        const bool under = (ex & 0x10u) != 0;
        if (lane == 0) {
            waves++;
            if (under) flagged++;
            if (any_mis) { if (under) mismatch_flagged++; else mismatch_unflagged++; }
        }
    }
    if (lane == 0) {
        atomicAdd(&counts[0], waves); atomicAdd(&counts[1], flagged); atomicAdd(&counts[2], mismatch_flagged); atomicAdd(&counts[3], mismatch_unflagged);
    }
}

// ---- part 3 --------------------------------------------------------------------------------------------------------------------
#define N_ITER 4000
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP>
__global__ void k_cost(float *out, float a, float b, unsigned long long mask) {
    float x[8], y[8];
    for (int i = 0; i < 8; i++) { x[i] = threadIdx.x + i; y[i] = threadIdx.x * 0.5f + i; }
    float av = a;
    asm volatile("" : "+v"(av));
    uint32_t acc = 0;
    for (int it = 0; it < N_ITER; it++) {
#define A_FMA(i)   asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(y[i]), "v"(y[(i + 1) & 7]));
#define A_CND(i)   asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "s"(mask));
#define A_FMAS(i)  asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(x[i]) : "s"(a), "v"(y[i]));
#define A_FMACS(i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(x[i]) : "s"(a), "v"(y[i]));
#define A_FMACV(i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(x[i]) : "v"(av), "v"(y[i]));
#define A_XFMAC(i) asm volatile("s_mov_b64 exec, %2\n\tv_fmac_f32_e32 %0, %1, %3" : "+v"(x[i]) : "v"(av), "s"(mask), "v"(y[i]));
#define A_AND(i)   asm volatile("v_and_b32_e32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
#define A_OR(i)    asm volatile("v_or_b32_e32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
#define A_MAXF(i)  asm volatile("v_max_f32_e32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
#define A_LSHL(i)  asm volatile("v_lshlrev_b32_e32 %0, 1, %0" : "+v"(x[i]));
#define A_ADD0(i)  asm volatile("v_add_f32_e32 %0, 0, %0" : "+v"(x[i]));
#define A_MULC(i)  asm volatile("v_mul_f32_e64 %0, %0, %1 clamp" : "+v"(x[i]) : "v"(y[i]));
        if (OP == 0) { REP8(A_FMA) REP8(A_FMA) }
        else if (OP == 1) {     // 16 fma + exception read + compare + branch
            REP8(A_FMA) REP8(A_FMA)
            uint32_t ex;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 4, 1)" : "=s"(ex)::"memory");
            if (__builtin_expect(ex != 0, 0)) { acc++; CLEAR_EXCP(); }
        }
        else if (OP == 2) {     // 4 of them per 16 fma (one per row of four pixels)
#define GRP(i, j, k2, l) A_FMA(i) A_FMA(j) A_FMA(k2) A_FMA(l) { uint32_t ex; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 4, 1)" : "=s"(ex)::"memory"); if (__builtin_expect(ex != 0, 0)) { acc++; CLEAR_EXCP(); } }
            GRP(0, 1, 2, 3) GRP(4, 5, 6, 7) GRP(0, 1, 2, 3) GRP(4, 5, 6, 7)
        }
        else if (OP == 3) { REP8(A_CND) REP8(A_CND) }
        else if (OP == 4) { REP8(A_XFMAC) REP8(A_XFMAC) asm volatile("s_mov_b64 exec, -1"); }
        else if (OP == 5) { REP8(A_FMACV) REP8(A_FMACV) }
        else if (OP == 6) { REP8(A_FMACS) REP8(A_FMACS) }
        else if (OP == 7) { REP8(A_FMAS) REP8(A_FMAS) }
        else if (OP == 8) { REP8(A_AND) REP8(A_AND) }
        else if (OP == 9) { REP8(A_OR) REP8(A_OR) }
        else if (OP == 10) { REP8(A_MAXF) REP8(A_MAXF) }
        else if (OP == 11) { REP8(A_LSHL) REP8(A_LSHL) }
        else if (OP == 12) { REP8(A_ADD0) REP8(A_ADD0) }
        else if (OP == 13) { REP8(A_MULC) REP8(A_MULC) }
        else if (OP == 14) {    // one row of four pixels as the new sweep would issue it: 16 sum fma, 12 divide, 4 med3, 4 sub, 4 fma, 4 sub, getreg, 4 masked fmac
#define PX(i) A_FMA(i) A_FMA(i) A_FMA(i) A_FMA(i) A_FMA(i) A_FMA(i) A_FMA(i) asm volatile("v_med3_f32 %0, %0, 0, %1" : "+v"(x[i]) : "v"(y[7])); A_FMA(i) A_FMA(i) A_FMA(i)
            PX(0) PX(1) PX(2) PX(3)
            { uint32_t ex; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 4, 1)" : "=s"(ex)::"memory"); if (__builtin_expect(ex != 0, 0)) { acc++; CLEAR_EXCP(); } }
            A_XFMAC(0) A_XFMAC(1) A_XFMAC(2) A_XFMAC(3) asm volatile("s_mov_b64 exec, -1");
        }
        else if (OP == 15) {    // the same row as it is issued today: + 4 lshl_add, 2 min, cmp + branch, SGPR gamma / omega, cndmask
#define PY(i) A_FMA(i) A_FMA(i) A_FMA(i) A_FMA(i) asm volatile("v_lshl_add_u32 %0, %1, 1, -1" : "=v"(y[i]) : "v"(x[i])); A_FMA(i) A_FMA(i) A_FMA(i) asm volatile("v_med3_f32 %0, %0, 0, %1" : "+v"(x[i]) : "s"(b)); A_FMA(i) A_FMAS(i) A_FMA(i) A_FMAS(i) A_CND(i)
            PY(0) PY(1) PY(2) PY(3)
            asm volatile("v_min_u32_e32 %0, %0, %1\n\tv_min3_u32 %0, %0, %2, %3\n\tv_cmp_gt_u32_e32 vcc, %4, %0\n\ts_cbranch_vccnz 1f\n1:" : "+v"(y[0]) : "v"(y[1]), "v"(y[2]), "v"(y[3]), "s"(0) : "vcc");
        }
    }
    float s = 0; for (int i = 0; i < 8; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + acc;
}
template <int OP>
void run(const char *name, int ops_per_iter, float *out) {
    printf("%-46s", name);
    for (int wpc : {8, 16}) {   // waves per CU (256 CUs)
        int threads = 256, blocks = 256 * wpc / 4;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k_cost<OP><<<blocks, threads>>>(out, 1.0001f, 255.0f, 0x5555aaaa3333ccccull);
        hipEventRecord(e0);
        k_cost<OP><<<blocks, threads>>>(out, 1.0001f, 255.0f, 0x5555aaaa3333ccccull);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double units = (double)blocks * 4 * N_ITER * ops_per_iter;
        printf("  %2d w/CU: %6.2f", wpc, ms * 1e-3 * 2.4e9 * 1024 / units);
    }
    printf("   cycles per unit per SIMD (@2.4 GHz nominal)\n");
}

int main() {
    uint32_t *o; hipMalloc(&o, 64 * 4);
    float den_exact; { uint32_t b = 1u << 19; memcpy(&den_exact, &b, 4); }     // 2^-130, a denormal
    k_semantics<<<1, 64>>>(o, 1e-30f, 1.0f / 3.0f, 3.0f, den_exact, 1.0f);
    uint32_t h[12]; hipMemcpy(h, o, sizeof h, hipMemcpyDeviceToHost);
    const char *what[12] = {"after clear", "1/3*3 read at once", "1/3*3 read late", "after clear", "tiny*tiny read at once", "tiny*tiny read late",
                            "sticky after a clean mul", "exact denormal result", "one lane of 64 underflows", "fma tiny inexact", "underflowing lane EXEC-masked off", "after clear"};
    printf("TRAPSTS.EXCP bits: 0 invalid, 1 input denormal, 2 div0, 3 overflow, 4 UNDERFLOW, 5 inexact, 6 int div0\n");
    for (int i = 0; i < 12; i++) printf("  %-36s 0x%03x\n", what[i], h[i]);

    unsigned long long *c; hipMalloc(&c, 4 * 8);
    const char *modes[4] = {"every lane tiny", "one lane tiny, 63 ordinary", "every lane ordinary", "around 2^-100"};
    for (int mode = 0; mode < 4; mode++) {
        hipMemset(c, 0, 32);
        k_soundness<<<2048, 256>>>(c, mode, 0x9E3779B9u + mode, 20000);
        unsigned long long hc[4]; hipMemcpy(hc, c, 32, hipMemcpyDeviceToHost);
        printf("soundness %-28s waves %llu  underflow flagged %llu  mismatch&flagged %llu  MISMATCH&UNFLAGGED %llu\n", modes[mode], hc[0], hc[1], hc[2], hc[3]);
    }

    float *out; hipMalloc(&out, 256 * 32 * 64 * 4 * 4);
    run<0>("16 v_fma_f32 (per instruction)", 16, out);
    run<1>("16 v_fma + 1 (getreg, cmp, branch)", 16, out);
    run<2>("16 v_fma + 4 (getreg, cmp, branch)", 16, out);
    run<3>("v_cndmask_b32_e64 SGPR mask", 16, out);
    run<4>("s_mov exec + v_fmac_f32 (VGPR omega)", 16, out);
    run<5>("v_fmac_f32_e32 VGPR omega", 16, out);
    run<6>("v_fmac_f32_e32 SGPR omega", 16, out);
    run<7>("v_fma_f32 SGPR operand (VOP3)", 16, out);
    run<8>("v_and_b32", 16, out); run<9>("v_or_b32", 16, out); run<10>("v_max_f32", 16, out); run<11>("v_lshlrev_b32", 16, out);
    run<12>("v_add_f32 0, v", 16, out); run<13>("v_mul_f32 clamp", 16, out);
    run<14>("NEW row of 4 px (per pixel)", 4, out);
    run<15>("OLD row of 4 px (per pixel)", 4, out);
    return 0;
}
