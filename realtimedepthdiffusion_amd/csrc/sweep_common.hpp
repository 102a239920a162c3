// sweep_common.hpp -- device helpers of the register-resident sweep kernels (sweep_blocked.hip): DPP lane
// shifts, the write-through store / sc1 load of the inter-workgroup hand-off, the 3-operation divide and the rounded reciprocal.
#pragma once
#include <hip/hip_runtime.h>

namespace rtdd {

// bound_ctrl: the lane without a source (lane 0 / lane 63) reads 0 -- its value is never used (the weight towards it is 0
// or the lane lies in the discarded halo) -- and the builtin needs no copy of `v` for the unwritten lane.
__device__ __forceinline__ float lane_from_prev(float v) {   // lane l <- lane l-1  (DPP wave_shr:1)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_from_next(float v) {   // lane l <- lane l+1  (DPP wave_shl:1)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xF, 0xF, true));
}

// The same two shifts through the LDS crossbar (ds_bpermute_b32: no LDS memory, no VALU slot), issued a group of rows ahead so that
// their latency hides behind the wait for the neighbouring waves' rows: k_sweep_blocked measured +3-5 % with them (1080p 1.17 -> 1.21
// Tpx-it/s, one tile alone 0.996 -> 0.945 us per sweep, same call).  What pointed here: the sweep's instruction mix replayed as a
// micro-benchmark issues at 3.26 cycles per wave-instruction and SIMD with its two v_mov_b32_dpp per row and at 2.27 with two
// ds_bpermute_b32 in their place (scripts/ubench/gen_block_bench.py, profiles/r03_block_replay.txt) -- in the kernel most of that
// difference does not show, and the column-layout kernel (8 shifts per thread and sweep) is SLOWER this way (EXPERIMENTS.md).
// `prev4` = 4 * ((lane - 1) & 63), held in a register by the caller; lane l + 1 is prev4 + 8 (the unit takes address bits 7:2).  Lane 0
// reads lane 63 and lane 63 lane 0 where DPP's bound_ctrl gave 0: both only ever meet a zero weight (sweep_tile_setup.inc: the right
// weight of a tile row's last pixel is 0, and lane 63 ends a tile row for every LX), and the values are finite.  The result arrives
// like an LDS read: the compiler waits on lgkmcnt before its first use.
__device__ __forceinline__ float lds_from_prev(int prev4, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(prev4, __float_as_int(v))); }
__device__ __forceinline__ float lds_from_next(int prev4, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(prev4 + 8, __float_as_int(v))); }

// write-through (sc1) 16-byte store for inter-workgroup hand-offs (no release fence needed; the storing wave drains vmcnt itself)
__device__ __forceinline__ void store_sc1(float4 *p, float4 v) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    const f4v t = {v.x, v.y, v.z, v.w};
#ifdef RTDD_TIMING_PLAIN_STRIPS
    // TIMING-ONLY diagnostic build (scripts/build_variant.sh; never the product: cross-XCD neighbours read stale strips): what the exchange
    // would cost if EVERY strip could be stored plain -- kept in the storing XCD's L2 -- i.e. the upper bound of storing same-XCD strips
    // without write-through (EXPERIMENTS.md round 6)
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
    return;
#endif
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
}

// 16-byte sc1 load to registers (bypasses this CU's L1, served by L2 / memory): with EVERY load of handed-off bytes of this form, every
// store of them sc1 and drained, and the flag protocol of persist_sync.hpp, the consumer needs no agent-scope acquire
// (MI355X_MICROARCH.md, "Valid forms", table row 1).  The value is NOT there when the statement returns: wait_loads() below.
typedef float f4v_t __attribute__((ext_vector_type(4)));
// `dst` is a read-write operand: on the path AROUND a conditional load the register keeps its old value, so the register allocator has no
// reason to give the load a register of its own and copy it at the join -- in front of the wait, where the copy would read stale bits.
// tests/test_isa_hazards.py checks in the disassembly that nothing reads or writes a load's registers before the next s_waitcnt vmcnt(0).
__device__ __forceinline__ void load_sc1(f4v_t &dst, const float *p) {
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "+v"(dst) : "v"(p) : "memory");
}

// Result stores: plain (non-temporal and write-through stores measured no faster: EXPERIMENTS.md).
__device__ __forceinline__ void store_result(float4 *p, float4 v) { *p = v; }

// ---- IEEE f32 division with a loop-invariant divisor ---------------------------------------------
// `sum / cnt` must be the correctly rounded quotient (the reference relies on nvcc's default
// -prec-div=true).  hipcc expands an IEEE divide into 11 VALU ops (v_div_scale x2, v_rcp, 5 fma,
// v_mul, v_div_fmas, v_div_fixup): measured 46 cycles per wave-instruction group, 60 % of a sweep.
// The divisor cnt is constant for a pixel, so its correctly rounded reciprocal y = RN(1/cnt) is
// computed once per launch (one full divide) and each sweep does Markstein's correction step
//      q0 = n*y;  r = fma(-d, q0, n);  q = fma(r, y, q0)
// which IS RN(n/d): verified EXHAUSTIVELY on gfx950 against hipcc's divide for all 2^23 divisor x 2^24
// numerator significands (scripts/ubench/div_exhaustive.hip, 1.4e14 pairs, 0 mismatches, 71 s; log in
// profiles/).  Powers of two scale every intermediate exactly, so that covers all operands for which no
// intermediate under/overflows: d normal (<= 4 here), n == 0 or |n| >= 2^-100 (then q0 is normal and the
// remainder, a multiple of 2^(e_n - 47), is exactly representable), n/d bounded (a weighted mean).
// Anything else takes the full divide under a wave-uniform branch.
__device__ __forceinline__ float div_tail(float n, float d, float y) {
    const float q0 = n * y;
    const float r = __builtin_fmaf(-d, q0, n);
    return __builtin_fmaf(r, y, q0);
}

// RN(1/d) for a normal d with a normal reciprocal: v_rcp_f32 (1 ulp) + one Newton step.  Equal to the IEEE quotient 1.0f/d for
// EVERY such f32 (all 4 227 858 434 of them checked on the GPU, scripts/ubench/rcp_exhaustive.hip, profiles/r01_rcp_exhaustive.log):
// 3 VALU ops instead of the 11 of the full divide, in the per-launch setup of every pixel.
__device__ __forceinline__ float rcp_rn(float d) {
    const float y0 = __builtin_amdgcn_rcpf(d);
    return __builtin_fmaf(__builtin_fmaf(-d, y0, 1.0f), y0, y0);
}

// ---- the last operation of a pixel's update under an EXEC mask ----------------------------------------------------------------
// x_{k+1} = fma(omega, t, x_{k-1}) overwrites x_{k-1}'s register -- for FREE pixels only; a Dirichlet pixel keeps its value, and its
// register holds that value in both iterates (k_prepare stores it in both planes, sweep_tile_setup.inc enforces it on the tile).  So
// the select of `(dirichlet ? x : v)` -- a v_cndmask_b32 with an SGPR-pair mask, half rate on gfx950 (4.4 cycles per wave-instruction
// against 3.1 for v_fmac_f32 with three VGPRs and 4.3 for any fma with an SGPR operand: scripts/ubench/excp_probe.hip,
// profiles/r05_excp_probe.txt) -- becomes the EXEC mask of the fma itself: s_mov_b64 exec (scalar unit) + v_fmac_f32 (full rate), omega
// in a VGPR.  ONE asm statement per row so that the compiler can schedule nothing between the EXEC writes.  Round 6: the
// statement no longer ASSUMES that EXEC is all ones on entry (true today: every thread of the whole-wave workgroup runs the sweeps and
// all control flow around them is wave-uniform -- but a compiler that sank the statement into a divergent region would have had its
// inactive lanes switched on by the old closing `s_mov_b64 exec, -1`): the first mask is applied by s_and_saveexec_b64, which also SAVES
// the entry EXEC; the other masks are `entry & mask`; the last instruction puts the entry EXEC back.  Same five scalar instructions per
// row as before, one SGPR pair more for the duration of the statement.  An SALU write of EXEC needs no wait state in front of a
// (non-DPP) VALU instruction on gfx9.
__device__ __forceinline__ void masked_fmac4(float &o0, float &o1, float &o2, float &o3, float w, float t0, float t1, float t2, float t3,
                                             unsigned long long m0, unsigned long long m1, unsigned long long m2, unsigned long long m3) {
    unsigned long long entry;                // EXEC as it was on entry: saved by the instruction that applies the first mask, put back by the last
    asm("s_and_saveexec_b64 %[e], %[m0]\n\tv_fmac_f32_e32 %[o0], %[w], %[t0]\n\t"
        "s_and_b64 exec, %[e], %[m1]\n\tv_fmac_f32_e32 %[o1], %[w], %[t1]\n\t"
        "s_and_b64 exec, %[e], %[m2]\n\tv_fmac_f32_e32 %[o2], %[w], %[t2]\n\t"
        "s_and_b64 exec, %[e], %[m3]\n\tv_fmac_f32_e32 %[o3], %[w], %[t3]\n\t"
        "s_mov_b64 exec, %[e]"
        : [o0] "+v"(o0), [o1] "+v"(o1), [o2] "+v"(o2), [o3] "+v"(o3), [e] "=&s"(entry)
        : [w] "v"(w), [t0] "v"(t0), [t1] "v"(t1), [t2] "v"(t2), [t3] "v"(t3), [m0] "s"(m0), [m1] "s"(m1), [m2] "s"(m2), [m3] "s"(m3)
        : "scc");
}
// the same for the un-contracted update, (omega * t) + x_{k-1}: the product is formed outside, the addition is masked
__device__ __forceinline__ void masked_add4(float &o0, float &o1, float &o2, float &o3, float p0, float p1, float p2, float p3,
                                            unsigned long long m0, unsigned long long m1, unsigned long long m2, unsigned long long m3) {
    unsigned long long entry;
    asm("s_and_saveexec_b64 %[e], %[m0]\n\tv_add_f32_e32 %[o0], %[p0], %[o0]\n\t"
        "s_and_b64 exec, %[e], %[m1]\n\tv_add_f32_e32 %[o1], %[p1], %[o1]\n\t"
        "s_and_b64 exec, %[e], %[m2]\n\tv_add_f32_e32 %[o2], %[p2], %[o2]\n\t"
        "s_and_b64 exec, %[e], %[m3]\n\tv_add_f32_e32 %[o3], %[p3], %[o3]\n\t"
        "s_mov_b64 exec, %[e]"
        : [o0] "+v"(o0), [o1] "+v"(o1), [o2] "+v"(o2), [o3] "+v"(o3), [e] "=&s"(entry)
        : [p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2), [p3] "v"(p3), [m0] "s"(m0), [m1] "s"(m1), [m2] "s"(m2), [m3] "s"(m3)
        : "scc");
}

}  // namespace rtdd
