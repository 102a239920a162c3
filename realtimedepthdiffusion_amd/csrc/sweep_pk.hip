// sweep_pk.hip -- the register-resident Chebyshev-Jacobi sweeps of sweep_blocked.hip, TWO pixels per vector instruction.
//
// Why: k_sweep_blocked issues ~16 VALU instructions per pixel-sweep and is bound by their issue rate (3.8 cycles per wave-instruction
// and SIMD for its mix, scripts/ubench/valu_mix.hip).  gfx950 has packed f32 forms -- v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32 -- that
// apply one IEEE operation to both halves of a 64-bit register pair at 4.8-5.2 cycles (scripts/ubench/pk_mix.hip,
// profiles/r03_pk_mix.txt): 2.5 cycles per operation instead of 3.2.  Eleven of the sixteen operations of a pixel-sweep (the four
// products of the weighted sum, the 3-operation divide, the four of the Chebyshev update) have a packed form; the clamp, the select
// of Dirichlet pixels, the test for tiny numerators and the DPP lane shifts stay single.  A row pair written this way costs 43
// cycles per pixel in the micro-benchmark against 61 for the scalar row.
//
// Which two pixels: a packed instruction reads both halves of each operand from ONE register pair, so the two pixels of a pair must
// have their neighbours in the same relative places.  The extended tile is cut into an upper and a lower half (P, Q), a thread owns
// the SAME 4-pixel x G-row block in both, and register pair (g, i) holds {P(g, i), Q(g, i)}: left / right / up / down neighbours of a
// pair are again pairs, the weights, divisors and reciprocals are pairs, and every operation of the sweep is the packed form of the
// one in sweep_blocked.hip -- same operations in the same order on every pixel, bit-identical results.
//   * horizontal neighbours across lanes: DPP wave shifts of each half (no packed DPP form exists);
//   * vertical neighbours across thread rows: through LDS, 32 bytes per thread and row ({P0 Q0 P1 Q1}, {P2 Q2 P3 Q3}: what is read
//     back IS four register pairs);
//   * the seam: the row below P's last row is Q's first row.  The last thread row's "down" pairs must be {Q-top of thread row 0, -}
//     and thread row 0's "up" pairs {-, P-bottom of the last thread row}, i.e. the other thread row's published row with its halves
//     SWAPPED.  The two thread rows at the seam publish a second, swapped copy of that row (a branch only their waves take); the
//     readers just use a different LDS address.  The waves therefore form a ring: wave 0 and the last wave are neighbours.
//   * the halves that are nobody's neighbour (above P's first row, below Q's last) read some finite value with weight 0 or lie in the
//     discarded halo, as in sweep_blocked.hip.
// Global memory stays in the planes' layout: tile loads, halo exchange and write-back gather / scatter the halves of the pairs
// (a few moves per 16-byte access, once per block of sweeps).
#include <cstdlib>
#include <type_traits>

#include "rtdd_internal.hpp"
#include "persist_sync.hpp"
#include "sweep_common.hpp"

namespace rtdd {

#ifndef RTDD_TL        // (scripts/ubench/pk_timeline.hip defines it: per-wave, per-sweep s_memtime stamps of one workgroup)
#define RTDD_TL(k, sw) do {} while (0)
#endif

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f8 __attribute__((ext_vector_type(8)));           // one tile row of a thread: {P0 Q0 P1 Q1 P2 Q2 P3 Q3}

__device__ __forceinline__ f2 pr(const f8 &r, int i) { return f2{r[2 * i], r[2 * i + 1]}; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

template <int LX, int NT, int G, int MINW, bool CONTRACT, bool PERSIST>
__global__ __launch_bounds__(NT, MINW) void k_sweep_pk(float *Xk, float *Xm, float *Yk, float *Ym,
                                                       const uint32_t *__restrict__ M, const float *__restrict__ lut_g,
                                                       const float *__restrict__ omegas, int ip, int rows, int cols,
                                                       int hx, int hy, int nsweeps, float gamma,
                                                       int block_sweeps, int *sync_words, int gx, int gy, int xcd_tiles, int flag_base) {
    // arguments, tile numbering, persistent protocol: as k_sweep_blocked (sweep_blocked.hip)
    constexpr int EW = 4 * LX, NTR = NT / LX, NSLOT = 2 * NTR + 2;
    __shared__ float lut[257];
    __shared__ float4 edge[2][NSLOT][2][LX];   // [buffer][slot][pixels 0-1 / 2-3][lane]; slot 2*tr = thread row tr's top row, 2*tr+1 its bottom row,
                                               // 2*NTR = the LAST thread row's bottom row, halves swapped, 2*NTR+1 = thread row 0's top row, halves swapped
    __shared__ int published[NT / 64 + 1];
    __shared__ int dead_s, seen_s;

    int bx = blockIdx.x, by = blockIdx.y;
    if (xcd_tiles > 0) {
        const int t = ((int)blockIdx.x & 7) * xcd_tiles + ((int)blockIdx.x >> 3);
        if (t >= gx * gy) return;
        bx = t % gx; by = t / gx;
    }
    const int tid = threadIdx.x;
    const int lx = tid % LX, tr = tid / LX;
    const int ntr = (int)blockDim.x / LX;          // thread rows actually launched
    const int sh = ntr * G, eh = 2 * sh;           // rows of one half, of the extended tile
    const int TW = EW - 2 * hx, TH = eh - 2 * hy;
    const int x0 = bx * TW - hx + 4 * lx;
    const int y0 = by * TH - hy + tr * G;          // P's rows: y0 + g, Q's rows: y0 + sh + g
    const bool colok = x0 >= 0 && x0 < cols;

    f8 a[G], b[G];                                 // a = x_k, b = x_{k-1}; roles alternate every sweep
    f8 wr[G], wd[G], cnt[G], rcp[G], wu0;
    bool unsafe = false;
    uint32_t dirichlet = 0;                        // bit (g*4 + i)*2 + half

    // ---- load the extended tile once (every load is issued before the weight table is staged: one memory round trip) ----
    float4 vxr[2][G], vpr[2][G];
    uint4 mr[2][G], mup[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int y = y0 + h * sh + g;
            vxr[h][g] = make_float4(0, 0, 0, 0); vpr[h][g] = vxr[h][g]; mr[h][g] = make_uint4(0, 0, 0, 0);
            if (colok && y >= 0 && y < rows) {
                const size_t off = (size_t)y * ip + x0;
                vxr[h][g] = *(const float4 *)(Xk + off);
                vpr[h][g] = *(const float4 *)(Xm + off);
                mr[h][g] = *(const uint4 *)(M + off);
            }
        }
        const int yu = y0 + h * sh - 1;                       // the row above the block: its down-weights
        mup[h] = make_uint4(0, 0, 0, 0);
        if (colok && yu >= 0 && yu + 1 < rows) mup[h] = *(const uint4 *)(M + (size_t)yu * ip + x0);
    }
    for (int i = tid; i < 257; i += (int)blockDim.x) lut[i] = lut_g[i];
    if (tid <= NT / 64) published[tid] = 0;
    if (tid == 0) { dead_s = PERSIST && __hip_atomic_load(&sync_words[kSyncStatus], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0; seen_s = flag_base; }
    __syncthreads();
    if (PERSIST && dead_s) return;

    // ---- registers of the tile: as sweep_tile_setup.inc, per half ----
#pragma unroll
    for (int g = 0; g < G; g++) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int y = y0 + h * sh + g;
            const bool ok = colok && y >= 0 && y < rows;
            const float xv[4] = {vxr[h][g].x, vxr[h][g].y, vxr[h][g].z, vxr[h][g].w}, pv[4] = {vpr[h][g].x, vpr[h][g].y, vpr[h][g].z, vpr[h][g].w};
            const uint32_t mv[4] = {mr[h][g].x, mr[h][g].y, mr[h][g].z, mr[h][g].w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const bool in = ok && x0 + i < cols;
                a[g][2 * i + h] = in ? xv[i] : 0.0f;
                b[g][2 * i + h] = in ? pv[i] : 0.0f;
                wr[g][2 * i + h] = (in && x0 + i + 1 < cols) ? lut[mv[i] & 255] : 0.0f;
                wd[g][2 * i + h] = (in && y + 1 < rows) ? lut[(mv[i] >> 8) & 255] : 0.0f;
                if (in && (mv[i] & kMetaDirichlet)) dirichlet |= 1u << ((g * 4 + i) * 2 + h);
            }
            // (the right weight of a tile row's LAST pixel is 0: sweep_tile_setup.inc)
            if (lx == LX - 1) wr[g][6 + h] = 0.0f;
        }
    }
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const uint32_t mv[4] = {mup[h].x, mup[h].y, mup[h].z, mup[h].w};
#pragma unroll
        for (int i = 0; i < 4; i++) wu0[2 * i + h] = (x0 + i < cols) ? lut[(mv[i] >> 8) & 255] : 0.0f;       // (mup is 0 -> index 0 where the row is absent: masked below)
        const int yu = y0 + h * sh - 1;
        if (!(colok && yu >= 0 && yu + 1 < rows)) {
#pragma unroll
            for (int i = 0; i < 4; i++) wu0[2 * i + h] = 0.0f;
        }
    }
#pragma unroll
    for (int g = 0; g < G; g++) {
        const f2 wl0 = f2{lane_from_prev(wr[g][6]), lane_from_prev(wr[g][7])};
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const float wl = i == 0 ? wl0[h] : wr[g][2 * (i - 1) + h];
                const float wu = g == 0 ? wu0[2 * i + h] : wd[g - 1][2 * i + h];
                float c = 0.0f;                    // count accumulates left, right, up, down (src/GPUSolver.cu:82,88,94,100)
                c += wl; c += wr[g][2 * i + h]; c += wu; c += wd[g][2 * i + h];
                c = c == 0.0f ? 1.0f : c;
                cnt[g][2 * i + h] = c;
                rcp[g][2 * i + h] = rcp_rn(c);
                unsafe |= c < 0x1p-126f;
            }
    }
    const bool wave_unsafe = __builtin_amdgcn_ballot_w64(unsafe) != 0;

    // ---- the per-wave hand-off of edge rows through LDS (sweep_tile_sweeps.inc), on a ring of waves ----
    constexpr uint32_t kTinyT = 2u * 0x0D800000u - 1u;          // bits(2^-100) = 27 << 23
    const int wv = tid >> 6, nwv = (int)blockDim.x >> 6;
    const int tile_id_tl = by * gx + bx; (void)tile_id_tl;
    const int slot_up = tr > 0 ? 2 * (tr - 1) + 1 : 2 * NTR;            // thread row 0: the last thread row's bottom row, swapped
    const int slot_dn = tr < ntr - 1 ? 2 * (tr + 1) : 2 * NTR + 1;      // the last thread row: thread row 0's top row, swapped
    auto publish = [&](const f8 &top, const f8 &bottom, int sweep_no, int buf) {
        *(float4 *)&edge[buf][2 * tr][0][lx] = make_float4(top[0], top[1], top[2], top[3]);
        *(float4 *)&edge[buf][2 * tr][1][lx] = make_float4(top[4], top[5], top[6], top[7]);
        *(float4 *)&edge[buf][2 * tr + 1][0][lx] = make_float4(bottom[0], bottom[1], bottom[2], bottom[3]);
        *(float4 *)&edge[buf][2 * tr + 1][1][lx] = make_float4(bottom[4], bottom[5], bottom[6], bottom[7]);
        if (tr == ntr - 1) {
            *(float4 *)&edge[buf][2 * NTR][0][lx] = make_float4(bottom[1], bottom[0], bottom[3], bottom[2]);
            *(float4 *)&edge[buf][2 * NTR][1][lx] = make_float4(bottom[5], bottom[4], bottom[7], bottom[6]);
        }
        if (tr == 0) {
            *(float4 *)&edge[buf][2 * NTR + 1][0][lx] = make_float4(top[1], top[0], top[3], top[2]);
            *(float4 *)&edge[buf][2 * NTR + 1][1][lx] = make_float4(top[5], top[4], top[7], top[6]);
        }
#ifdef RTDD_PK_RELAXED_PUBLISH     // (the LDS serves one wave's accesses in order: compiler barriers only, as the launch-per-block k_sweep_blocked)
        asm volatile("" ::: "memory");
        __hip_atomic_store(&published[wv], sweep_no + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
#else
        __hip_atomic_store(&published[wv], sweep_no + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
    };
    unsigned lds_spins = 0;
    bool gone = false;
    auto await = [&](int sweep_no) {
        if (gone) return;
        const int idx = (tid & 63) == 0 ? (wv > 0 ? wv - 1 : nwv - 1) : (wv < nwv - 1 ? wv + 1 : 0);
        auto there = [&]() { return __builtin_amdgcn_ballot_w64(__hip_atomic_load(&published[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < sweep_no + 1) == 0; };
        if (__builtin_expect(!there(), 0)) {
            for (;;) {
                __builtin_amdgcn_s_sleep(1);
                if (there()) break;
                if ((++lds_spins & 1023u) == 0) {
                    if (__hip_atomic_load(&dead_s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) { gone = true; break; }
                    if (lds_spins > (1u << 22)) {
                        __hip_atomic_store(&sync_words[kSyncStatus], 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(&dead_s, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        gone = true; break;
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    const f2 zero2 = f2{0.0f, 0.0f};
    auto sweep = [&](f8 (&cur)[G], f8 (&oth)[G], int s, auto fast, bool last_of_block, auto parity) {
        constexpr bool FAST = decltype(fast)::value;
        constexpr int buf = decltype(parity)::value;         // = s & 1
        RTDD_TL(0, s);
#if defined(RTDD_DIAG_NOLDS)        // (diagnostic ablation, timing only: no wait for the neighbours, no LDS reads)
        const f8 up = cur[0], dn = cur[G - 1];
#else
#ifndef RTDD_DIAG_NOPOLL           // (diagnostic ablation, timing only: the LDS reads without the wait for the neighbours' counters)
        await(s);
#endif
        const float4 u0 = edge[buf][slot_up][0][lx], u1 = edge[buf][slot_up][1][lx];
        const float4 d0 = edge[buf][slot_dn][0][lx], d1 = edge[buf][slot_dn][1][lx];
        const f8 up = f8{u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w}, dn = f8{d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#endif
#ifdef RTDD_TIMELINE
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        RTDD_TL(1, s);
#endif
        const float omega = omegas[s];
        const f2 om2 = f2{omega, omega}, ga2 = f2{gamma, gamma};
        f2 xl0[G], xr3[G];
        // weighted sums of pair (g, i): solveDiffusion, src/GPUSolver.cu:73-106, absent neighbours carried as (w = 0, x = 0)
        auto wsum = [&](int g, int i) {
            const f2 xr = i == 3 ? xr3[g] : pr(cur[g], i + 1);
            const f2 xu = g == 0 ? pr(up, i) : pr(cur[g - 1], i);
            const f2 xd = g == G - 1 ? pr(dn, i) : pr(cur[g + 1], i);
            const f2 wu = g == 0 ? pr(wu0, i) : pr(wd[g - 1], i);
            // xl0[g] holds the previous lane's wr[g][3] * cur[g][3]; RN(wl * xl) + 0 is what fma(wl, xl, 0) and 0 + wl * xl both give
            f2 sum = i == 0 ? zero2 + xl0[g] : (CONTRACT ? fma2(pr(wr[g], i - 1), pr(cur[g], i - 1), zero2) : zero2 + pr(wr[g], i - 1) * pr(cur[g], i - 1));
            sum = CONTRACT ? fma2(pr(wr[g], i), xr, sum) : sum + pr(wr[g], i) * xr;
            sum = CONTRACT ? fma2(wu, xu, sum) : sum + wu * xu;
            sum = CONTRACT ? fma2(pr(wd[g], i), xd, sum) : sum + pr(wd[g], i) * xd;
            return sum;
        };
        // x_{k+1} of pair (g, i) from its quotient: clamp (:104; v_med3_f32 returns min3 when an operand is NaN, i.e. 0 here, like fmax /
        // fmin), then src/GPUSolver.cu:259
        auto update = [&](f2 qv, int g, int i) {
            const f2 r = f2{__builtin_amdgcn_fmed3f(qv.x, 0.0f, 255.0f), __builtin_amdgcn_fmed3f(qv.y, 0.0f, 255.0f)};
            const f2 x = pr(cur[g], i), prev = pr(oth[g], i);
            return CONTRACT ? fma2(om2, fma2(ga2, r - x, x) - prev, prev) : (om2 * (ga2 * (r - x) + x - prev)) + prev;
        };
        // Rows g of the group selected by `pick`.  The waves issue in order, and the test for tiny numerators ends in a serial chain
        // (numerator -> shift -> four dependent v_min3 -> compare -> branch): whatever the compiler puts behind that branch waits for it.
        // Left alone it SINKS the divides and the update there (the rare path recomputes them).  So the whole fast path -- sums,
        // 3-operation divides, clamp, update -- is computed into v[] BEFORE the branch and pinned there (the empty asm), the rare path
        // overwrites v[], and only the commit (the select of Dirichlet pixels into x_{k-1}'s registers) comes after it.
        // The fast path is written STAGE by stage across the four pairs of a row with a scheduling barrier between stages: at the
        // register limit the scheduler otherwise finishes one pair before it starts the next -- a chain of ~16 dependent packed
        // instructions, each waiting out the latency of the one before (measured 5.4 cycles per instruction against 4.3 for the same
        // mix with independent neighbours; three waves per SIMD do not cover that).
#ifdef RTDD_PK_NO_STAGES
#define RTDD_SB() do {} while (0)
#else
#define RTDD_SB() __builtin_amdgcn_sched_barrier(0)
#endif
        auto acc = [&](f2 w, f2 x, f2 s) { return CONTRACT ? fma2(w, x, s) : s + w * x; };
        auto group = [&](auto pick) {
            f2 v[G][4];
            uint32_t tmin = 0xFFFFFFFFu;
#pragma unroll
            for (int g = 0; g < G; g++) {
                if (!pick(g)) continue;
                if (FAST) {
                    const f2 t3 = pr(wr[g], 3) * pr(cur[g], 3);
                    const f2 xr3g = f2{lane_from_next(cur[g][0]), lane_from_next(cur[g][1])};
                    f2 sum[4], q0[4], rem[4], r[4], d[4];
                    uint32_t t[8];
#pragma unroll
                    for (int i = 1; i < 4; i++) sum[i] = acc(pr(wr[g], i - 1), pr(cur[g], i - 1), zero2);
                    RTDD_SB();
                    const f2 xl0g = f2{lane_from_prev(t3.x), lane_from_prev(t3.y)};
                    sum[0] = zero2 + xl0g;           // RN(wl * xl) + 0 is what fma(wl, xl, 0) and 0 + wl * xl both give
                    RTDD_SB();
#pragma unroll
                    for (int i = 0; i < 4; i++) sum[i] = acc(pr(wr[g], i), i == 3 ? xr3g : pr(cur[g], i + 1), sum[i]);
                    RTDD_SB();
#pragma unroll
                    for (int i = 0; i < 4; i++) sum[i] = acc(g == 0 ? pr(wu0, i) : pr(wd[g - 1], i), g == 0 ? pr(up, i) : pr(cur[g - 1], i), sum[i]);
                    RTDD_SB();
#pragma unroll
                    for (int i = 0; i < 4; i++) sum[i] = acc(pr(wd[g], i), g == G - 1 ? pr(dn, i) : pr(cur[g + 1], i), sum[i]);
                    RTDD_SB();
#pragma unroll
                    for (int i = 0; i < 4; i++) {            // the 3-operation divide (sweep_common.hpp div_tail), both halves; the tiny test's shifts beside it
                        q0[i] = sum[i] * pr(rcp[g], i);
                        t[2 * i] = (__float_as_uint(sum[i].x) << 1) + 0xFFFFFFFFu; t[2 * i + 1] = (__float_as_uint(sum[i].y) << 1) + 0xFFFFFFFFu;
                    }
                    RTDD_SB();
#pragma unroll
                    for (int i = 0; i < 4; i++) rem[i] = fma2(-pr(cnt[g], i), q0[i], sum[i]);
                    tmin = min(min(tmin, t[0]), t[1]);
                    RTDD_SB();
#pragma unroll
                    for (int i = 0; i < 4; i++) q0[i] = fma2(rem[i], pr(rcp[g], i), q0[i]);
                    tmin = min(min(tmin, t[2]), t[3]);
                    RTDD_SB();
#pragma unroll
                    for (int i = 0; i < 4; i++) r[i] = f2{__builtin_amdgcn_fmed3f(q0[i].x, 0.0f, 255.0f), __builtin_amdgcn_fmed3f(q0[i].y, 0.0f, 255.0f)};   // :104
                    tmin = min(min(tmin, t[4]), t[5]);
                    RTDD_SB();
                    if (CONTRACT) {                          // src/GPUSolver.cu:259
#pragma unroll
                        for (int i = 0; i < 4; i++) d[i] = r[i] - pr(cur[g], i);
                        tmin = min(min(tmin, t[6]), t[7]);
                        RTDD_SB();
#pragma unroll
                        for (int i = 0; i < 4; i++) d[i] = fma2(ga2, d[i], pr(cur[g], i));
                        RTDD_SB();
#pragma unroll
                        for (int i = 0; i < 4; i++) d[i] = d[i] - pr(oth[g], i);
                        RTDD_SB();
#pragma unroll
                        for (int i = 0; i < 4; i++) v[g][i] = fma2(om2, d[i], pr(oth[g], i));
                    } else {
                        tmin = min(min(tmin, t[6]), t[7]);
#pragma unroll
                        for (int i = 0; i < 4; i++) v[g][i] = (om2 * (ga2 * (r[i] - pr(cur[g], i)) + pr(cur[g], i) - pr(oth[g], i))) + pr(oth[g], i);
                    }
                    RTDD_SB();
#ifndef RTDD_PK_NO_PIN
                    asm volatile("" : "+v"(v[g][0]), "+v"(v[g][1]), "+v"(v[g][2]), "+v"(v[g][3]));
#endif
                } else {
                    const f2 t3 = pr(wr[g], 3) * pr(cur[g], 3);
                    xl0[g] = f2{lane_from_prev(t3.x), lane_from_prev(t3.y)};
                    xr3[g] = f2{lane_from_next(cur[g][0]), lane_from_next(cur[g][1])};
#pragma unroll
                    for (int i = 0; i < 4; i++) { const f2 sum = wsum(g, i); v[g][i] = update(f2{sum.x / cnt[g][2 * i], sum.y / cnt[g][2 * i + 1]}, g, i); }
                }
            }
#ifdef RTDD_DIAG_NOTINY          // (diagnostic ablation: timing only)
            tmin = 0xFFFFFFFFu;
#endif
            if (FAST && __builtin_expect(__builtin_amdgcn_ballot_w64(tmin < kTinyT) != 0, 0)) {      // wave-uniform, rare: numerators below 2^-100
#pragma unroll
                for (int g = 0; g < G; g++) {
                    if (!pick(g)) continue;
                    const f2 t3 = pr(wr[g], 3) * pr(cur[g], 3);
                    xl0[g] = f2{lane_from_prev(t3.x), lane_from_prev(t3.y)};
                    xr3[g] = f2{lane_from_next(cur[g][0]), lane_from_next(cur[g][1])};
#pragma unroll
                    for (int i = 0; i < 4; i++) { const f2 sum = wsum(g, i); v[g][i] = update(f2{sum.x / cnt[g][2 * i], sum.y / cnt[g][2 * i + 1]}, g, i); }
                }
            }
#pragma unroll
            for (int g = 0; g < G; g++) {
                if (!pick(g)) continue;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    oth[g][2 * i] = (dirichlet >> ((g * 4 + i) * 2)) & 1u ? cur[g][2 * i] : v[g][i].x;       // x_{k+1} replaces x_{k-1}
                    oth[g][2 * i + 1] = (dirichlet >> ((g * 4 + i) * 2 + 1)) & 1u ? cur[g][2 * i + 1] : v[g][i].y;
                }
            }
        };
#ifdef RTDD_PK_MERGE_GROUPS        // first and last row in ONE group: eight independent chains, one tiny test
        group([](int g) { return g == 0 || g == G - 1; });
#else
        group([](int g) { return g == 0; });
        if (G > 1) group([](int g) { return g == G - 1; });
#endif
        RTDD_TL(2, s);
        if (!last_of_block) publish(oth[0], oth[G - 1], s + 1, buf ^ 1);
#pragma unroll
        for (int gi = 1; gi < G - 1; gi++) group([gi](int g) { return g == gi; });
        RTDD_TL(3, s);
    };

    const int tile_id = by * gx + bx;
    int s = 0, blk = 0;
    bool odd = false;
    for (;; blk++) {
        const int s_end = min(s + block_sweeps, nsweeps);
        publish(a[0], a[G - 1], s, 0);               // block prologue (a = newest here; s is even)
        using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, 1>;
        if (!wave_unsafe) {
            for (; s + 1 < s_end; s += 2) {
                sweep(a, b, s, std::true_type{}, false, P0{});
                sweep(b, a, s + 1, std::true_type{}, s + 2 >= s_end, P1{});
            }
            if (s < s_end) { sweep(a, b, s, std::true_type{}, true, P0{}); s++; odd = true; }
        } else {
            for (; s + 1 < s_end; s += 2) {
                sweep(a, b, s, std::false_type{}, false, P0{});
                sweep(b, a, s + 1, std::false_type{}, s + 2 >= s_end, P1{});
            }
            if (s < s_end) { sweep(a, b, s, std::false_type{}, true, P0{}); s++; odd = true; }
        }
        if (!PERSIST || s >= nsweeps) break;

        // ---- persistent mode: refresh the halo from the neighbouring tiles (protocol: persist_sync.hpp; buffers alternate by block parity) ----
        {
            float *Ek = (blk & 1) ? Xk : Yk, *Em = (blk & 1) ? Xm : Ym;
            int tid_x = tid;
            asm volatile("" : "+v"(tid_x));          // (keeps this address arithmetic out of the sweep loops: sweep_blocked.hip)
            const int lx = tid_x % LX, tr = tid_x / LX;
            const int x0 = bx * TW - hx + 4 * lx, y0 = by * TH - hy + tr * G;
            const bool colok = x0 >= 0 && x0 < cols;
            const bool xin = colok && 4 * lx >= hx && 4 * lx < EW - hx;
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int g = 0; g < G; g++) {
                    const int ty = h * sh + tr * G + g, y = y0 + h * sh + g;
                    const bool central = xin && ty >= hy && ty < eh - hy && y < rows;
                    const bool band = ty < 2 * hy || ty >= eh - 2 * hy || 4 * lx < 2 * hx || 4 * lx >= EW - 2 * hx;
                    if (central && band) {
                        const size_t off = (size_t)y * ip + x0;
                        store_sc1((float4 *)(Ek + off), make_float4(a[g][h], a[g][2 + h], a[g][4 + h], a[g][6 + h]));
                        store_sc1((float4 *)(Em + off), make_float4(b[g][h], b[g][2 + h], b[g][4 + h], b[g][6 + h]));
                    }
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every storing wave drains its write-through stores
            __syncthreads();
            if (exchange_wait<false>(sync_words, &dead_s, &seen_s, tid, tile_id, bx, by, gx, gy, flag_base + blk + 1)) return;
            // 16-byte sc1 loads straight into registers, no acquire (sweep_blocked.hip); all loads issued, then ONE wait the loaded registers pass through
            f4v_t hk[2][G], hm[2][G];
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int g = 0; g < G; g++) {
                    const int ty = h * sh + tr * G + g, y = y0 + h * sh + g;
                    const bool central = xin && ty >= hy && ty < eh - hy;
                    const bool ok = colok && y >= 0 && y < rows;
                    hk[h][g] = f4v_t{0, 0, 0, 0}; hm[h][g] = hk[h][g];
                    if (ok && !central) {                                    // a halo pixel inside the image: some neighbour's centre
                        const size_t off = (size_t)y * ip + x0;
                        load_sc1(hk[h][g], Ek + off); load_sc1(hm[h][g], Em + off);
                    }
                }
            if constexpr (G == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(hk[0][0]), "+v"(hm[0][0]), "+v"(hk[1][0]), "+v"(hm[1][0]) :: "memory");
            else if constexpr (G == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(hk[0][0]), "+v"(hm[0][0]), "+v"(hk[1][0]), "+v"(hm[1][0]), "+v"(hk[0][1]), "+v"(hm[0][1]), "+v"(hk[1][1]), "+v"(hm[1][1]) :: "memory");
            else {
                static_assert(G <= 3, "add a wait for this G");
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(hk[0][0]), "+v"(hm[0][0]), "+v"(hk[1][0]), "+v"(hm[1][0]), "+v"(hk[0][1]), "+v"(hm[0][1]), "+v"(hk[1][1]), "+v"(hm[1][1]),
                                                    "+v"(hk[0][2]), "+v"(hm[0][2]), "+v"(hk[1][2]), "+v"(hm[1][2]) :: "memory");
            }
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int g = 0; g < G; g++) {
                    const int ty = h * sh + tr * G + g, y = y0 + h * sh + g;
                    const bool central = xin && ty >= hy && ty < eh - hy;
                    const bool ok = colok && y >= 0 && y < rows;
                    if (ok && !central) {
#pragma unroll
                        for (int i = 0; i < 4; i++) { const bool in = x0 + i < cols; a[g][2 * i + h] = in ? hk[h][g][i] : 0.0f; b[g][2 * i + h] = in ? hm[h][g][i] : 0.0f; }
                    }
                }
        }
    }
    // results of the last block go to the exchange buffer of ITS parity (blk = 0 -> Yk/Ym)
    if (PERSIST && (blk & 1)) { Yk = Xk; Ym = Xm; }

    // ---- write back the part that is still exact ----
    int tid_w = tid;
    asm volatile("" : "+v"(tid_w));
    const int lx_w = tid_w % LX, tr_w = tid_w / LX;
    const int x0_w = bx * TW - hx + 4 * lx_w, y0_w = by * TH - hy + tr_w * G;
    const bool xin_w = x0_w >= 0 && x0_w < cols && 4 * lx_w >= hx && 4 * lx_w < EW - hx;
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int ty = h * sh + tr_w * G + g, y = y0_w + h * sh + g;
            if (xin_w && ty >= hy && ty < eh - hy && y < rows) {
                const size_t off = (size_t)y * ip + x0_w;
                // newest iterate -> Yk, the one before it -> Ym
                const float4 va = make_float4(a[g][h], a[g][2 + h], a[g][4 + h], a[g][6 + h]), vb = make_float4(b[g][h], b[g][2 + h], b[g][4 + h], b[g][6 + h]);
                const float4 vk = make_float4(odd ? vb.x : va.x, odd ? vb.y : va.y, odd ? vb.z : va.z, odd ? vb.w : va.w);
                const float4 vm = make_float4(odd ? va.x : vb.x, odd ? va.y : vb.y, odd ? va.z : vb.z, odd ? va.w : vb.w);
                store_result((float4 *)(Yk + off), vk);
                store_result((float4 *)(Ym + off), vm);
            }
        }
}

// ---- host side: the packed tiles of sweep_blocked.hip's table ----------------------------------------------------------------
#define RTDD_PK_TILES RTDD_PK_CASE(17, 32, 768, 2, 3) RTDD_PK_CASE(18, 32, 512, 3, 2) RTDD_PK_CASE(19, 16, 384, 2, 3)

bool pk_persistent_possible(rtdd_ctx *ctx, int tile, int nthreads) {
    signed char &c = ctx->persist_fit[tile][ctx->opt.fp_contract ? 1 : 0];
    if (c < 0) {
        int nb = 0;
        hipError_t e = hipErrorInvalidValue;
        switch (tile) {
#define RTDD_PK_CASE(id, LX_, NT_, G_, W_) case id: e = ctx->opt.fp_contract ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_sweep_pk<LX_, NT_, G_, W_, true, true>, nthreads, 0) \
                                                                          : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_sweep_pk<LX_, NT_, G_, W_, false, true>, nthreads, 0); break;
            RTDD_PK_TILES
#undef RTDD_PK_CASE
            default: break;
        }
        c = (e == hipSuccess && nb >= 1) ? 1 : 0;
    }
    return c == 1;
}

void launch_sweep_pk(rtdd_ctx *ctx, int tile, dim3 grid, int xcd_tiles, int nthreads, float *Xk, float *Xm, float *Yk, float *Ym, const uint32_t *M,
                     const float *omegas, int ip, int rows, int cols, int hx, int hy, int n, float gamma, int block_sweeps, int flag_base) {
    const bool persist = block_sweeps < n;
    const dim3 launch_grid = xcd_tiles > 0 ? dim3(8 * xcd_tiles) : grid;
#define RTDD_LAUNCH(LX_, NT_, G_, W_, C, P) hipLaunchKernelGGL((k_sweep_pk<LX_, NT_, G_, W_, C, P>), launch_grid, dim3(nthreads), 0, ctx->stream, Xk, Xm, Yk, Ym, M, ctx->lut_dev, omegas, ip, rows, cols, hx, hy, n, gamma, block_sweeps, ctx->sync_words, (int)grid.x, (int)grid.y, xcd_tiles, flag_base)
    switch (tile) {
#define RTDD_PK_CASE(id, LX_, NT_, G_, W_) case id: \
        if (ctx->opt.fp_contract) { if (persist) RTDD_LAUNCH(LX_, NT_, G_, W_, true, true); else RTDD_LAUNCH(LX_, NT_, G_, W_, true, false); } \
        else { if (persist) RTDD_LAUNCH(LX_, NT_, G_, W_, false, true); else RTDD_LAUNCH(LX_, NT_, G_, W_, false, false); } \
        break;
        RTDD_PK_TILES
#undef RTDD_PK_CASE
        default: break;
    }
#undef RTDD_LAUNCH
}

}  // namespace rtdd
