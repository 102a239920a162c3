// sweep_blocked.hip -- temporally blocked Chebyshev-Jacobi sweeps: n sweeps per launch.
//
// Why: one sweep per launch moves 16 B/pixel through HBM/L2 and does ~25 flops on it, so the
// one-sweep kernel (solver_kernels.hip) can never beat bytes/bandwidth, and at 1080p its 7 us
// of work is the same order as the launch boundary.  Jacobi sweeps are order-independent, so
// a workgroup can load a tile plus a halo of n pixels ONCE, run n sweeps on it entirely in
// registers, and write back the part of the tile that is still exact (everything further than
// n pixels from a tile edge that is not an image border).  Values are bit-identical to n
// separate sweeps -- only the schedule changes.
//
// Layout of one workgroup (template LX lanes per tile row, NT threads, G rows per thread):
//   * the extended tile is EW = 4*LX pixels wide and EH = (NT/LX)*G rows tall;
//   * a thread owns a 4-pixel-wide, G-row-tall block: x_k, x_{k-1}, its right/down weights (as
//     f32, gathered from the LDS copy of the LUT once per launch), and 1/... nothing else;
//   * horizontal neighbours come from the adjacent lane by DPP wave shifts (no LDS);
//   * vertical neighbours inside the block are the thread's own registers; only the block's top
//     and bottom rows go through LDS (two ds_write_b128 + two ds_read_b128 per thread per sweep,
//     double buffered so one barrier per sweep suffices);
//   * x_{k+1} overwrites x_{k-1}'s registers (each pixel reads only its own x_{k-1}), and the
//     sweep loop is unrolled by two so the role swap costs no moves.
// Outside-image pixels are held at x = 0 with all weights 0, which reproduces the reference's
// "skip the missing neighbour" (src/GPUSolver.cu:79-101) exactly: fma(0, 0, s) == s up to the
// sign of a zero sum, and that sign never reaches the output ("Zero-weight borders").
#include <cstdlib>
#include <type_traits>

#include "rtdd_internal.hpp"
#include <cstring>
#include "persist_sync.hpp"
#include "sweep_common.hpp"
#include "sweep_diag.hpp"      // RTDD_STAMP / RTDD_TL / RTDD_XT: empty unless a diagnostic micro-benchmark asks for them

// RTDD_MASKED_UPDATE (default 1): the update's last fma under an EXEC mask instead of a v_cndmask behind it, omega / gamma in VGPRs
// (sweep_common.hpp masked_fmac4); 0 = the round-4 form, kept for A/B builds (scripts/build_variant.sh).
#ifndef RTDD_MASKED_UPDATE
#define RTDD_MASKED_UPDATE 1
#endif

// RTDD_OCC_G2 (A/B builds; default 4 = no change): waves per SIMD asked of the compiler for the 64 x 64 tile of 8 pixels per thread (tile 10):
// 6 would put three of its workgroups on a CU (85 registers per thread) -- round 6's question whether more workgroups per CU hide the
// tile loads at 4K better than the 64 x 96 tile's two (EXPERIMENTS.md)
#ifndef RTDD_OCC_G2
#define RTDD_OCC_G2 4
#endif

namespace rtdd {


template <bool CONTRACT, bool FAST>
__device__ __forceinline__ float relax(float xl, float xr, float xu, float xd, float wl, float wr, float wu, float wd,
                                       float cnt, float rcp, float x, float prev, float omega, float gamma) {
    // solveDiffusion, src/GPUSolver.cu:73-106, with absent neighbours carried as (w = 0, x = 0)
    float sum = 0.0f;
    sum = CONTRACT ? __builtin_fmaf(wl, xl, sum) : sum + wl * xl;
    sum = CONTRACT ? __builtin_fmaf(wr, xr, sum) : sum + wr * xr;
    sum = CONTRACT ? __builtin_fmaf(wu, xu, sum) : sum + wu * xu;
    sum = CONTRACT ? __builtin_fmaf(wd, xd, sum) : sum + wd * xd;
    float r;                                   // cnt == 0 was replaced by 1 (sum is 0 there): r = 0 (:103)
    if (FAST) {
        r = div_tail(sum, cnt, rcp);
        const bool tiny = __builtin_fabsf(sum) < 0x1p-100f && sum != 0.0f;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(tiny) != 0, 0)) r = tiny ? sum / cnt : r;   // wave-uniform, rare
    } else {
        r = sum / cnt;
    }
    // :104  min(max(r,0),255).  v_med3_f32 returns min3 when an operand is NaN, i.e. 0 here -- the same as fmax/fmin --
    // and, unlike fminf(fmaxf()), needs no canonicalising v_max in front of it (one VALU op per pixel less).
    r = __builtin_amdgcn_fmed3f(r, 0.0f, 255.0f);
    // src/GPUSolver.cu:259
    if (CONTRACT) return __builtin_fmaf(omega, __builtin_fmaf(gamma, r - x, x) - prev, prev);
    return (omega * (gamma * (r - x) + x - prev)) + prev;
}

// Register budget: a G = 4 thread holds 6 x 16 + 8 live values; capping it at 128 VGPRs spills into the
// sweep loop (measured 25 % slower), so G = 4 tiles ask for 3 waves/SIMD (168 VGPRs) unless the
// workgroup is 1024 threads (which needs 4 waves/SIMD to be launchable at all).  G <= 3 fits 128.
template <int LX, int NT, int G, bool CONTRACT, bool PERSIST>
__global__ __launch_bounds__(NT, (G == 2 && NT == 512 ? RTDD_OCC_G2 : NT >= 1024 || G <= 3 ? 4 : G <= 4 ? 3 : 2)) void k_sweep_blocked(float *Xk, float *Xm, float *Yk, float *Ym,
                                                      const uint32_t *__restrict__ M, const float *__restrict__ lut_g,
                                                      const float *__restrict__ omegas, int ip, int rows, int cols,
                                                      int hx, int hy, int nsweeps, float gamma,
                                                      int block_sweeps, int *sync_words, int gx, int gy, int xcd_tiles, int flag_base, size_t zPlane) {
    // block_sweeps == nsweeps: the plain time-blocked launch (results -> Yk/Ym).
    // block_sweeps <  nsweeps: PERSISTENT mode -- the workgroup keeps its tile in registers for the whole solve and,
    // every block_sweeps (= halo width, even) sweeps, trades halo strips with its 8 neighbours through memory instead
    // of ending the kernel.  Needs every workgroup co-resident (the host only uses it when grid <= #CUs).
    constexpr int EW = 4 * LX, NTR = NT / LX;
    __shared__ float lut[257];
    __shared__ float4 edge[2][NTR][2][LX];     // [buffer][thread row][0 = its top row, 1 = its bottom row][lane]
    __shared__ int published[NT / 64 + 1];     // per wave: number of sweeps whose edge rows it has published; [NT/64]: the maximum over the waves
    __shared__ int dead_s;                     // the launch has failed (persist_sync.hpp): leave

    // (gx, gy) = the grid of tiles.  xcd_tiles > 0 (every multi-tile launch): a 1-D launch of 8 * xcd_tiles workgroups in which
    // workgroup p -- dispatched to XCD p % 8 -- takes tile number (p % 8) * xcd_tiles + p / 8, so that each XCD owns a run
    // of consecutive tile numbers and most halo strips are traded inside one L2 instead of through memory (+3.5 % at 1080p);
    // surplus workgroups leave.
    int bx = blockIdx.x, by = blockIdx.y;
    if (xcd_tiles > 0) {
        const int t = ((int)blockIdx.x & 7) * xcd_tiles + ((int)blockIdx.x >> 3);
        if (t >= gx * gy) return;
        bx = t % gx; by = t / gx;           // (numbering the tiles in compact 8x4 patches instead of row bands measured the same)
    }
    // batched launch (rtdd_estimate_depth_batch): blockIdx.z = image, its planes zPlane bytes behind the previous image's, its tiles' flags
    // behind the previous image's tiles'
    RTDD_Z(Xk, zPlane); RTDD_Z(Xm, zPlane); RTDD_Z(Yk, zPlane); RTDD_Z(Ym, zPlane); RTDD_Z(M, zPlane);
    RTDD_STAMP(0);
    // Tile load and setup at a raised wave priority: where two workgroups share a CU (4K, 8K) the one that has just arrived gets through
    // its loads, table gathers and reciprocals ahead of the other one's sweeps and joins them sooner (4K +2.7 %, 8K +2.3 %; levels 1-3:
    // the same; the write-back at a raised priority too: no more).
    __builtin_amdgcn_s_setprio(2);
    const int tid = threadIdx.x;
    const int lx = tid % LX, tr = tid / LX;
    const int ntr = (int)blockDim.x / LX;          // thread rows actually launched (blockDim.x <= NT: small levels launch only the thread rows they need)
    const int eh = ntr * G;                        // rows of the extended tile actually covered
    const int TW = EW - 2 * hx, TH = eh - 2 * hy;
    const int x0 = bx * TW - hx + 4 * lx;
    const int y0 = by * TH - hy + tr * G;
    const bool colok = x0 >= 0 && x0 < cols;

    // ext_vector_type keeps each 4-pixel group in 4 consecutive VGPRs, so the 16-byte LDS / global accesses need no moves
    typedef float f4r __attribute__((ext_vector_type(4)));
    f4r a[G], b[G];                            // a = x_k, b = x_{k-1}; roles alternate every sweep
    float wr[G][4], wd[G][4], wl0[G], wu0[4], cnt[G][4], rcp[G][4];
    bool unsafe = false;                       // some pixel of this lane has a denormal divisor
    uint32_t dirichlet = 0;                    // bit g*4+i

    // ---- load the extended tile once: every load is issued BEFORE the weight table is staged in LDS and the barrier behind it, so
    // that a launch pays one memory round trip, not two in a row (launch-per-block: 125 launches per 1000 sweeps at 4K) ------------
    float4 vxr[G], vpr[G];
    uint4 mr[G], mup = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int g = 0; g < G; g++) {
        const int y = y0 + g;
        vxr[g] = make_float4(0, 0, 0, 0); vpr[g] = vxr[g]; mr[g] = make_uint4(0, 0, 0, 0);
        if (colok && y >= 0 && y < rows) {
            const size_t off = (size_t)y * ip + x0;
            vxr[g] = *(const float4 *)(Xk + off);
            vpr[g] = *(const float4 *)(Xm + off);
            mr[g] = *(const uint4 *)(M + off);
        }
    }
    if (colok && y0 - 1 >= 0 && y0 < rows) mup = *(const uint4 *)(M + (size_t)(y0 - 1) * ip + x0);     // the row above the block: its down-weights
    for (int i = tid; i < 257; i += (int)blockDim.x) lut[i] = lut_g[i];
    if (tid <= NT / 64) published[tid] = 0;
    if (tid == 0) { dead_s = PERSIST && __hip_atomic_load(&sync_words[kSyncStatus], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0; }
    const int ntiles = gx * gy, tile_base = (int)blockIdx.z * ntiles, tile_id = by * gx + bx;
    if (PERSIST) {
        // Arrival (persist_sync.hpp kArrivalPollLimit): the tile's flag := the launch's base value, then the 8 neighbours' -- while the
        // tile's loads are in flight.  A workgroup that is not resident never announces itself, its neighbours give up within a
        // millisecond, the status word is set and every workgroup leaves here, before any sweep (also at once when an earlier
        // persistent launch of this context has timed out).  Its barrier is also the one the table's staging needs.
        if (exchange_wait<false, true>(sync_words, &dead_s, tid, tile_id, bx, by, gx, gy, flag_base, tile_base)) return;
    } else {
        __syncthreads();
    }

    RTDD_STAMP_LOADED(4);
    // wave-uniform (scalar: the tile's coordinates): the whole extended tile, every pixel's right and lower neighbour and the row above the
    // tile are inside the image -- the setup then needs none of its border selects (sweep_tile_setup.inc)
    const int tx0 = bx * TW - hx, ty0 = by * TH - hy;
    const bool tile_inside = tx0 >= 0 && tx0 + EW < cols && ty0 >= 1 && ty0 + eh < rows;
#include "sweep_tile_setup.inc"       // vxr / vpr / mr / mup -> a, b, weights, divisors, reciprocals

    RTDD_STAMP(1);
    __builtin_amdgcn_s_setprio(0);
    // ---- n sweeps in registers -------------------------------------------------------------------
    // One sweep, written for instruction-level parallelism: the weighted sums and quotients of a GROUP of rows first (12 independent
    // 7-deep chains the scheduler can interleave -- a wave alone on its SIMD issues a dependent VALU instruction only every ~6.6 cycles,
    // an independent one every 4), ONE wave-uniform test for numerators too small for the 3-op divide (below), then the updates.
    //   * "tiny" test: the 3-op divide needs sum == 0 or |sum| >= 2^-100.  Per pixel t = 2 * bits(|sum|) - 1 (one v_lshl_add_u32:
    //     the shift drops the sign, the wrap-around sends an exact zero to 0xFFFFFFFF), a running v_min3_u32 over the group, one
    //     compare: 1.5 VALU ops per pixel and no branch, instead of two compares, two SALU ops and a branch per pixel.  A wave that
    //     does see such a numerator (values decaying through 1e-31, pixels whose four weights are all ~1e-38) recomputes the group
    //     with the full IEEE divide; x_{k-1} is still intact then because the updates come after the test.
    //   * the rows the neighbouring thread rows need (a thread's first and last) are computed first and PUBLISHED as soon as they
    //     are updated, before the interior rows; the wait for the neighbours' rows at the top of the next sweep then usually finds
    //     them there.  Buffer reuse: I write buffer (s+1)&1 in sweep s after my wait for the neighbours' counters >= s+1, which they
    //     set after consuming (reading and waiting for) my sweep-(s-1) rows from that buffer.
    //   * one row per group: 4 chains cover the VALU latency, and only 4 quotients + 4 sums are live at a time (the 1024-thread tiles
    //     have 128 registers; two rows per group spilled inside the loop).
    // Measured and dropped: publishing without draining lgkmcnt, fetching the neighbours' rows speculatively
    // before their counter is known, fetching counter and rows in one go, s_setprio feedback for lagging waves, a half-sweep
    // stagger between groups of four waves, a wave-level (barrier-free) hand-off between workgroups, and a "zebra" row order (even
    // waves top-down, odd waves bottom-up, every edge row published as soon as it is computed and fetched a step ahead).
#include "sweep_tile_sweeps.inc"      // publish / await / sweep lambdas
    // (the divide variant is chosen per wave; the neighbour handshake above does not care which one a wave runs)
    int s = 0, blk = 0;
    bool odd = false;
    RTDD_XT_BEGIN;
    for (;; blk++) {
        const int s_end = min(s + block_sweeps, nsweeps);
        publish(a[0], a[G - 1], s, 0);               // block prologue: the rows the first sweep of this block reads (a = newest here; s is even)
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
        RTDD_XT(0);

        // ---- persistent mode: refresh the halo from the neighbours (block_sweeps is even here, so a = newest) ----
        // Protocol: persist_sync.hpp (write-through payload stores, every storing wave drains, barrier, then exchange_wait()).
        // Exchange buffers alternate between (Yk,Ym) and (Xk,Xm) by block parity: a neighbour publishes block b+1 only
        // after consuming my block-b strips, and I overwrite that buffer (block b+2) only after waiting for its b+1.
        {
            float *Ek = (blk & 1) ? Xk : Yk, *Em = (blk & 1) ? Xm : Ym;
            // The exchange's addresses and predicates are recomputed here from "laundered" copies of the thread coordinates: left
            // to itself the compiler hoists them (six 64-bit offsets, a dozen lane masks) out of the block loop, where they stay
            // live across the sweeps and push the sweep loop's operands into scratch.
            int tid_x = tid;
            asm volatile("" : "+v"(tid_x));
            const int lx = tid_x % LX, tr = tid_x / LX;
            const int x0 = bx * TW - hx + 4 * lx, y0 = by * TH - hy + tr * G;
            const bool colok = x0 >= 0 && x0 < cols;
            const bool xin = colok && 4 * lx >= hx && 4 * lx < EW - hx;
#pragma unroll
            for (int g = 0; g < G; g++) {
                const int y = y0 + g, ty = tr * G + g;
                const bool central = xin && ty >= hy && ty < eh - hy && y < rows;
                const bool band = ty < 2 * hy || ty >= eh - 2 * hy || 4 * lx < 2 * hx || 4 * lx >= EW - 2 * hx;
                if (central && band) {
                    const size_t off = (size_t)y * ip + x0;
                    store_sc1((float4 *)(Ek + off), make_float4(a[g][0], a[g][1], a[g][2], a[g][3]));
                    store_sc1((float4 *)(Em + off), make_float4(b[g][0], b[g][1], b[g][2], b[g][3]));
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every storing wave drains its write-through stores
            __syncthreads();
            RTDD_XT(1);
#ifndef RTDD_EXCHANGE_ACQUIRE
#define RTDD_EXCHANGE_ACQUIRE 0
#endif
            if (exchange_wait<RTDD_EXCHANGE_ACQUIRE != 0>(sync_words, &dead_s, tid, tile_id, bx, by, gx, gy, flag_base + blk + 1, tile_base)) return;      // flag, bounded poll, (acquire,) barrier
            RTDD_XT(2);
            RTDD_XT(3);
#if RTDD_EXCHANGE_ACQUIRE
#pragma unroll
            for (int g = 0; g < G; g++) {
                const int y = y0 + g, ty = tr * G + g;
                const bool central = xin && ty >= hy && ty < eh - hy;
                const bool ok = colok && y >= 0 && y < rows;
                if (ok && !central) {                                    // a halo pixel inside the image: some neighbour's centre
                    const size_t off = (size_t)y * ip + x0;
                    const float4 vx = *(const float4 *)(Ek + off), vp = *(const float4 *)(Em + off);      // plain vector loads behind the agent acquire
                    const float xv[4] = {vx.x, vx.y, vx.z, vx.w}, pv[4] = {vp.x, vp.y, vp.z, vp.w};
#pragma unroll
                    for (int i = 0; i < 4; i++) { const bool in = x0 + i < cols; a[g][i] = in ? xv[i] : 0.0f; b[g][i] = in ? pv[i] : 0.0f; }
                }
            }
#else
            // No acquire: every halo load is a 16-byte sc1 load straight into the tile's registers (the strips were stored sc1 and drained
            // before their owner's flag; the polling wave loads after its poll, the others after the barrier exchange_wait ends with).
            // Round 2 measured the acquire at 0.8-0.9 us of a 15.7 us block.  All loads are issued, then ONE wait that the loaded
            // registers pass through (so that no use can be scheduled in front of it).
            f4v_t hk[G], hm[G];
#pragma unroll
            for (int g = 0; g < G; g++) {
                const int y = y0 + g, ty = tr * G + g;
                const bool central = xin && ty >= hy && ty < eh - hy;
                const bool ok = colok && y >= 0 && y < rows;
                hk[g] = a[g]; hm[g] = b[g];
                if (ok && !central) {                                    // a halo pixel inside the image: some neighbour's centre
                    const size_t off = (size_t)y * ip + x0;
                    load_sc1(hk[g], Ek + off); load_sc1(hm[g], Em + off);
                }
            }
            if constexpr (G == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(hk[0]), "+v"(hm[0]) :: "memory");
            else if constexpr (G == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(hk[0]), "+v"(hm[0]), "+v"(hk[1]), "+v"(hm[1]) :: "memory");
            else if constexpr (G == 3) asm volatile("s_waitcnt vmcnt(0)" : "+v"(hk[0]), "+v"(hm[0]), "+v"(hk[1]), "+v"(hm[1]), "+v"(hk[2]), "+v"(hm[2]) :: "memory");
            else if constexpr (G == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(hk[0]), "+v"(hm[0]), "+v"(hk[1]), "+v"(hm[1]), "+v"(hk[2]), "+v"(hm[2]), "+v"(hk[3]), "+v"(hm[3]) :: "memory");
            else {
                static_assert(G == 6, "add a wait for this G");
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(hk[0]), "+v"(hm[0]), "+v"(hk[1]), "+v"(hm[1]), "+v"(hk[2]), "+v"(hm[2]) :: "memory");
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(hk[3]), "+v"(hm[3]), "+v"(hk[4]), "+v"(hm[4]), "+v"(hk[5]), "+v"(hm[5]) :: "memory");
            }
#pragma unroll
            for (int g = 0; g < G; g++) {
                const int y = y0 + g, ty = tr * G + g;
                const bool central = xin && ty >= hy && ty < eh - hy;
                const bool ok = colok && y >= 0 && y < rows;
                if (ok && !central) {
#pragma unroll
                    for (int i = 0; i < 4; i++) { const bool in = x0 + i < cols; a[g][i] = in ? hk[g][i] : 0.0f; b[g][i] = in ? hm[g][i] : 0.0f; }
                }
            }
#endif
        }
        (void)ntiles;
        RTDD_XT(4);
    }
    // results of the last block go to the exchange buffer of ITS parity (free by the argument above; blk = 0 -> Yk/Ym)
    if (PERSIST && (blk & 1)) { Yk = Xk; Ym = Xm; }
    RTDD_STAMP(2);

    // ---- write back the part that is still exact ---------------------------------------------------
    int tid_w = tid;
    asm volatile("" : "+v"(tid_w));                  // (as in the exchange: keep this address arithmetic out of the sweep loops)
    const int lx_w = tid_w % LX, tr_w = tid_w / LX;
    const int x0_w = bx * TW - hx + 4 * lx_w, y0_w = by * TH - hy + tr_w * G;
    const bool xin_w = x0_w >= 0 && x0_w < cols && 4 * lx_w >= hx && 4 * lx_w < EW - hx;
#pragma unroll
    for (int g = 0; g < G; g++) {
        const int y = y0_w + g, ty = tr_w * G + g, x0 = x0_w;
        if (xin_w && ty >= hy && ty < eh - hy && y < rows) {
            const size_t off = (size_t)y * ip + x0;
            // newest iterate -> Yk, the one before it -> Ym (componentwise selects: a pointer-select would go through scratch)
            const float4 vk = make_float4(odd ? b[g][0] : a[g][0], odd ? b[g][1] : a[g][1], odd ? b[g][2] : a[g][2], odd ? b[g][3] : a[g][3]);
            const float4 vm = make_float4(odd ? a[g][0] : b[g][0], odd ? a[g][1] : b[g][1], odd ? a[g][2] : b[g][2], odd ? a[g][3] : b[g][3]);
            store_result((float4 *)(Yk + off), vk);
            store_result((float4 *)(Ym + off), vm);
        }
    }
    RTDD_STAMP_LAST(3);
}

// ---- the same sweeps in a COLUMN layout (tile 14): a thread owns 1 pixel x 4 rows --------------------------------------------
// For the small pyramid levels every sweep is a chain -- counter poll, row reads, arithmetic, publish -- and with one row of 4
// pixels per thread (tile 9) EVERY row is an edge row: all of a wave's arithmetic sits between its wait and its publish, and each
// thread moves 16 B out and 32 B in through LDS per sweep.  Here a wave is 64 pixels wide and 4 rows tall: horizontal neighbours are
// DPP lane shifts, vertical neighbours are the thread's own registers except above row 0 and below row 3, which are ONE float
// each from LDS (8 B out, 8 B in per thread and sweep), and rows 1 and 2 are computed after the publish, off the neighbours'
// critical path.  Same geometry as tile 9 (64 x 64 extended tile, up to 16 waves), same arithmetic per pixel, bit-identical.
template <bool CONTRACT>
__global__ __launch_bounds__(1024) void k_sweep_col(const float *__restrict__ Xk, const float *__restrict__ Xm, float *__restrict__ Yk, float *__restrict__ Ym,
                                                    const uint32_t *__restrict__ M, const float *__restrict__ lut_g, const float *__restrict__ omegas,
                                                    int ip, int rows, int cols, int hx, int hy, int nsweeps, float gamma, int gx, int gy, int xcd_tiles, int *sync_words, size_t zPlane) {
    constexpr int R = 4;
    RTDD_Z(Xk, zPlane); RTDD_Z(Xm, zPlane); RTDD_Z(Yk, zPlane); RTDD_Z(Ym, zPlane); RTDD_Z(M, zPlane);
    __shared__ float lut[257];
    // [buffer][wave][lane] = (top row value, tag, bottom row value, tag): the tag is the number of the sweep the values are for, + 1,
    // written by a second instruction BEHIND the values (the LDS serves a wave's accesses in order), so a reader that finds the tag
    // it wants in its one 8-byte read of (value, tag) holds the right value -- no separate counter poll, one LDS round trip per sweep.
    __shared__ int4 edge[2][16][64];
    int bx = blockIdx.x, by = blockIdx.y;
    if (xcd_tiles > 0) {
        const int t = ((int)blockIdx.x & 7) * xcd_tiles + ((int)blockIdx.x >> 3);
        if (t >= gx * gy) return;
        bx = t % gx; by = t / gx;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = (int)blockDim.x >> 6;
    const int eh = R * nwv, TW = 64 - 2 * hx, TH = eh - 2 * hy;
    const int x = bx * TW - hx + lane, y0 = by * TH - hy + R * wv;
    const bool colok = x >= 0 && x < cols;
    float a[R], b[R], wr[R], wd[R], wl[R], cnt[R], rcp[R], wu0;
    bool dir[R], unsafe = false;
    // The tile's loads are issued BEFORE the weight table is staged in LDS, so that the launch pays one memory round trip, not two in a
    // row (these levels are 36 + 21 launches of ~13.6 us: -0.7 us each).
    uint32_t mraw[R], mup = 0;
#pragma unroll
    for (int g = 0; g < R; g++) {
        const int y = y0 + g;
        const bool in = colok && y >= 0 && y < rows;
        const size_t off = (size_t)(in ? y : 0) * ip + (in ? x : 0);
        mraw[g] = in ? M[off] : 0u;
        a[g] = in ? Xk[off] : 0.0f;
        b[g] = in ? Xm[off] : 0.0f;
    }
    {
        const int y = y0 - 1;
        if (colok && y >= 0 && y + 1 < rows) mup = M[(size_t)y * ip + x];
    }
    for (int i = tid; i < 257; i += (int)blockDim.x) lut[i] = lut_g[i];
    for (int i = tid; i < 2 * 16 * 64; i += (int)blockDim.x) (&edge[0][0][0])[i] = make_int4(0, 0, 0, 0);
    __syncthreads();
#pragma unroll
    for (int g = 0; g < R; g++) {
        const int y = y0 + g;
        const bool in = colok && y >= 0 && y < rows;
        const uint32_t m = mraw[g];
        wr[g] = (in && x + 1 < cols) ? lut[m & 255] : 0.0f;
        wd[g] = (in && y + 1 < rows) ? lut[(m >> 8) & 255] : 0.0f;
        dir[g] = in && (m & kMetaDirichlet);
        wl[g] = lane_from_prev(wr[g]);           // 0 for lane 0 (bound_ctrl): the image border, or discarded halo
    }
    {
        const int y = y0 - 1;
        const bool ok = colok && y >= 0 && y + 1 < rows;
        wu0 = ok ? lut[(mup >> 8) & 255] : 0.0f;
    }
#pragma unroll
    for (int g = 0; g < R; g++) {
        float c = 0.0f;                          // left, right, up, down (src/GPUSolver.cu:82,88,94,100)
        c += wl[g]; c += wr[g]; c += g == 0 ? wu0 : wd[g - 1]; c += wd[g];
        cnt[g] = c == 0.0f ? 1.0f : c;
        rcp[g] = rcp_rn(cnt[g]);
        unsafe |= cnt[g] < 0x1p-126f;
    }
#ifdef RTDD_TIMING_ASSUME_SAFE
    const bool wave_unsafe = false; (void)unsafe;      // (timing-only diagnostic build: sweep_tile_setup.inc)
#else
    const bool wave_unsafe = __builtin_amdgcn_ballot_w64(unsafe) != 0;
#endif
#ifdef RTDD_TIMING_NO_TINY
    constexpr uint32_t kTinyT = 0u;
#else
    constexpr uint32_t kTinyT = 2u * 0x0D800000u - 1u;
#endif
    const int up_w = wv > 0 ? wv - 1 : wv, dn_w = wv < nwv - 1 ? wv + 1 : wv;
    // (the first / last wave reads its own row instead of a missing neighbour: weighted 0 at the image border, discarded halo elsewhere)
    const int up_half = wv > 0 ? 1 : 0, dn_half = wv < nwv - 1 ? 0 : 1;        // which (value, tag) pair of the entry: 0 = top, 1 = bottom
    auto publish = [&](int buf, float top, float bottom, int tag) {
        int *e = (int *)&edge[buf][wv][lane];
        asm volatile("" ::: "memory");
        e[0] = __float_as_int(top); e[2] = __float_as_int(bottom);
        asm volatile("" ::: "memory");                                      // values first, tags behind them
        __hip_atomic_store(&e[1], tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&e[3], tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
    };

    bool gone = false;                           // a wait inside this workgroup ran into its bound (see the read loop): stop waiting
    const int tile_id_tl = by * gx + bx; (void)tile_id_tl;
    auto sweep = [&](float (&cur)[R], float (&oth)[R], int s, auto fast, bool last, auto parity) {
        constexpr bool FAST = decltype(fast)::value;
        constexpr int buf = decltype(parity)::value;
        float up, dn;
        RTDD_TL(0, s);
        {
            const long long *pu = (const long long *)&edge[buf][up_w][lane] + up_half, *pd = (const long long *)&edge[buf][dn_w][lane] + dn_half;
            auto read_both = [&]() {                 // (value, tag) of either neighbour in one 8-byte read each; true when both tags are there
                const long long u = __hip_atomic_load(pu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const long long d = __hip_atomic_load(pd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                up = __int_as_float((int)u); dn = __int_as_float((int)d);
                const int tu = (int)(u >> 32), td = (int)(d >> 32);
                return __builtin_amdgcn_ballot_w64((tu < td ? tu : td) < s + 1) == 0;
            };
            if (__builtin_expect(!read_both(), 0)) {                 // the straight path above is the hot one: keep the waiting out of it
                // Every wave of a workgroup is resident and a leaving wave marks its tags first, so a wait of seconds is a bug: report it
                // (status 2, persist_sync.hpp) and stop waiting instead of hanging the GPU -- the results are garbage from here on.
                unsigned spins = 0;
                while (!gone) {
                    __builtin_amdgcn_s_sleep(1);
                    if (read_both()) break;
                    if (++spins > (1u << 22)) {
                        __hip_atomic_store(&sync_words[kSyncStatus], 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        gone = true;
                    }
                }
                if (gone) { up = 0.0f; dn = 0.0f; }
            }
        }
        RTDD_TL_ROWS_IN_HAND(s);
        const float omega = omegas[s], gamma_v = gamma;     // (scalar operands: the vector form measured no faster, see k_sweep_blocked)
        // (the lane shifts stay DPP here: through the LDS crossbar -- 8 ds_bpermute_b32 per thread and sweep, as k_sweep_blocked does with
        // its 2 per row -- the coarse levels measured 20 % SLOWER, and with no shifts at all (timing only) no faster: EXPERIMENTS.md)
        auto wsum = [&](int g) {
            const float xl = lane_from_prev(cur[g]);
            const float xu = g == 0 ? up : cur[g - 1], xd = g == R - 1 ? dn : cur[g + 1];
            const float wu = g == 0 ? wu0 : wd[g - 1];
            float sum = 0.0f;
            sum = CONTRACT ? __builtin_fmaf(wl[g], xl, sum) : sum + wl[g] * xl;
            if (CONTRACT) {
                // sum = fma(wr, x of the NEXT lane, sum) with the lane shift as the instruction's own DPP operand (lane 63 reads 0, as
                // lane_from_next gives it): one instruction instead of v_mov_b32_dpp + v_fmac -- hipcc does not fold wave shifts itself
                // (a DPP read needs two wait states after a VALU write of its SOURCE register and five after a VALU write of EXEC; the hazard
                // recogniser does not look inside asm.  The source here is cur[g], last written by the previous sweep's update dozens of
                // instructions earlier, and the loop has no v_cmpx: tests/test_isa_hazards.py checks both on the disassembly of every build
                // instead of paying an s_nop 1 -- 1.5 of ~60 issue cycles per pixel -- in front of every one)
                asm volatile("v_fmac_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(sum) : "v"(cur[g]), "v"(wr[g]));
            } else {
                sum = sum + wr[g] * lane_from_next(cur[g]);
            }
            sum = CONTRACT ? __builtin_fmaf(wu, xu, sum) : sum + wu * xu;
            sum = CONTRACT ? __builtin_fmaf(wd[g], xd, sum) : sum + wd[g] * xd;
            return sum;
        };
        auto pair = [&](int g0, int g1) {        // two rows: sums, quotients, ONE tiny test, updates (as in k_sweep_blocked)
            const float s0 = wsum(g0), s1 = wsum(g1);
            float q0, q1;
            if (FAST) {
                q0 = div_tail(s0, cnt[g0], rcp[g0]); q1 = div_tail(s1, cnt[g1], rcp[g1]);
                const uint32_t t0 = (__float_as_uint(s0) << 1) + 0xFFFFFFFFu, t1 = (__float_as_uint(s1) << 1) + 0xFFFFFFFFu;
                if (__builtin_expect(__builtin_amdgcn_ballot_w64((t0 < t1 ? t0 : t1) < kTinyT) != 0, 0)) { q0 = s0 / cnt[g0]; q1 = s1 / cnt[g1]; }
            } else { q0 = s0 / cnt[g0]; q1 = s1 / cnt[g1]; }
            const int gs[2] = {g0, g1}; const float qs[2] = {q0, q1};
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int g = gs[k];
                const float r = __builtin_amdgcn_fmed3f(qs[k], 0.0f, 255.0f);
                const float xc = cur[g], prev = oth[g];
                const float v = CONTRACT ? __builtin_fmaf(omega, __builtin_fmaf(gamma_v, r - xc, xc) - prev, prev)      // src/GPUSolver.cu:259
                                         : (omega * (gamma_v * (r - xc) + xc - prev)) + prev;
                oth[g] = dir[g] ? xc : v;
            }
        };
        // The rows the neighbouring waves wait for are computed at a raised wave priority and the interior rows at the normal one: the
        // chain read -> first / last row -> publish is what a sweep of these small levels waits on, and with four waves per SIMD the
        // arbiter otherwise gives a wave's interior rows the same share (estimate 1.145 -> 1.119 ms).
        __builtin_amdgcn_s_setprio(1);
        pair(0, R - 1);
        if (!last) publish(buf ^ 1, oth[0], oth[R - 1], s + 2);
        __builtin_amdgcn_s_setprio(0);
        RTDD_TL(2, s);
        pair(1, 2);
        RTDD_TL(3, s);
    };

    publish(0, a[0], a[R - 1], 1);
    using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, 1>;
    // TRAPEZOID: what sweep s has to produce is the written-back region grown by the sweeps still to come, rows
    // [hy - (n-1-s), eh - hy + (n-1-s)) -- it shrinks by a row per sweep from either side.  A wave none of whose four rows lie in it
    // any more has nothing left to contribute (the region only shrinks): it sets its counter to "for ever" so that its neighbours
    // never wait for it again, and leaves.  What its inner neighbour then reads from it is stale but finite, and the rows that
    // value spoils move inwards one per sweep -- exactly as fast as the region's edge, so they are never rows anyone needs.
    // At depth 28 on the 64-row tile 59 % of the wave-sweeps remain, and the last sweeps run on two waves instead of sixteen.
    // (a tile that is the whole level has hy = 0: every wave stays to the end)
    const int n = nsweeps;
    const int a0 = n + 2 - hy + R * wv, a1 = n - 2 - R * wv + eh - hy;
    const int active_until = min(n - 1, min(a0, a1));                // the last sweep this wave takes part in (wave-uniform)
    int s = 0;
    bool odd = false;
    if (!wave_unsafe) {
        for (; s + 1 < n && s + 1 <= active_until; s += 2) { sweep(a, b, s, std::true_type{}, false, P0{}); sweep(b, a, s + 1, std::true_type{}, s + 2 >= n, P1{}); }
        if (s < n && s <= active_until) { sweep(a, b, s, std::true_type{}, s + 1 >= n, P0{}); odd = true; }
    } else {
        for (; s + 1 < n && s + 1 <= active_until; s += 2) { sweep(a, b, s, std::false_type{}, false, P0{}); sweep(b, a, s + 1, std::false_type{}, s + 2 >= n, P1{}); }
        if (s < n && s <= active_until) { sweep(a, b, s, std::false_type{}, s + 1 >= n, P0{}); odd = true; }
    }
    if (active_until < n - 1) {                                      // left early: never hold a neighbour up, nothing of mine is written back
        for (int bf = 0; bf < 2; bf++) {                             // tags "for ever" in both buffers, values left as they are (finite)
            int *e = (int *)&edge[bf][wv][lane];
            __hip_atomic_store(&e[1], 0x7FFFFFFF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(&e[3], 0x7FFFFFFF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        return;
    }
    // write back the part that is still exact: newest iterate -> Yk, the one before -> Ym
    const bool xin = colok && lane >= hx && lane < 64 - hx;
#pragma unroll
    for (int g = 0; g < R; g++) {
        const int y = y0 + g, ty = R * wv + g;
        if (xin && ty >= hy && ty < eh - hy && y < rows) {
            const size_t off = (size_t)y * ip + x;
            Yk[off] = odd ? b[g] : a[g];
            Ym[off] = odd ? a[g] : b[g];
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------
struct TileCfg { int lx, nt, g; };
// id -> (lanes per tile row, threads, rows per thread); extended tile = 4*lx wide, nt/lx*g tall
static const TileCfg kTiles[] = {{0, 0, 0}, {16, 256, 4}, {32, 512, 4}, {32, 1024, 4}, {32, 1024, 3}, {32, 512, 3}, {16, 512, 3}, {16, 256, 3}, {32, 1024, 2},
                                 {16, 1024, 1}, {16, 512, 2}, {32, 1024, 1}, {32, 768, 4}, {32, 512, 6},
                                 {16, 1024, 1} /* 14: tile 9's geometry in the column layout (k_sweep_col) */,
                                 {16, 512, 1} /* 15: the column layout on 64 x 32 (8 waves) */, {16, 768, 1} /* 16: ... on 64 x 48 (12 waves) */};
constexpr int kNumTiles = 16;
static inline bool is_col_tile(int tile) { return tile >= 14; }
static inline int tile_rows(int tile, int nthreads) { return nthreads / kTiles[tile].lx * kTiles[tile].g; }

// Can every workgroup of a persistent launch be resident at once?  Asked of the runtime once per kernel (the answer depends on the
// kernel's registers and LDS): at least one workgroup of `nthreads` threads per CU, and no more workgroups than CUs.  A launch that
// fails this falls back to one launch per block.  (A cooperative launch would make the same check at +15..19 us of host time per
// launch and gives no other guarantee -- MI355X_MICROARCH.md, "Residency and cooperative launch"; a GPU shared with other work
// can still starve a resident-by-count launch, which is what the bounded poll in persist_sync.hpp is for.)
template <typename K>
static bool fits_one_per_cu(K kernel, int nthreads) {
    int nb = 0;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, nthreads, 0) == hipSuccess && nb >= 1;
}

template <int LX, int NT, int G>
static bool persistent_possible(rtdd_ctx *ctx, int tile, int nthreads) {
    // the answer is a property of (device, kernel): cached in the context, which belongs to one device and is driven by one host
    // thread at a time (include/rtdd.h) -- not in a function-local static shared by every context and thread
    signed char &c = ctx->persist_fit[tile][ctx->opt.fp_contract ? 1 : 0];
    if (c < 0) c = ctx->opt.fp_contract ? fits_one_per_cu(k_sweep_blocked<LX, NT, G, true, true>, nthreads) : fits_one_per_cu(k_sweep_blocked<LX, NT, G, false, true>, nthreads);
    return c == 1;
}

template <int LX, int NT, int G>
static void launch_cfg(rtdd_ctx *ctx, dim3 grid, int xcd_tiles, int nthreads, float *Xk, float *Xm, float *Yk, float *Ym, const uint32_t *M,
                       const float *omegas, int ip, int rows, int cols, int hx, int hy, int n, float gamma, int block_sweeps, int flag_base, size_t zPlane, int images) {
    const bool persist = block_sweeps < n;
    const dim3 launch_grid = xcd_tiles > 0 ? dim3(8 * xcd_tiles, 1, images) : dim3(grid.x, grid.y, images);
#define RTDD_LAUNCH(C, P) hipLaunchKernelGGL((k_sweep_blocked<LX, NT, G, C, P>), launch_grid, dim3(nthreads), 0, ctx->stream, Xk, Xm, Yk, Ym, M, ctx->lut_dev, omegas, ip, rows, cols, hx, hy, n, gamma, block_sweeps, ctx->sync_words, (int)grid.x, (int)grid.y, xcd_tiles, flag_base, zPlane)
    if (ctx->opt.fp_contract) { if (persist) RTDD_LAUNCH(true, true); else RTDD_LAUNCH(true, false); }
    else { if (persist) RTDD_LAUNCH(false, true); else RTDD_LAUNCH(false, false); }
#undef RTDD_LAUNCH
}

// ---- configuration cost model -------------------------------------------------------------------------
// Estimated microseconds per sweep of the whole image for (tile, depth T, persistent or not), from measured
// constants (MI355X; scripts/tile_sweep*.sh, size_sweep.sh, ubench/blocked_phases.hip, ubench/launch_gap*.hip):
//   a sweep of one workgroup   max(latency 0.25 + 0.28*G us  [dependent chains, one wave per SIMD is enough to see it],
//                                  throughput 1.5 us per 12288 extended pixels resident on the CU [VALU bound])
//   tile load                  12 B per extended pixel at 22.6 GB/s per CU (5.8 TB/s chip-wide); stores are posted
//   kernel boundary            ~6 us when tens of MB move, ~3.5 us for small levels
//   one round (nWG <= slots)   boundary + load + T sweeps, nothing overlaps
//   several rounds             tiles with >= 2 workgroups per CU hide the load behind the other workgroup's arithmetic
//                              (x1.25 for imperfect overlap); 1-per-CU tiles pay load + sweeps + store per workgroup
//   persistent                 T sweeps + ~8 us halo exchange (6 us for the small tiles); needs nWG <= #CUs, T even,
//                              halo no wider than a neighbour's centre
// It only has to rank candidates; it reproduces the measured launch times within ~15 %.
static const int kWgPerCu[kNumTiles + 1] = {0, 3, 1, 1, 1, 2, 2, 4, 1, 2, 2, 2, 1, 1, 2, 4, 2};

// (images: a batched launch runs the level of that many independent images at once -- rtdd_estimate_depth_batch -- so the chip sees
// images x tiles workgroups: small levels then want SHALLOW halos, few tiles per image and one round, where a single image wants deep
// halos to save launches: 120 x 67 x 64 images: tile 9 at depth 8, 6 tiles per image, instead of the column tile at depth 28, 135 per image)
static double config_cost(const rtdd_ctx *ctx, int rows, int cols, int n, int tile, int T, bool persist, int images = 1) {
    const int G = kTiles[tile].g;
    const int EW = 4 * kTiles[tile].lx, EH = tile_rows(tile, kTiles[tile].nt);
    const int hx = (T + 3) / 4 * 4, TW = EW - 2 * hx, TH = EH - 2 * T;
    if (TW < 8 || TH < 8) return 1e30;
    const double nwg = (double)((cols + TW - 1) / TW) * ((rows + TH - 1) / TH) * images;
    const double cus = ctx->num_cus, ext = (double)EW * EH;
    const double lat = 0.25 + 0.28 * G;
    const double thr1 = 1.5 * ext / 12288.0;                    // one workgroup alone on a CU
    const double img_bytes = (double)rows * cols * 20.0 * images;
    const double small = img_bytes < 2e6 ? 0.5 : 1.0;           // tiny levels (< 100 Kpx) sit in L2: cheaper loads and boundaries
    const double bw = img_bytes > 2e8 ? 17600.0 : 22600.0;      // bytes/us per CU: 4.5 TB/s from HBM (8K), 5.8 TB/s from the Infinity Cache
    const double load1 = small * ext * 12.0 / bw, store1 = small * (double)TW * TH * 8.0 / bw;
    if (persist) {
        if (is_col_tile(tile) || nwg > cus || (T & 1) || hx > TW || T > TH || n <= T) return 1e30;
        return (T * (lat > thr1 ? lat : thr1) + 5.0 + 3.0 * ext / 12288.0) / T;
    }
    const int k = kWgPerCu[tile];
    const double boundary = small < 1.0 ? 3.5 : 6.0;
    const double m = nwg / cus;                                 // workgroups each CU has to process
    double t;
    if (m <= k) {                                               // one round
        const double on_cu = m < 1.0 ? 1.0 : ceil(m);
        const double thr = thr1 * on_cu;
        t = boundary + load1 * on_cu + T * (lat > thr ? lat : thr) + 0.5;
    } else if (k >= 2) {
        const double rounds = ceil(nwg / (cus * k));
        const double m_eff = rounds <= 3 ? rounds * k : m;     // few rounds: the last, partly filled one costs a whole round
        // (1.45: what the single-image choices of rounds 1-3 were calibrated with; the kernels have become faster since, and the batch
        // measurements of round 5 -- scripts/batch_level_ab.py: 64 x 480x270 tile 6 depth 8 7.8 us per sweep, 64 x 960x540 29, 64 x 1080p 110; tiles
        // 5 and 7 alike -- fit 0.95 for the tiles of three rows per thread; tile 9 (one row per thread: 13 us where 0.95 says 8) keeps 1.45)
        const double comp = (images > 1 && G >= 3 ? 0.95 : 1.45) * T * thr1, mem = load1 + store1;
        t = boundary + load1 + m_eff * (comp > mem ? comp : mem);
    } else {
        t = boundary + ceil(m) * (load1 + T * (lat > thr1 ? lat : thr1) + 0.5 * store1);
    }
    // the column layout with its shrinking set of waves (k_sweep_col): measured 0.90-0.92 of tile 9 when every tile is resident at once
    // (120x67 depth 28, 240x135 depth 24).  Only offered there: over several rounds it beats tile 9 but not the larger tiles (4K: 794
    // against 1100 Gpx-it/s).
    if (tile == 14) { if (m > k) return 1e30; t *= 0.9; }
    return t / T;
}

static double choose_config(const rtdd_ctx *ctx, int rows, int cols, int n, int fixed_tile, int fixed_T, int *tile, int *T, bool *persist, int images) {
    static const int tiles[] = {4, 8, 9, 14, 6, 5, 7, 12};
    static const int depths[] = {4, 8, 12, 16, 24, 28};
    double best = 1e30;
    *tile = 9; *T = 8; *persist = false;
    for (int ti : tiles) {
        if (fixed_tile && ti != fixed_tile) continue;
        for (int d : depths) {
            if (fixed_T && d != fixed_T) continue;
            for (int p = 0; p < 2; p++) {
                if (p && !ctx->opt.persistent) continue;
                const double c = config_cost(ctx, rows, cols, n, ti, d, p != 0, images);
                if (c < best) { best = c; *tile = ti; *T = d; *persist = p != 0; }
            }
        }
    }
    if (fixed_tile && best >= 1e30) *tile = fixed_tile;
    if (fixed_T && best >= 1e30) *T = fixed_T;
    if (getenv("RTDD_DEBUG_CONFIG"))
        fprintf(stderr, "[rtdd] %dx%d x %d image(s) n=%d -> tile %d depth %d persistent %d (model %.3f us/sweep)\n", cols, rows, images, n, *tile, *T, (int)*persist, best);
    return best;
}

#define RTDD_ALL_TILES \
    RTDD_TILE_CASE(1, 16, 256, 4) RTDD_TILE_CASE(2, 32, 512, 4) RTDD_TILE_CASE(3, 32, 1024, 4) RTDD_TILE_CASE(4, 32, 1024, 3) RTDD_TILE_CASE(5, 32, 512, 3) \
    RTDD_TILE_CASE(6, 16, 512, 3) RTDD_TILE_CASE(7, 16, 256, 3) RTDD_TILE_CASE(8, 32, 1024, 2) RTDD_TILE_CASE(9, 16, 1024, 1) RTDD_TILE_CASE(10, 16, 512, 2) \
    RTDD_TILE_CASE(11, 32, 1024, 1) RTDD_TILE_CASE(12, 32, 768, 4) RTDD_TILE_CASE(13, 32, 512, 6)

static int launch_sweeps_blocked_impl(rtdd_ctx *ctx, const Level &L, size_t ip, int rows, int cols, const float *omegas_dev, int n, int *pk, int *pm, int *launches, int images);

// Runs n sweeps starting from planes (pk = x_k, pm = x_{k-1}); on return *pk / *pm name the planes
// holding x_{k+n} / x_{k+n-1}.  omegas_dev[0..n) must already be on the device.
// A batch (`images` images, rtdd_estimate_depth_batch) runs as ONE sequence of launches over all images (blockIdx.z) -- unless one
// image alone already fills the chip and would run persistently (1080p: 252 tiles): then image after image, each with the launch a
// single solve gets, which the model prices lower than a launch per block over the whole batch.
int launch_sweeps_blocked(rtdd_ctx *ctx, const Level &L, size_t ip, int rows, int cols, const float *omegas_dev, int n,
                          int *pk, int *pm, int *launches, int images) {
#ifdef RTDD_FORCE_CFG_HOOK
    {   // developer's hook (A/B builds only): RTDD_FORCE_CFG="cols,rows,tile,depth,persistent,per_image;..." pins the choice for a level size
        static const char *env = getenv("RTDD_FORCE_CFG");
        for (const char *q = env; q && *q;) {
            int c = 0, r = 0, ti = 0, d = 0, pe = 0, pi = 0;
            if (sscanf(q, "%d,%d,%d,%d,%d,%d", &c, &r, &ti, &d, &pe, &pi) == 6 && c == cols && r == rows) {
                const Options saved_opt = ctx->opt;
                ctx->opt.tile = ti; ctx->opt.temporal_depth = d; ctx->opt.persistent = pe;
                int rc = RTDD_OK, a = *pk, b = *pm, total = 0;
                if (pi) {
                    for (int i = 0; i < images && rc == RTDD_OK; i++) {
                        a = *pk; b = *pm; int ln = 0;
                        rc = launch_sweeps_blocked_impl(ctx, L.view(i), ip, rows, cols, omegas_dev, n, &a, &b, &ln, 1); total += ln;
                    }
                } else rc = launch_sweeps_blocked_impl(ctx, L, ip, rows, cols, omegas_dev, n, &a, &b, &total, images);
                ctx->opt = saved_opt;
                *pk = a; *pm = b; *launches = total;
                return rc;
            }
            q = strchr(q, ';'); if (q) q++;
        }
    }
#endif
    if (images > 1 && ctx->opt.tile == 0 && ctx->opt.temporal_depth == 0 && !(cols <= 128 && rows <= 96)) {
        int t1, T1, tb, Tb; bool p1, pb;
        const double c1 = choose_config(ctx, rows, cols, n, 0, 0, &t1, &T1, &p1, 1);
        const double cb = choose_config(ctx, rows, cols, n, 0, 0, &tb, &Tb, &pb, images);
        // (the model prices a persistent 1080p launch at 2.5 us per sweep; it runs at 1.5 -- profiles/r05_1080p_jacobi1000_* -- and 64 images
        // one after the other take 5.96 ms where one launch per block over the batch takes 6.83: scripts/batch_level_ab.py)
        if (p1 && 0.6 * c1 * images < cb) {
            int rc = RTDD_OK, a = *pk, b = *pm, total = 0;
            for (int i = 0; i < images && rc == RTDD_OK; i++) {
                a = *pk; b = *pm;
                int ln = 0;
                rc = launch_sweeps_blocked_impl(ctx, L.view(i), ip, rows, cols, omegas_dev, n, &a, &b, &ln, 1);
                total += ln;
            }
            *pk = a; *pm = b; *launches = total;
            return rc;
        }
    }
    return launch_sweeps_blocked_impl(ctx, L, ip, rows, cols, omegas_dev, n, pk, pm, launches, images);
}

static int launch_sweeps_blocked_impl(rtdd_ctx *ctx, const Level &L, size_t ip, int rows, int cols, const float *omegas_dev, int n,
                                      int *pk, int *pm, int *launches, int images) {
    const float gamma = 0.99;
    // Tile / depth / persistence choice: a small cost model calibrated on MI355X measurements
    // (scripts/tile_sweep*.sh, scripts/size_sweep.sh, scripts/ubench/*; tables in profiles/).
    int tile = ctx->opt.tile, T = ctx->opt.temporal_depth;
    bool want_persistent = ctx->opt.persistent != 0;
    if (tile == 0 || T == 0) {
        int bt = 0, bT = 0; bool bp = false;
        if (cols <= 64 && rows <= 64) { bt = 9; bT = 8; }                  // ONE tile, 4 px/thread: all sweeps in one launch
        else if (cols <= 128 && rows <= 32) { bt = 11; bT = 8; }           // ditto
        // a batch whose images each fit ONE 128 x 96 tile: a workgroup per image, every sweep in one launch, no halo and no exchange
        // (64 x 120x67 x 1000 sweeps: 0.76 ms against 1.92 for four persistent tiles per image; a single image is better off spread
        // over the chip in the column layout, 0.46 ms)
        else if (images > 1 && cols <= 128 && rows <= 96) { bt = 4; bT = 8; }
        else choose_config(ctx, rows, cols, n, tile, T, &bt, &bT, &bp, images);
        if (tile == 0) tile = bt;
        if (T == 0) T = bT;
        if (ctx->opt.tile == 0 && ctx->opt.temporal_depth == 0) want_persistent = want_persistent && bp;
    }
    if (tile < 1 || tile > kNumTiles) tile = 1;
    const int EW = 4 * kTiles[tile].lx, EH = tile_rows(tile, kTiles[tile].nt);
    const bool single = cols <= EW && rows <= EH;
    if (T > 28) T = 28;
    while (T > 1 && (EW - 2 * ((T + 3) / 4 * 4) < 8 || EH - 2 * T < 8)) T--;   // keep a non-degenerate written-back region
    int done = 0;
    *launches = 0;
    ctx->last_nominal_depth = single ? n : T;     // (last_info.temporal_depth reports the LAST launch: possibly the short tail block)
    while (done < n) {
        int m = single ? n - done : (n - done < T ? n - done : T);
        const int hy = single ? 0 : (n - done < T ? n - done : T);
        const int hx = single ? 0 : (hy + 3) / 4 * 4;
        // a single tile launches only the thread rows the image needs (whole waves), e.g. 120x67 -> 23 of 32 rows
        int nthreads = kTiles[tile].nt;
        if (single) {
            const int need = (rows + kTiles[tile].g - 1) / kTiles[tile].g * kTiles[tile].lx;
            nthreads = (need + 63) / 64 * 64;
            if (nthreads > kTiles[tile].nt) nthreads = kTiles[tile].nt;
        }
        const int eh = tile_rows(tile, nthreads);
        const int TW = EW - 2 * hx, TH = eh - 2 * hy;
        const dim3 grid((cols + TW - 1) / TW, (rows + TH - 1) / TH);
        // PERSISTENT mode: all remaining sweeps in ONE launch, neighbouring workgroups trade halo strips every T sweeps.
        // Only when every workgroup is certainly co-resident (grid <= #CUs), T is even, and there is more than one block.
        int block_sweeps = m;
        bool persistent = !single && want_persistent && (int)(grid.x * grid.y) * images <= ctx->num_cus && grid.x * grid.y * images <= (unsigned)kSyncMaxTiles &&
                                (T % 2 == 0) && n - done > T && hy == T &&
                                hx <= TW && hy <= TH;      // the halo must lie inside the 8 immediate neighbours' centres
        if (is_col_tile(tile)) persistent = false;          // (the column-layout kernel has no persistent mode)
        if (persistent) {
#define RTDD_TILE_CASE(id, LX_, NT_, G_) case id: persistent = persistent_possible<LX_, NT_, G_>(ctx, tile, kTiles[tile].nt); break;
            switch (tile) { RTDD_ALL_TILES }
#undef RTDD_TILE_CASE
        }
        int flag_base = 0;
        if (persistent) {
            block_sweeps = T;
            m = n - done;
            { const int rc_ = prepare_persistent_launch(ctx, (m + T - 1) / T, &flag_base); if (rc_ != RTDD_OK) return rc_; }   // this launch's flag values, debug words
            note_status_writer(ctx);
        }
        // XCD-aware tile placement (RTDD_XCD_REMAP=0 turns it off): +1.5-4 % persistent (strips traded inside one L2), +8 % at 4K
        // launch-per-block (a tile's halo is its neighbours' centre: the same XCD reads both)
        static const bool xcd_remap = !(getenv("RTDD_XCD_REMAP") && atoi(getenv("RTDD_XCD_REMAP")) == 0);
        const int xcd_tiles = (!single && xcd_remap) ? ((int)(grid.x * grid.y) + 7) / 8 : 0;   // (filling one XCD before the next measured the same)
        // outputs go to the two spare planes, then the pairs swap
        int free0 = -1, free1 = -1;
        for (int i = 0; i < 4; i++) if (i != *pk && i != *pm) { if (free0 < 0) free0 = i; else free1 = i; }
        float *Xk = L.P(*pk, ip), *Xm = L.P(*pm, ip);
        float *Yk = L.P(free0, ip), *Ym = L.P(free1, ip);
        const size_t zPlane = L.elems * sizeof(float);
#define RTDD_TILE_CASE(id, LX_, NT_, G_) \
    case id: launch_cfg<LX_, NT_, G_>(ctx, grid, xcd_tiles, nthreads, Xk, Xm, Yk, Ym, L.M(ip), omegas_dev + done, (int)ip, rows, cols, hx, hy, m, gamma, block_sweeps, flag_base, zPlane, images); break;
        if (is_col_tile(tile)) {
            const dim3 launch_grid = xcd_tiles > 0 ? dim3(8 * xcd_tiles, 1, images) : dim3(grid.x, grid.y, images);
            if (ctx->opt.fp_contract) hipLaunchKernelGGL(k_sweep_col<true>, launch_grid, dim3(nthreads), 0, ctx->stream, Xk, Xm, Yk, Ym, L.M(ip), ctx->lut_dev, omegas_dev + done, (int)ip, rows, cols, hx, hy, m, gamma, (int)grid.x, (int)grid.y, xcd_tiles, ctx->sync_words, zPlane);
            else hipLaunchKernelGGL(k_sweep_col<false>, launch_grid, dim3(nthreads), 0, ctx->stream, Xk, Xm, Yk, Ym, L.M(ip), ctx->lut_dev, omegas_dev + done, (int)ip, rows, cols, hx, hy, m, gamma, (int)grid.x, (int)grid.y, xcd_tiles, ctx->sync_words, zPlane);
        } else
        switch (tile) { RTDD_ALL_TILES }
#undef RTDD_TILE_CASE
        // every blocked launch can set the status word (the bounded wait of a wave for its neighbour waves, status 2, exists in
        // the launch-per-block instantiation too): the next synchronising call must read it whatever the mode
        note_status_writer(ctx);
        if (ctx->opt.debug_force_status) {                          // testing aid (include/rtdd.h): as if a wave of this launch had given up
            RTDD_HIP(ctx, hipMemsetD32Async((hipDeviceptr_t)(ctx->sync_words + kSyncStatus), ctx->opt.debug_force_status == 3 ? 1 : ctx->opt.debug_force_status, 1, ctx->stream));
            ctx->opt.debug_force_status = 0;
        }
        ctx->last_info.kernel = 2; ctx->last_info.tile = tile; ctx->last_info.temporal_depth = persistent ? block_sweeps : m; ctx->last_info.persistent = persistent ? 1 : 0;
        ctx->last_launch_images = images;
        // where the results are: the plain launch writes the spare pair; the persistent one the exchange buffer of its
        // last block's parity (blocks 0,2,.. -> spare pair, 1,3,.. -> the input pair)
        const int nblocks = (m + block_sweeps - 1) / block_sweeps;
        if (((nblocks - 1) & 1) == 0) { *pk = free0; *pm = free1; }
        done += m;
        (*launches)++;
    }
    RTDD_LAUNCH_CHECK(ctx, "k_sweep_blocked");
    return RTDD_OK;
}

}  // namespace rtdd
