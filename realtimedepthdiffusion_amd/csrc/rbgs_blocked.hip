// rbgs_blocked.hip -- red-black Gauss-Seidel sweeps (EXTENSION: the reference has no such solver, SURVEY.md
// section 0), built like the temporally blocked Jacobi kernel: a workgroup loads a tile + halo once, keeps values,
// f32 weights and reciprocals in registers, runs n full sweeps (2n half-sweeps) and writes back what is still exact.
//
// One half-sweep updates the pixels of one colour in place from their four neighbours, all of the other colour:
//     x_i <- clamp(sum_j w_ij x_j / sum_j w_ij, 0, 255)        (the reference's solveDiffusion, src/GPUSolver.cu:73-106,
//                                                                without the Chebyshev extrapolation)
// so inside a half-sweep the order does not matter and the result is bit-identical to oracle/rtdd_oracle.c's sweep.
// With G (rows per thread) even and all tile offsets even, the colour of pixel (g, i) of a thread is the compile-time
// constant (g + i) & 1, so each half-sweep is straight-line code over exactly half of the thread's pixels.
// Every half-sweep invalidates one more ring of the halo: n sweeps need a halo of 2n pixels.
// Vertical neighbours across threads go through LDS with the per-wave handshake of sweep_blocked.hip, horizontal ones
// through DPP wave shifts; the divide is the exhaustively verified 3-op form with the same tiny-numerator fallback.
#include <type_traits>

#include "rtdd_internal.hpp"
#include "persist_sync.hpp"

namespace rtdd {
namespace {

__device__ __forceinline__ float from_prev_lane(float v) {          // (bound_ctrl: the lane without a source reads 0, never used -- sweep_blocked.hip)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xF, 0xF, true));
}
__device__ __forceinline__ float from_next_lane(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xF, 0xF, true));
}
__device__ __forceinline__ float div3(float n, float d, float y) {     // see sweep_blocked.hip: == RN(n/d) for normal d, n = 0 or |n| >= 2^-100
    const float q0 = n * y;
    const float r = __builtin_fmaf(-d, q0, n);
    return __builtin_fmaf(r, y, q0);
}

// SOR: successive over-relaxation, x <- clamp(x + omega (clamp(gs) - x)); omega = 1 is plain Gauss-Seidel and takes the !SOR path.
__device__ __forceinline__ float rcp_rn(float d) {       // == 1.0f/d for every normal d with a normal reciprocal (sweep_blocked.hip)
    const float y0 = __builtin_amdgcn_rcpf(d);
    return __builtin_fmaf(__builtin_fmaf(-d, y0, 1.0f), y0, y0);
}

// write-through (sc1) 16-byte store for the inter-workgroup hand-off of the persistent mode (see sweep_blocked.hip)
__device__ __forceinline__ void store_sc1(float4 *p, float4 v) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    const f4v t = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
}

}  // namespace

// PERSIST: ONE launch for all nsweeps.  The workgroup keeps its tile in registers and, every block_sweeps sweeps (halo =
// 2 * block_sweeps), trades the strips of its centre that the neighbours' halos overlap -- the protocol of
// sweep_blocked.hip's persistent mode (write-through stores, drain, barrier, per-tile flag, bounded poll, one agent
// acquire).  Exchange buffers alternate between Y and the input plane X by block parity; the result goes to the buffer of
// the last block's parity.  Needs every workgroup resident at once (host: grid <= #CUs).
template <int LX, int NT, int G, bool CONTRACT, bool SOR, bool PERSIST>
__global__ __launch_bounds__(NT, 4) void k_rbgs_blocked(float *X, float *Y, const uint32_t *__restrict__ M,
                                                        const float *__restrict__ lut_g, int ip, int rows, int cols, int hx, int hy, int nsweeps, float omega, int gx, int gy, int xcd_tiles,
                                                        int block_sweeps, int *sync_words, int flag_base) {
    static_assert(G % 2 == 0, "the compile-time colour pattern needs an even number of rows per thread");
    constexpr int EW = 4 * LX, NTR = NT / LX;
    typedef float f4r __attribute__((ext_vector_type(4)));
    __shared__ float lut[257];
    __shared__ float4 edge[2][NTR][2][LX];
    __shared__ int published[NT / 64];
    __shared__ int dead_s;                     // the launch has failed (persist_sync.hpp): leave

    // XCD-aware placement as in sweep_blocked.hip: workgroup p (on XCD p % 8) takes tile (p % 8) * xcd_tiles + p / 8
    int bx = blockIdx.x, by = blockIdx.y;
    if (xcd_tiles > 0) {
        const int t = ((int)blockIdx.x & 7) * xcd_tiles + ((int)blockIdx.x >> 3);
        if (t >= gx * gy) return;
        bx = t % gx; by = t / gx;
    }
    const int tid = threadIdx.x;
    for (int i = tid; i < 257; i += (int)blockDim.x) lut[i] = lut_g[i];
    if (tid < NT / 64) published[tid] = 0;
    if (tid == 0) { dead_s = PERSIST && __hip_atomic_load(&sync_words[kSyncStatus], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0; }
    if (PERSIST) {
        // arrival, as in k_sweep_blocked: announce, see the 8 neighbours announce within a millisecond, or leave before any sweep
        if (exchange_wait<false, true>(sync_words, &dead_s, tid, by * gx + bx, bx, by, gx, gy, flag_base)) return;
    } else {
        __syncthreads();
    }

    const int lx = tid % LX, tr = tid / LX;
    const int ntr = (int)blockDim.x / LX, eh = ntr * G;
    const int TW = EW - 2 * hx, TH = eh - 2 * hy;                  // hx multiple of 4, hy even, TH even: pixel (g,i) has colour (g+i)&1
    const int x0 = bx * TW - hx + 4 * lx;
    const int y0 = by * TH - hy + tr * G;
    const bool colok = x0 >= 0 && x0 < cols;

    f4r a[G];
    float wr[G][4], wd[G][4], wl0[G], wu0[4], cnt[G][4], rcp[G][4];
    uint32_t dirichlet = 0;
    bool unsafe = false;
#pragma unroll
    for (int g = 0; g < G; g++) {
        const int y = y0 + g;
        const bool ok = colok && y >= 0 && y < rows;
        float4 vx = make_float4(0, 0, 0, 0);
        uint4 m = make_uint4(0, 0, 0, 0);
        if (ok) { const size_t off = (size_t)y * ip + x0; vx = *(const float4 *)(X + off); m = *(const uint4 *)(M + off); }
        const float xv[4] = {vx.x, vx.y, vx.z, vx.w};
        const uint32_t mv[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const bool in = ok && x0 + i < cols;
            a[g][i] = in ? xv[i] : 0.0f;
            wr[g][i] = (in && x0 + i + 1 < cols) ? lut[mv[i] & 255] : 0.0f;
            wd[g][i] = (in && y + 1 < rows) ? lut[(mv[i] >> 8) & 255] : 0.0f;
            if (!in || (mv[i] & kMetaDirichlet)) dirichlet |= 1u << (g * 4 + i);      // padding is never updated either
        }
        const float w = from_prev_lane(wr[g][3]);
        wl0[g] = (lx > 0 && x0 > 0) ? w : 0.0f;
    }
    {
        const int y = y0 - 1;
        uint4 m = make_uint4(0, 0, 0, 0);
        const bool ok = colok && y >= 0 && y + 1 < rows;
        if (ok) m = *(const uint4 *)(M + (size_t)y * ip + x0);
        const uint32_t mv[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
        for (int i = 0; i < 4; i++) wu0[i] = (ok && x0 + i < cols) ? lut[(mv[i] >> 8) & 255] : 0.0f;
    }
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float wl = i == 0 ? wl0[g] : wr[g][i - 1];
            const float wu = g == 0 ? wu0[i] : wd[g - 1][i];
            float c = 0.0f;
            c += wl; c += wr[g][i]; c += wu; c += wd[g][i];
            cnt[g][i] = c == 0.0f ? 1.0f : c;
            rcp[g][i] = rcp_rn(cnt[g][i]);
            unsafe |= cnt[g][i] < 0x1p-126f;
        }
    const bool wave_unsafe = __builtin_amdgcn_ballot_w64(unsafe) != 0;

    // One half-sweep, written like sweep_blocked.hip's sweep body (see the comments there): the pixels of the colour in two GROUPS
    // of rows -- the thread's first and last row, whose new values the neighbouring thread rows need, then the interior rows --,
    // each group as independent chains (sums, 3-op quotients, ONE branch-free test for numerators too small for the 3-op divide,
    // then clamp / SOR step / Dirichlet select); the edge rows are PUBLISHED before the interior rows are computed; the wait reads
    // both neighbouring waves' counters in one LDS access.  In-place update is safe inside a group: every operand of a pixel of
    // one colour is of the other colour.
    constexpr uint32_t kTinyT = 2u * 0x0D800000u - 1u;          // 2 * bits(2^-100) - 1
    const int wv = tid >> 6, nwv = (int)blockDim.x >> 6;
    const int flag_idx = (tid & 63) == 0 ? (wv > 0 ? wv - 1 : wv) : (wv < nwv - 1 ? wv + 1 : wv);
    auto publish = [&](int h, int buf) {                           // the rows half-sweep h of the neighbours reads
        *(f4r *)&edge[buf][tr][0][lx] = a[0];
        *(f4r *)&edge[buf][tr][1][lx] = a[G - 1];
        __hip_atomic_store(&published[wv], h + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto half = [&](int h, auto fast, auto col, bool last_of_block) {      // h = running half-sweep index; colour = h & 1 = buffer
        constexpr bool FAST = decltype(fast)::value;
        constexpr int C = decltype(col)::value;
        for (;;) {
            const int f = __hip_atomic_load(&published[flag_idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (__builtin_amdgcn_ballot_w64(f < h + 1) == 0) break;      // (one compare and a branch on the lane mask: sweep_blocked.hip)
            __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // the first / last thread row reads its OWN published row instead of a row above / below (finite, and either weighted 0 or discarded halo)
        const float4 up4 = edge[C][tr > 0 ? tr - 1 : 0][tr > 0 ? 1 : 0][lx];
        const float4 dn4 = edge[C][tr < ntr - 1 ? tr + 1 : tr][tr < ntr - 1 ? 0 : 1][lx];
        const float up[4] = {up4.x, up4.y, up4.z, up4.w}, dn[4] = {dn4.x, dn4.y, dn4.z, dn4.w};
        float xl0[G], xr3[G];
        auto wsum = [&](int g, int i) {
            const float xl = i == 0 ? xl0[g] : a[g][i - 1];
            const float xr = i == 3 ? xr3[g] : a[g][i + 1];
            const float xu = g == 0 ? up[i] : a[g - 1][i];
            const float xd = g == G - 1 ? dn[i] : a[g + 1][i];
            const float wl = i == 0 ? wl0[g] : wr[g][i - 1];
            const float wu = g == 0 ? wu0[i] : wd[g - 1][i];
            float sum = 0.0f;
            sum = CONTRACT ? __builtin_fmaf(wl, xl, sum) : sum + wl * xl;
            sum = CONTRACT ? __builtin_fmaf(wr[g][i], xr, sum) : sum + wr[g][i] * xr;
            sum = CONTRACT ? __builtin_fmaf(wu, xu, sum) : sum + wu * xu;
            sum = CONTRACT ? __builtin_fmaf(wd[g][i], xd, sum) : sum + wd[g][i] * xd;
            return sum;
        };
        auto group = [&](auto pick) {
            float q[G][4];
            uint32_t tmin = 0xFFFFFFFFu;
#pragma unroll
            for (int g = 0; g < G; g++) {
                if (!pick(g)) continue;
                // a pixel of colour C in column 0 / 3 needs the neighbouring lane's value (of the other colour: not touched by this half-sweep)
                if (((g + 0) & 1) == C) xl0[g] = from_prev_lane(a[g][3]);
                if (((g + 3) & 1) == C) xr3[g] = from_next_lane(a[g][0]);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if (((g + i) & 1) != C) continue;
                    const float sum = wsum(g, i);
                    if (FAST) { q[g][i] = div3(sum, cnt[g][i], rcp[g][i]); tmin = min(tmin, (__float_as_uint(sum) << 1) + 0xFFFFFFFFu); }
                    else q[g][i] = sum / cnt[g][i];
                }
            }
            if (FAST && __builtin_expect(__builtin_amdgcn_ballot_w64(tmin < kTinyT) != 0, 0)) {      // wave-uniform, rare: full IEEE divide
#pragma unroll
                for (int g = 0; g < G; g++) {
                    if (!pick(g)) continue;
#pragma unroll
                    for (int i = 0; i < 4; i++) if (((g + i) & 1) == C) q[g][i] = wsum(g, i) / cnt[g][i];
                }
            }
#pragma unroll
            for (int g = 0; g < G; g++) {
                if (!pick(g)) continue;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if (((g + i) & 1) != C) continue;
                    const float x = a[g][i];
                    float r = __builtin_amdgcn_fmed3f(q[g][i], 0.0f, 255.0f);
                    if (SOR) { r = CONTRACT ? __builtin_fmaf(omega, r - x, x) : x + omega * (r - x); r = __builtin_amdgcn_fmed3f(r, 0.0f, 255.0f); }
                    a[g][i] = (dirichlet >> (g * 4 + i)) & 1u ? x : r;
                }
            }
        };
        // (first / last rows at a raised wave priority, as k_sweep_blocked: sweep_tile_sweeps.inc)
        __builtin_amdgcn_s_setprio(1);
        group([](int g) { return g == 0 || g == G - 1; });
        if (!last_of_block) publish(h + 1, C ^ 1);
        __builtin_amdgcn_s_setprio(0);
        group([](int g) { return g != 0 && g != G - 1; });
    };

    // colour 0 of the oracle = (x + y) even.  Pixel (g, i) sits at (x0 + i, y0 + g) with x0 % 4 == 0 and y0 even, so its
    // image colour is (g + i) & 1.  (y0 = by*TH - hy + tr*G: every term even.)
    const int tile_id = by * gx + bx;
    int h = 0, s = 0, blk = 0;
    for (;; blk++) {
        const int s_end = min(s + block_sweeps, nsweeps);
        publish(h, 0);                                             // block prologue: the rows the first half-sweep reads (h is even)
        using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>;
        for (; s < s_end; s++) {
            if (!wave_unsafe) { half(h, std::true_type{}, C0{}, false); half(h + 1, std::true_type{}, C1{}, s + 1 >= s_end); }
            else { half(h, std::false_type{}, C0{}, false); half(h + 1, std::false_type{}, C1{}, s + 1 >= s_end); }
            h += 2;
        }
        if (!PERSIST || s >= nsweeps) break;
        float *Ex = (blk & 1) ? X : Y;
        // (addresses and predicates recomputed from a laundered thread index: hoisted out of the block loop they would stay live
        // across the sweeps and push the sweep loop's operands into scratch -- as in sweep_blocked.hip)
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
            if (central && band) store_sc1((float4 *)(Ex + (size_t)y * ip + x0), make_float4(a[g][0], a[g][1], a[g][2], a[g][3]));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // every storing wave drains its write-through stores
        __syncthreads();
#ifndef RTDD_EXCHANGE_ACQUIRE
#define RTDD_EXCHANGE_ACQUIRE 0      // 1: one agent-scope acquire by wave 0 and plain vector loads (the fallback form; built and tested once per round: tests/test_isa_hazards.py, scripts/build_variant.sh)
#endif
        if (exchange_wait<RTDD_EXCHANGE_ACQUIRE != 0>(sync_words, &dead_s, tid, tile_id, bx, by, gx, gy, flag_base + blk + 1)) return;      // flag, bounded poll, (acquire,) barrier (persist_sync.hpp)
#if RTDD_EXCHANGE_ACQUIRE
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int y = y0 + g, ty = tr * G + g;
            const bool central = xin && ty >= hy && ty < eh - hy;
            if (colok && y >= 0 && y < rows && !central) {             // a halo pixel inside the image: some neighbour's centre
                const float4 v = *(const float4 *)(Ex + (size_t)y * ip + x0);      // plain vector load behind the agent acquire
                const float xv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int i = 0; i < 4; i++) a[g][i] = x0 + i < cols ? xv[i] : 0.0f;
            }
        }
#else
        // no agent acquire: every halo load is a 16-byte sc1 load into the tile's registers, all issued, then ONE wait the loaded
        // registers pass through (as in sweep_blocked.hip; MI355X_MICROARCH.md "Valid forms", table row 1).  The load's operand is
        // read-write ("+v"): the register that holds a[g] on the path around the load IS the one the load fills, so the join needs no
        // copy in front of the wait (tests/test_isa_hazards.py checks the disassembly: nothing touches a load's registers before it).
        typedef float f4v_t __attribute__((ext_vector_type(4)));
        f4v_t hv[G];
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int y = y0 + g, ty = tr * G + g;
            const bool central = xin && ty >= hy && ty < eh - hy;
            hv[g] = a[g];
            if (colok && y >= 0 && y < rows && !central) {             // a halo pixel inside the image: some neighbour's centre
                const float *q = Ex + (size_t)y * ip + x0;
                asm volatile("global_load_dwordx4 %0, %1, off sc1" : "+v"(hv[g]) : "v"(q) : "memory");
            }
        }
        static_assert(G == 4, "the wait below lists four registers");
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(hv[0]), "+v"(hv[1]), "+v"(hv[2]), "+v"(hv[3]) :: "memory");
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int y = y0 + g, ty = tr * G + g;
            const bool central = xin && ty >= hy && ty < eh - hy;
            if (colok && y >= 0 && y < rows && !central) {
#pragma unroll
                for (int i = 0; i < 4; i++) a[g][i] = x0 + i < cols ? hv[g][i] : 0.0f;
            }
        }
#endif
    }
    if (PERSIST && (blk & 1)) Y = X;                                     // the last block's parity names the result buffer

    int tid_w = tid;
    asm volatile("" : "+v"(tid_w));
    const int lx_w = tid_w % LX, tr_w = tid_w / LX;
    const int x0_w = bx * TW - hx + 4 * lx_w, y0_w = by * TH - hy + tr_w * G;
    const bool xin_w = x0_w >= 0 && x0_w < cols && 4 * lx_w >= hx && 4 * lx_w < EW - hx;
#pragma unroll
    for (int g = 0; g < G; g++) {
        const int y = y0_w + g, ty = tr_w * G + g;
        if (xin_w && ty >= hy && ty < eh - hy && y < rows) *(f4r *)(Y + (size_t)y * ip + x0_w) = a[g];
    }
}

// 1024-thread tiles waste less on halo (centre 112x112 of 128x128 vs 112x48 of 128x64 at 4 sweeps per launch) but
// there is one per CU instead of two: prefer them when the grid still fills the chip evenly.
static bool rbgs_prefers_big_tile(const rtdd_ctx *ctx, int rows, int cols) {
    // measured (scripts/rbgs_tile_sweep*.sh): 128x128 tiles at 8 sweeps per launch beat 128x64 at 4 from 1080p up
    // (658 vs 462, 771 vs 663, 824 vs 766 Gpx-sweeps/s at 1080p / 4K / 8K); below that the grid no longer covers the chip
    const int centre = 128 - 2 * 16;
    const long tiles = (long)((cols + centre - 1) / centre) * ((rows + centre - 1) / centre);
    return tiles >= (long)ctx->num_cus * 3 / 4;
}

// n full sweeps from plane *plane; on return *plane names the plane holding the result.
int launch_rbgs_blocked(rtdd_ctx *ctx, const Level &L, size_t ip, int rows, int cols, int n, float omega, int *plane, int *launches, int keep) {
    // two shapes: 128x64 (512 threads, 16 px/thread, two workgroups per CU) and 128x128 (1024 threads, one per CU); an image
    // that fits ONE 128x128 tile runs all its sweeps in one launch.  RTDD_OPT_TILE 1/2 forces a shape, RTDD_OPT_TEMPORAL_DEPTH the
    // sweeps per launch (default 8 = 16 half-sweeps = a 16-pixel halo).
    const bool single = cols <= 128 && rows <= 128;
    const bool big = single || ctx->opt.tile == 2 || (ctx->opt.tile == 0 && rbgs_prefers_big_tile(ctx, rows, cols));
    const int EW = 128;
    int depth = ctx->opt.temporal_depth > 0 ? ctx->opt.temporal_depth : 8;
    if (depth > (big ? 24 : 12)) depth = big ? 24 : 12;               // keep a centre of at least 32x16 / 32x32
    int done = 0;
    *launches = 0;
    while (done < n) {
        int m = single ? n - done : (n - done < depth ? n - done : depth);
        const int hy = single ? 0 : 2 * m, hx = single ? 0 : (2 * m + 3) / 4 * 4;
        int nthreads = big ? 1024 : 512;
        if (single) { const int need = (rows + 3) / 4 * 32; nthreads = (need + 63) / 64 * 64; if (nthreads > 1024) nthreads = 1024; }
        const int eh = nthreads / 32 * 4;
        const int TW = EW - 2 * hx, TH = eh - 2 * hy;
        const dim3 grid((cols + TW - 1) / TW, (rows + TH - 1) / TH);
        // persistent: all remaining sweeps in one launch when every tile is resident at once (as in sweep_blocked.hip).  Pays only
        // when the chip is at least half full (1080p: 727 -> 767 Gpx-sweeps/s); smaller grids are launch-bound and an exchange
        // costs more than a launch there (540x960: 265 -> 250), so they keep one launch per block.
        const bool persistent = !single && ctx->opt.persistent && (int)(grid.x * grid.y) <= ctx->num_cus && (int)(grid.x * grid.y) >= ctx->num_cus / 2 && grid.x * grid.y <= 1000 &&
                                n - done > m && m == depth && hx <= TW && hy <= TH;
        int block_sweeps = m, flag_base = 0;
        if (persistent) {
            m = n - done;
            { const int rc_ = prepare_persistent_launch(ctx, (m + block_sweeps - 1) / block_sweeps, &flag_base); if (rc_ != RTDD_OK) return rc_; }   // this launch's flag values, debug words
            note_status_writer(ctx);
        }
        int out = -1;
        for (int i = 0; i < 4; i++) if (i != *plane && i != keep) { out = i; break; }      // `keep`: a plane the caller still needs (-1: none)
        float *X = L.P(*plane, ip);
        float *Y = L.P(out, ip);
        const bool sor = omega != 1.0f;
        const int xcd_tiles = single ? 0 : ((int)(grid.x * grid.y) + 7) / 8;
        const dim3 launch_grid = xcd_tiles > 0 ? dim3(8 * xcd_tiles) : grid;
#define RTDD_RBGS_GO(NT_, C_, S_) do { if (persistent) hipLaunchKernelGGL((k_rbgs_blocked<32, NT_, 4, C_, S_, true>), launch_grid, dim3(nthreads), 0, ctx->stream, X, Y, L.M(ip), ctx->lut_dev, (int)ip, rows, cols, hx, hy, m, omega, (int)grid.x, (int)grid.y, xcd_tiles, block_sweeps, ctx->sync_words, flag_base); \
        else hipLaunchKernelGGL((k_rbgs_blocked<32, NT_, 4, C_, S_, false>), launch_grid, dim3(nthreads), 0, ctx->stream, X, Y, L.M(ip), ctx->lut_dev, (int)ip, rows, cols, hx, hy, m, omega, (int)grid.x, (int)grid.y, xcd_tiles, block_sweeps, ctx->sync_words, flag_base); } while (0)
        const int variant = (big ? 4 : 0) | (ctx->opt.fp_contract ? 2 : 0) | (sor ? 1 : 0);
        switch (variant) {
            case 0: RTDD_RBGS_GO(512, false, false); break;
            case 1: RTDD_RBGS_GO(512, false, true); break;
            case 2: RTDD_RBGS_GO(512, true, false); break;
            case 3: RTDD_RBGS_GO(512, true, true); break;
            case 4: RTDD_RBGS_GO(1024, false, false); break;
            case 5: RTDD_RBGS_GO(1024, false, true); break;
            case 6: RTDD_RBGS_GO(1024, true, false); break;
            default: RTDD_RBGS_GO(1024, true, true); break;
        }
#undef RTDD_RBGS_GO
        ctx->last_info.kernel = 4; ctx->last_info.tile = big ? 2 : 1; ctx->last_info.temporal_depth = persistent ? block_sweeps : m; ctx->last_info.persistent = persistent ? 1 : 0;
        // every image pixel belongs to exactly one tile's centre, so the result plane is complete; it is the spare plane, or --
        // after an odd number of persistent exchanges -- the input plane again
        const int nblocks = (m + block_sweeps - 1) / block_sweeps;
        if (((nblocks - 1) & 1) == 0) *plane = out;
        done += m;
        (*launches)++;
    }
    RTDD_LAUNCH_CHECK(ctx, "k_rbgs_blocked");
    return RTDD_OK;
}

}  // namespace rtdd
