// sweep_stream.hip -- Chebyshev-Jacobi sweeps as a STREAM (round 6, RTDD_OPT_SWEEP_KERNEL = 3; experimental: EXPERIMENTS.md).
//
// k_sweep_blocked keeps a tile in registers for T sweeps and pays for it with a halo on all four sides (1.6 x redundant arithmetic
// at 4K / 8K), LDS hand-offs of edge rows between the waves of a workgroup every sweep, and a tile load + setup per launch that two
// workgroups per CU only partly hide.  Here a WAVE is on its own: it owns a strip of 64 lanes x 4 pixels and walks DOWN it.  Row s of
// the two input iterates enters at step s; in the same step sweep level j (1 .. T) produces row s - j of iterate k + j from the three
// rows of iterate k + j - 1 around it and row s - j of iterate k + j - 2 -- all of them in the wave's own registers: a window of three
// rows per iterate and the per-row constants (weights, divisor, reciprocal, Dirichlet masks) of the T rows in flight.  Row s - T of the
// two newest iterates leaves for memory.  No LDS but the weight table, no barrier, no wait for another wave; a halo only in x (two lanes
// either side) and a warm-up of T rows above and below a chunk of rows.
//
// The windows ROTATE: row r lives in slot r mod 3 of its iterate's window and in slot r mod T of the constants.  With T = 6 a loop
// unrolled six times makes every slot a compile-time register (6 steps x 6 levels of ~70 instructions: ~20 KB of code).
//
// Values: the same operations in the same order as k_sweep_blocked / oracle/rtdd_oracle.c orc_sweep -- weighted sum left, right, up,
// down (src/GPUSolver.cu:79-101), the 3-operation divide on a per-launch reciprocal with the tiny-numerator redo (sweep_common.hpp),
// clamp, update (:259), Dirichlet pixels kept by an EXEC mask on the update's last operation.  Bit-identical.
#include <cstdlib>
#include <type_traits>

#include "rtdd_internal.hpp"
#include "sweep_common.hpp"

namespace rtdd {

constexpr int kStT = 6;            // sweeps per launch = rows of constants in flight
constexpr int kStHx = 8;           // halo in x (pixels; a multiple of the 4 a lane holds, >= kStT)

typedef float st4 __attribute__((ext_vector_type(4)));

template <bool CONTRACT>
__global__ __launch_bounds__(64, 2) void k_sweep_stream(const float *__restrict__ Xk, const float *__restrict__ Xm, float *__restrict__ Yk, float *__restrict__ Ym,
                                                        const uint32_t *__restrict__ M, const float *__restrict__ lut_g, const float *__restrict__ omegas,
                                                        int ip, int rows, int cols, int KH, float gamma) {
    constexpr int T = kStT;
    __shared__ float lut[257];
    const int lane = threadIdx.x;
    for (int i = lane; i < 257; i += 64) lut[i] = lut_g[i];
    __syncthreads();

    constexpr int KW = 256 - 2 * kStHx;
    const int x0 = (int)blockIdx.x * KW - kStHx + 4 * lane;
    const int c0 = (int)blockIdx.y * KH, c1 = min(c0 + KH, rows);
    const int ra = c0 - T, rb = c1 + T;                      // rows that enter: [ra, rb)
    const bool colok = x0 >= 0 && x0 < cols;
    bool cin[4];
#pragma unroll
    for (int i = 0; i < 4; i++) cin[i] = colok && x0 + i < cols;

    st4 X[T + 1][3];                                        // X[j][slot]: iterate k - 1 + j (0: x_{k-1}, 1: x_k, ...), row r in slot (r - ra) mod 3
    st4 Kwr[T], Kwu[T], Kcn[T], Krc[T];                     // per-row constants, row r in slot (r - ra) mod T
    unsigned long long Kfree[T][4];                         // lane masks of the FREE pixels of the row
    bool Kslow[T];                                          // the row holds a denormal divisor: the full divide for it
#pragma unroll
    for (int j = 0; j <= T; j++)
#pragma unroll
        for (int k = 0; k < 3; k++) X[j][k] = st4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < T; k++) {
        Kwr[k] = Kwu[k] = st4{0.0f, 0.0f, 0.0f, 0.0f}; Kcn[k] = Krc[k] = st4{1.0f, 1.0f, 1.0f, 1.0f};
        Kslow[k] = false;
#pragma unroll
        for (int i = 0; i < 4; i++) Kfree[k][i] = ~0ull;
    }
    float om[T];
#pragma unroll
    for (int j = 0; j < T; j++) om[j] = omegas[j];

    // the staged row (loaded a step ahead) and the down-weights of the row before it
    st4 sx = st4{0.0f, 0.0f, 0.0f, 0.0f}, sm = sx;
    uint4 smeta = make_uint4(0, 0, 0, 0);
    auto stage = [&](int y) {                               // issue the loads of row y
        sx = st4{0.0f, 0.0f, 0.0f, 0.0f}; sm = sx; smeta = make_uint4(0, 0, 0, 0);
        if (colok && y >= 0 && y < rows && y < rb) {
            const size_t off = (size_t)y * ip + x0;
            sx = *(const st4 *)(Xk + off); sm = *(const st4 *)(Xm + off); smeta = *(const uint4 *)(M + off);
        }
    };
    st4 wd_prev = st4{0.0f, 0.0f, 0.0f, 0.0f};              // down-weights of the row that entered last = up-weights of the one entering now
    {
        const int y = ra - 1;
        if (colok && y >= 0 && y + 1 < rows) {
            const uint4 m = *(const uint4 *)(M + (size_t)y * ip + x0);
            const uint32_t mv[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
            for (int i = 0; i < 4; i++) wd_prev[i] = lut[cin[i] ? ((mv[i] >> 8) & 255) : 256u];
        }
    }
    stage(ra);

    constexpr uint32_t kTinyT = 2u * 0x0D800000u - 1u;      // bits(2^-100) = 27 << 23 (sweep_tile_sweeps.inc)
    const bool keep_lane = colok && lane >= kStHx / 4 && lane < 64 - kStHx / 4;

    // one sweep level of one row: `o` enters as the row of iterate k + j - 2 and leaves as the row of iterate k + j
    auto level = [&](const st4 &up, const st4 &cur, const st4 &dn, st4 &o, const st4 &wr, const st4 &wu, const st4 &wd, const st4 &cn, const st4 &rc,
                     const unsigned long long (&fm)[4], bool slow, float omega) {
        const float xl0 = lane_from_prev(wr[3] * cur[3]);   // the previous lane's whole left term (its wr[3] is 0 where there is no neighbour)
        const float xr3 = lane_from_next(cur[0]);
        float q[4], sum[4];
        uint32_t tmin = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float xr = i == 3 ? xr3 : cur[i + 1];
            float s = i == 0 ? 0.0f + xl0 : (CONTRACT ? __builtin_fmaf(wr[i - 1], cur[i - 1], 0.0f) : 0.0f + wr[i - 1] * cur[i - 1]);
            s = CONTRACT ? __builtin_fmaf(wr[i], xr, s) : s + wr[i] * xr;
            s = CONTRACT ? __builtin_fmaf(wu[i], up[i], s) : s + wu[i] * up[i];
            s = CONTRACT ? __builtin_fmaf(wd[i], dn[i], s) : s + wd[i] * dn[i];
            sum[i] = s;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            q[i] = div_tail(sum[i], cn[i], rc[i]);
            tmin = min(tmin, (__float_as_uint(sum[i]) << 1) + 0xFFFFFFFFu);
        }
        // ONE copy of the full divide per level: a row with a denormal divisor and a tiny numerator take the same redo
        if (__builtin_expect(slow || __builtin_amdgcn_ballot_w64(tmin < kTinyT) != 0, 0)) {
#pragma unroll
            for (int i = 0; i < 4; i++) q[i] = sum[i] / cn[i];
        }
        float omega_v = omega, gamma_v = gamma;
        asm volatile("" : "+v"(omega_v), "+v"(gamma_v));
        float t[4], ov[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float r = __builtin_amdgcn_fmed3f(q[i], 0.0f, 255.0f);
            const float x = cur[i];
            ov[i] = o[i];
            t[i] = (CONTRACT ? __builtin_fmaf(gamma_v, r - x, x) : gamma_v * (r - x) + x) - ov[i];
            if (!CONTRACT) t[i] = omega_v * t[i];
        }
        if (CONTRACT) masked_fmac4(ov[0], ov[1], ov[2], ov[3], omega_v, t[0], t[1], t[2], t[3], fm[0], fm[1], fm[2], fm[3]);
        else masked_add4(ov[0], ov[1], ov[2], ov[3], t[0], t[1], t[2], t[3], fm[0], fm[1], fm[2], fm[3]);
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = ov[i];
    };

    auto step = [&](auto S_, int s) {
        constexpr int S = decltype(S_)::value;               // (s - ra) mod 6: every slot below is a compile-time constant
        constexpr int s3 = S % 3, s6 = S % T;
        // (1) row s enters: the staged values become the newest rows of the two input iterates
        const bool row_in = s >= 0 && s < rows && s < rb;
        const uint32_t mv[4] = {smeta.x, smeta.y, smeta.z, smeta.w};
        st4 xin, min_;
#pragma unroll
        for (int i = 0; i < 4; i++) { const bool in = row_in && cin[i]; xin[i] = in ? sx[i] : 0.0f; min_[i] = in ? sm[i] : 0.0f; }
        X[1][s3] = xin; X[0][s3] = min_;
        const st4 wu_s = wd_prev;                            // up-weights of row s = down-weights of row s - 1
        // (2) the next row's loads: a whole step to land
        stage(s + 1);
        // (3) the levels: level j produces row s - j of iterate k + j
        st4 newest = st4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int j = 1; j <= T; j++) {
            const int a = (S - j + 12) % 3, au = (S - j - 1 + 12) % 3, ad = (S - j + 1 + 12) % 3;     // slots of rows s - j, s - j - 1, s - j + 1
            const int k = (S - j + 12) % T, kd = (S - j + 1 + 12) % T;                                 // constants of row s - j; of the row below it
            const st4 wd = j == 1 ? wu_s : Kwu[kd];          // down-weights of row r = up-weights of row r + 1
            st4 o = X[j - 1][a];
            level(X[j][au], X[j][a], X[j][ad], o, Kwr[k], Kwu[k], wd, Kcn[k], Krc[k], Kfree[k], Kslow[k], om[j - 1]);
            if (j < T) X[j + 1][a] = o; else newest = o;
        }
        // (4) row s - T of the two newest iterates leaves
        {
            const int y = s - T;
            if (y >= c0 && y < c1 && keep_lane) {
                const size_t off = (size_t)y * ip + x0;
                *(st4 *)(Yk + off) = newest;
                *(st4 *)(Ym + off) = X[T][(S - T + 12) % 3];
            }
        }
        // (5) the constants of row s (first used by level 1 in the next step)
        {
            st4 wr, wd, cn, rc;
            uint32_t dirbits = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const bool in = row_in && cin[i];
                wr[i] = lut[(in && x0 + i + 1 < cols) ? (mv[i] & 255) : 256u];
                wd[i] = lut[(in && s + 1 < rows) ? ((mv[i] >> 8) & 255) : 256u];
                if (in && (mv[i] & kMetaDirichlet)) dirbits |= 1u << i;
            }
            if (lane == 63) wr[3] = 0.0f;                   // no lane to the right: discarded halo (or the image border, where it is 0 anyway)
            const float wl0 = lane_from_prev(wr[3]);
            bool unsafe = false;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float wl = i == 0 ? wl0 : wr[i - 1];
                float c = 0.0f;                              // left, right, up, down (src/GPUSolver.cu:82,88,94,100)
                c += wl; c += wr[i]; c += wu_s[i]; c += wd[i];
                cn[i] = c == 0.0f ? 1.0f : c;
                rc[i] = rcp_rn(cn[i]);
                unsafe |= cn[i] < 0x1p-126f;
            }
            Kwr[s6] = wr; Kwu[s6] = wu_s; Kcn[s6] = cn; Krc[s6] = rc;
            Kslow[s6] = __builtin_amdgcn_ballot_w64(unsafe) != 0;
#pragma unroll
            for (int i = 0; i < 4; i++) Kfree[s6][i] = __builtin_amdgcn_ballot_w64(!((dirbits >> i) & 1u));
            wd_prev = wd;
        }
    };

    for (int s = ra; s < rb; s += 6) {
        step(std::integral_constant<int, 0>{}, s);
        if (s + 1 < rb) step(std::integral_constant<int, 1>{}, s + 1);
        if (s + 2 < rb) step(std::integral_constant<int, 2>{}, s + 2);
        if (s + 3 < rb) step(std::integral_constant<int, 3>{}, s + 3);
        if (s + 4 < rb) step(std::integral_constant<int, 4>{}, s + 4);
        if (s + 5 < rb) step(std::integral_constant<int, 5>{}, s + 5);
    }
}

// n sweeps from (plane *pk = x_k, plane *pm = x_{k-1}); on return *pk / *pm name the planes holding x_{k+n} / x_{k+n-1}.  Whole launches of
// kStT sweeps stream; the last n mod kStT sweeps take the blocked kernel.
int launch_sweeps_stream(rtdd_ctx *ctx, const Level &L, size_t ip, int rows, int cols, const float *omegas_dev, int n, int *pk, int *pm, int *launches) {
    const float gamma = 0.99;
    static const int kh_env = getenv("RTDD_STREAM_ROWS") ? atoi(getenv("RTDD_STREAM_ROWS")) : 0;      // developer's knob: rows per chunk
    constexpr int KW = 256 - 2 * kStHx;
    const int strips = (cols + KW - 1) / KW;
    // rows per chunk: about two waves per SIMD over the chip (2048 waves), at least 32 rows
    int KH = kh_env > 0 ? kh_env : (int)(((long)rows * strips + 2047) / 2048);
    if (KH < 32) KH = 32;
    if (KH > rows) KH = rows;
    const dim3 grid(strips, (rows + KH - 1) / KH);
    int done = 0;
    *launches = 0;
    while (n - done >= kStT) {
        int free0 = -1, free1 = -1;
        for (int i = 0; i < 4; i++) if (i != *pk && i != *pm) { if (free0 < 0) free0 = i; else free1 = i; }
        if (ctx->opt.fp_contract)
            hipLaunchKernelGGL(k_sweep_stream<true>, grid, dim3(64), 0, ctx->stream, L.P(*pk, ip), L.P(*pm, ip), L.P(free0, ip), L.P(free1, ip), L.M(ip), ctx->lut_dev,
                               omegas_dev + done, (int)ip, rows, cols, KH, gamma);
        else
            hipLaunchKernelGGL(k_sweep_stream<false>, grid, dim3(64), 0, ctx->stream, L.P(*pk, ip), L.P(*pm, ip), L.P(free0, ip), L.P(free1, ip), L.M(ip), ctx->lut_dev,
                               omegas_dev + done, (int)ip, rows, cols, KH, gamma);
        *pk = free0; *pm = free1;
        done += kStT;
        (*launches)++;
    }
    RTDD_LAUNCH_CHECK(ctx, "k_sweep_stream");
    ctx->last_info.kernel = 5; ctx->last_info.tile = 0; ctx->last_info.temporal_depth = kStT; ctx->last_info.persistent = 0;
    ctx->last_launch_images = 1; ctx->last_nominal_depth = kStT;
    if (done < n) {
        int ln = 0;
        const rtdd_solve_info keep = ctx->last_info;
        const int rc = launch_sweeps_blocked(ctx, L, ip, rows, cols, omegas_dev + done, n - done, pk, pm, &ln, 1);
        if (rc != RTDD_OK) return rc;
        *launches += ln;
        if (done > 0) ctx->last_info = keep;
    }
    return RTDD_OK;
}

}  // namespace rtdd
