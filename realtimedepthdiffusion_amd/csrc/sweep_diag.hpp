// sweep_diag.hpp -- the timing hooks of the blocked sweep kernels.  In the product build every hook is empty.  The two diagnostic
// micro-benchmarks define RTDD_STAMPS (scripts/ubench/blocked_phases.hip: per-workgroup phase time stamps, and the time a workgroup
// spends in each step of the persistent hand-off) or RTDD_TIMELINE (scripts/ubench/sweep_timeline.hip: per-wave, per-sweep stamps of ONE
// workgroup) before they include sweep_blocked.hip; nothing else does.
#pragma once

#ifdef RTDD_STAMPS
__device__ unsigned long long g_stamps[4096][5];
// k = 0: earliest wave (atomicMin would need init; wave 0 starts first in practice); k >= 1: LATEST wave of the workgroup
// (atomicMax) -- without a barrier the oldest wave of each SIMD runs ahead, so stamping only wave 0 under-reports.
#define RTDD_STAMP(k) do { if ((threadIdx.x & 63) == 0 && blockIdx.y * gridDim.x + blockIdx.x < 4096) { \
        if ((k) == 0) { if (threadIdx.x == 0) g_stamps[blockIdx.y * gridDim.x + blockIdx.x][0] = __builtin_amdgcn_s_memrealtime(); } \
        else atomicMax(&g_stamps[blockIdx.y * gridDim.x + blockIdx.x][k], (unsigned long long)__builtin_amdgcn_s_memrealtime()); } } while (0)
#define RTDD_STAMP_LAST(k) do { __builtin_amdgcn_s_waitcnt(0); RTDD_STAMP(k); } while (0)
// (round 6) the tile's loads have LANDED: a wait the product build does not have at this point, so that the stamp splits `load + setup`
#define RTDD_STAMP_LOADED(k) do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); RTDD_STAMP(k); } while (0)
__device__ unsigned long long g_xphase[4096][6];
#define RTDD_XT(k) do { if (threadIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); g_xphase[blockIdx.y * gridDim.x + blockIdx.x][k] += t_ - xt_; xt_ = t_; } } while (0)
#define RTDD_XT_BEGIN unsigned long long xt_ = __builtin_amdgcn_s_memrealtime()
#else
#define RTDD_STAMP(k) do {} while (0)
#define RTDD_STAMP_LAST(k) do {} while (0)
#define RTDD_STAMP_LOADED(k) do {} while (0)
#define RTDD_XT(k) do {} while (0)
#define RTDD_XT_BEGIN do {} while (0)
#endif

#ifdef RTDD_TIMELINE
__device__ unsigned long long g_tl[16][64][4];      // [wave][sweep][0 top of sweep, 1 neighbours' rows in hand, 2 own edge rows published, 3 end]
__device__ int g_tl_tile = 100;
#define RTDD_TL(k, sw) do { if (tile_id_tl == g_tl_tile && (threadIdx.x & 63) == 0 && (sw) < 64) g_tl[threadIdx.x >> 6][(sw)][k] = __builtin_amdgcn_s_memtime(); } while (0)
#define RTDD_TL_ROWS_IN_HAND(sw) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); RTDD_TL(1, sw); } while (0)
#else
#define RTDD_TL(k, sw) do {} while (0)
#define RTDD_TL_ROWS_IN_HAND(sw) do {} while (0)
#endif
