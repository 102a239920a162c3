// persist_sync.hpp -- the inter-workgroup hand-off of the persistent kernels (sweep_blocked.hip, rbgs_blocked.hip), device side,
// and the host helpers around it (api.cpp).
//
// Control words (rtdd_ctx::sync_words, device memory, one set per context):
//   [kSyncStatus]    0 = fine; 1 = a workgroup gave up waiting for a neighbouring TILE (the launch was not fully co-resident:
//                    GPU shared with another stream or process); 2 = a wave gave up waiting for a neighbouring WAVE of its own
//                    workgroup (cannot happen -- every wave of a workgroup is resident -- so: a protocol bug).  Sticky until the
//                    host reads it (rtdd_ctx_synchronize and every other call that synchronises anyway) and returns
//                    RTDD_ERR_TIMEOUT.  Once it is set every workgroup that sees it stops exchanging and returns, and every
//                    later persistent launch returns at once, so a failed estimate drains in microseconds instead of spinning
//                    through its remaining exchanges.
//   [kSyncWithhold]  debug (RTDD_OPT_DEBUG_WITHHOLD_TILE): tile number + 1 whose flag is never published (0 = off), to make the
//                    timeout path testable.
//   [kSyncLimit]     poll limit in 10 ns ticks of s_memrealtime (0 = kDefaultPollLimit).
//   [kSyncFailedSeq] sequence number of the first solve whose copy-back kernel (k_finish / k_pyrup_inject) found the status word set and
//                    therefore stored nothing: the host re-runs the pending calls from that one on (api.cpp heal_pending).  0 = none.
//   [kSyncConfirmPtr] (two ints, an address) where the copy-back kernels report the sequence number of a solve whose result they DID publish:
//                    a word in page-locked host memory, so that the host can drop confirmed calls from its log without synchronising
//                    (api.cpp prune_confirmed).  Written by one lane, only while the status word is clear.
//   [kSyncFlags ..]  one block counter per tile.  Monotonic over the life of the context: launch L's workgroups publish base_L + block
//                    number, base_L handed in by the host (api.cpp prepare_persistent_launch), so nothing is zeroed between launches.
#pragma once
#include <hip/hip_runtime.h>

namespace rtdd {

// kSyncFlagStride: ints between the flags of consecutive tiles.  64 = one flag per 256 bytes (its own line, and neighbouring tiles on different memory channels): a tile's flag is stored once and polled
// by up to 8 neighbours, all through memory (sc1); packed 32 to a line (round 2) every store and poll of 32 tiles met on one line: 1080p 1.17 -> 1.24 Tpx-it/s with one line each, +1.5 % more at 256 bytes.
constexpr int kSyncStatus = 0, kSyncWithhold = 1, kSyncLimit = 2, kSyncNonLocal = 3 /* k_defocus_tile met windows beyond its region (effect_kernels.hip) */, kSyncFailedSeq = 4,
              kSyncConfirmPtr = 6 /* two ints: the device address of the context's page-locked `confirmed sequence number` word */, kSyncFlags = 32, kSyncMaxTiles = 1024, kSyncFlagStride = 64;
constexpr int kSyncWords = kSyncFlags + kSyncMaxTiles * kSyncFlagStride;              // size of sync_words in ints
constexpr unsigned long long kDefaultPollLimit = 20000000ull;      // 200 ms: legitimate waits are microseconds
// Round 6: the FIRST thing a persistent workgroup does is announce itself (its flag := the launch's base value) and wait for its
// neighbours' announcements -- while its tile loads are in flight -- with THIS bound: on a GPU shared with other work a launch whose
// workgroups are not all resident is found out here, before any sweep, in 1.5 ms instead of 200 (profiles/r05_shared_gpu.txt: a
// 200 ms stall per time-out in a 1 ms frame loop).  A resident launch's workgroups all start within tens of microseconds.  The
// exchanges behind the first keep the long bound: once every workgroup has been seen running, a long wait is no scheduling accident.
constexpr unsigned long long kArrivalPollLimit = 150000ull;        // 1.5 ms

#ifdef __HIPCC__
// The kernels that publish a solve's result into the caller's buffers call this first (every thread; `first` = one thread of the
// grid): true = a sweep launch in front of them gave up, store nothing.  The reference's solver always leaves a valid depth map
// (src/GPUSolver.cu:311-314); here a failed solve leaves its INPUT in place so that the host can run it again.
__device__ __forceinline__ bool solve_is_dead(int *sync_words, int seq, bool first) {
    if (!sync_words) return false;
    if (__hip_atomic_load(&sync_words[kSyncStatus], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        if (first && seq != 0) {                   // this solve's result is being published: tell the host (one lane, one posted write)
            int *confirm = *(int *const *)(sync_words + kSyncConfirmPtr);
            if (confirm) __hip_atomic_store(confirm, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return false;
    }
    if (first && __hip_atomic_load(&sync_words[kSyncFailedSeq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
        __hip_atomic_store(&sync_words[kSyncFailedSeq], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

// Called by EVERY thread of the workgroup after its payload stores are drained (s_waitcnt vmcnt(0)) and a __syncthreads().
// Publishes this tile's counter, waits for the up-to-8 neighbouring tiles' counters, makes their payload visible (one agent
// acquire by wave 0) and ends with a __syncthreads().  Returns true when the launch is dead (see kSyncStatus): the caller
// leaves.  `dead_lds` is a __shared__ int, zero at kernel start.
// Protocol (cdna_hip_programming.md Guideline 16, R1): write-through (sc1) payload stores; EVERY storing wave drains vmcnt;
// workgroup barrier; ONE lane stores the flag (agent-scope atomic); 8 lanes poll the neighbours' flags relaxed with s_sleep;
// ONE agent acquire; barrier; plain vector loads.
template <bool ACQUIRE = true, bool ARRIVAL = false>
__device__ __forceinline__ bool exchange_wait(int *sync_words, int *dead_lds, int tid, int tile_id, int bx, int by, int gx, int gy, int value, int tile_base = 0) {
    int *flags = sync_words + kSyncFlags + tile_base * kSyncFlagStride;      // (tile_base: a batched launch gives every image's tiles flags of their own)
    if (tid == 0 && __hip_atomic_load(&sync_words[kSyncWithhold], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tile_id + 1)
        __hip_atomic_store(&flags[tile_id * kSyncFlagStride], value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // Lane i (of 9, i != 4) of wave 0 polls neighbour (i%3-1, i/3-1).  A poll is a round trip to memory (the flag was stored sc1: it is in
    // no L2).  (Several polling waves, each started a fraction of a trip after the one before and the first to see all its neighbours
    // telling the others through LDS, measured no faster: EXPERIMENTS.md.)
    const int pw = tid >> 6, pl = tid & 63;
    if (pw == 0 && pl < 9 && pl != 4) {
        const int nx = bx + pl % 3 - 1, ny = by + pl / 3 - 1;
        if (nx >= 0 && ny >= 0 && nx < gx && ny < gy) {
            const int nb = ny * gx + nx;
            unsigned long long t0 = 0, limit = 0;
            unsigned spins = 0;
            while (__hip_atomic_load(&flags[nb * kSyncFlagStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - value < 0) {
                __builtin_amdgcn_s_sleep(4);
                if ((++spins & 63u) == 0) {                              // every 64 polls (tens of microseconds): clock and status word
                    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                    if (t0 == 0) {
                        t0 = now;
                        const int l = __hip_atomic_load(&sync_words[kSyncLimit], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        limit = l > 0 ? (unsigned long long)l : kDefaultPollLimit;
                        if (ARRIVAL && limit > kArrivalPollLimit) limit = kArrivalPollLimit;
                    }
                    const bool failed = __hip_atomic_load(&sync_words[kSyncStatus], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                    if (failed || now - t0 > limit) {
                        if (!failed) __hip_atomic_store(&sync_words[kSyncStatus], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(dead_lds, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        break;
                    }
                }
            }
        }
    }
    // ACQUIRE = false: the caller reads the handed-off bytes with 16-byte sc1 loads only (MI355X_MICROARCH.md, "Valid forms": the
    // polling wave after its poll has matched, the other waves after the barrier below)
    if (ACQUIRE && tid < 64) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    if (!ACQUIRE && tid < 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the polls themselves have returned
    __syncthreads();
    return __hip_atomic_load(dead_lds, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
}

#endif

}  // namespace rtdd
