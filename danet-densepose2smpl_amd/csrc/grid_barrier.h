// A grid-wide barrier for launches whose workgroups are all resident at once (gfx950; the caller guarantees the co-residency:
// norm_act.hip's one-pass BatchNorm backward, smpl_lbs.hip's one-launch SMPL backward).  Shared by the kernels of ONE stream: the
// state is reused launch after launch and launches that use it must not overlap.
#pragma once
#include <hip/hip_runtime.h>

namespace danet {

// State (GRID_BAR_WORDS uints, zeroed once by the caller): word 2 = error flag (non-zero: a wait expired; the value is the caller's err_code); group g (workgroups with id % 8 == g:
// observed to share an XCD, which only matters for speed) owns the 64-byte lines at 16 * (1 + g) (arrivals) and
// 16 * (9 + g) (generation); the line at 16 * 17 counts the groups that are complete.
constexpr int GRID_BAR_WORDS = 16 * 18;
__device__ inline void grid_barrier(unsigned* bar, unsigned nblocks, unsigned err_code = 1u) {
    // Two levels: a workgroup arrives at its group's counter; the last one of a group arrives at the top counter; the last
    // group bumps every group's generation word, on which that group's workgroups spin.  512 arrivals on one word and 512
    // pollers of one word cost 17 us per barrier (measured with the phase knob of tools/experiments/bn_micro.cpp); spread
    // over 8 + 1 words the hand-off is a few us.
    // No agent-scope fences: on this chip they write back / invalidate the XCD's whole L2 (78 us per barrier, measured).
    // What crosses the barrier are device-scope float atomics (performed at the memory side, coherent across the XCDs'
    // L2s) that every wave has waited for (workgroup-scope release = s_waitcnt) before its workgroup arrives; after the
    // barrier they are read with agent-scope (sc1, L1-bypassing) loads: reduce_replicas_sc1.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned g = blockIdx.x & 7u;
        const unsigned gsize = (nblocks + 7u - g) >> 3;                          // ids congruent to g below nblocks
        const unsigned ngroups = nblocks < 8u ? nblocks : 8u;
        unsigned* const cnt = bar + 16 * (1 + g);
        unsigned* const gen = bar + 16 * (9 + g);
        unsigned* const top = bar + 16 * 17;
        const unsigned my_gen = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool released = false;
        if (__hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1) {
            __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngroups - 1) {
                __hip_atomic_store(top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");          // the resets are performed before the release
                for (unsigned h = 0; h < ngroups; ++h)
                    __hip_atomic_fetch_add(bar + 16 * (9 + h), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                released = true;
            }
        }
        if (!released) {
            unsigned spins = 0;
            while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == my_gen) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 22)) { __hip_atomic_store(&bar[2], err_code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
        }
    }
    __syncthreads();
}


// The same barrier for data that crosses it as PLAIN global stores / loads (bulk partial results, not just atomics): every wave
// releases at agent scope before its workgroup arrives (its stores are written back to where the other XCDs' L2s can see them) and
// acquires after the barrier (stale lines are dropped).  The write-back costs time in proportion to the dirty lines of the XCD's L2
// -- 78 us per barrier inside the 100 MB BatchNorm kernel, which is why grid_barrier itself carries no fences --, a few us in the SMPL
// backward, whose kernels have written a few MB.
__device__ inline void grid_barrier_fenced(unsigned* bar, unsigned nblocks, unsigned err_code = 1u) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    grid_barrier(bar, nblocks, err_code);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

}  // namespace danet
