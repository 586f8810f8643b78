// chain_wait.cuh — the GPU side of the commit's root -> challenge hand-off: wait for the host to publish a round's fold constants.
// Shared by the wait kernel, the fused small round (round_kernels.hip) and the fold kernels (hash_kernels.hip), which wait themselves
// when the records are on the card (api_fri.hip: ensure_chain).
#pragma once
#include "kernels.h"

#define POLL_WORD(p) __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
// Post-mortem of a wait that ran out of spin, into the chain header (pinned host memory, 16 words): what was expected, what the
// record held — read once more with the polling load, with a non-temporal load, and its neighbour word in the same 64-byte
// line (the host never writes that one: it must read 0) —, where, how many polls, on which XCC, and the 100 MHz wall clock at
// the start of the spin and now.  The host compares those two stamps with its own record of when it published (api_fri.hip).
__device__ static void gave_up_record(uint32_t* h_abort, const uint32_t* h_seq, uint32_t gen, uint32_t where, uint32_t polls, uint64_t t0, bool round_wg) {
    const uint64_t t1 = __builtin_amdgcn_s_memrealtime();
    __hip_atomic_store(h_abort + 1, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(h_abort + 2, POLL_WORD(h_seq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(h_abort + 3, (uint32_t)(reinterpret_cast<uintptr_t>(h_seq) - reinterpret_cast<uintptr_t>(h_abort)) | (round_wg ? 0x80000000u : 0u), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(h_abort + 4, where, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(h_abort + 5, polls, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(h_abort + 6, (uint32_t)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(h_abort + 7, __builtin_nontemporal_load(h_seq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(reinterpret_cast<uint64_t*>(h_abort + 8), t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(reinterpret_cast<uint64_t*>(h_abort + 10), t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(h_abort + 14, POLL_WORD(h_seq + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(h_abort, 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Lane 0 of a workgroup: poll the sequence word of a c0 record until it carries w.gen (bit 31 = the host aborts), the device abort word
// is raised (another workgroup or an earlier kernel gave up) or the spin budget runs out (then raise the abort word and leave the
// post-mortem).  Returns 0 = published, 1 = abort, 2 = gave up.
__device__ __forceinline__ uint32_t chain_wait_poll(const ChainWait& w, bool round_wg) {
    uint32_t st = 0;
    if (__hip_atomic_load(w.d_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return 1;
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    uint32_t polls = 0;
    for (;;) {
        const uint32_t v = POLL_WORD(w.seq);
        if ((v & 0x7fffffffu) == w.gen) { if (v >> 31) st = 1; break; }
        if (__builtin_amdgcn_s_memrealtime() - t0 > w.timeout_ticks) { st = 2; break; }
        if ((++polls & 63u) == 0) {
            if (__hip_atomic_load(w.d_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return 1;
            if ((polls & 255u) == 0) __atomic_thread_fence(__ATOMIC_ACQUIRE);            // belt and braces: drop any cached copy of the record's line
        }
        __builtin_amdgcn_s_sleep(2);
    }
    if (st) {
        const uint32_t before = __hip_atomic_exchange(w.d_abort, st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (st == 2 && before == 0) gave_up_record(w.h_abort, w.seq, w.gen, blockIdx.x | (gridDim.x << 16), polls, t0, round_wg);   // the first to give up writes the post-mortem
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);                                             // system scope: the c0 loads come after the poll
    return st;
}
// All threads of a workgroup (one LDS word `s_state`): true = go on, false = the round is abandoned, return at once.
__device__ __forceinline__ bool chain_wait_wg(const ChainWait& w, uint32_t* s_state, bool round_wg) {
    if (threadIdx.x == 0) *s_state = chain_wait_poll(w, round_wg);
    __syncthreads();
    return *s_state == 0;
}
// a fold constant out of its record (pinned host memory or uncached device memory the host writes through the BAR)
__device__ __forceinline__ fp load_fp_sys(const fp* p) {
    const uint64_t* s = reinterpret_cast<const uint64_t*>(p);
    return fp_from_u64x2(__hip_atomic_load(s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), __hip_atomic_load(s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}
