// quad_kernels.hip — latency-optimised top of the Merkle tree (merkle.rs:48-60, the levels where there are
// fewer nodes than lanes on the chip).
//
// Why.  One Blake2b compression is a ~2.3k-instruction dependent stream; a lone wave issues one VALU
// instruction every ~5 cycles, so a level of the tree costs >= 4.5 us however few nodes it has, and a FRI
// commit walks sum_r log2(L_r) ~ 200-300 such levels (the root -> challenge -> fold chain is
// strictly serial).  At 2^20 that chain, not throughput, is the whole run time.
//
// How.  FOUR lanes share one compression: lane q of a quad owns column q of the 4x4 state (a=v[q], b=v[4+q],
// c=v[8+q], d=v[12+q]), so the four column G's — and, after rotating b/c/d by 1/2/3 lanes with DPP
// quad_perm moves (no LDS, no extra latency), the four diagonal G's — run in parallel: ~0.9k instructions
// per lane per compression instead of ~2.3k.  The message (the two child digests) is read straight out of
// the LDS level buffer with per-lane sigma offsets.  Levels live in LDS; one workgroup reduces a chunk of
// up to 512 nodes to one.  Per node this costs ~1.5x the lane-instructions of the one-lane form, so it is
// used only where lanes are idle anyway (count <= 2^17); the wide levels use hash_kernels.hip.
#include "kernels.h"
#include "quad_b2b.cuh"

#define QR_THREADS 512

// out[blockIdx] = Merkle root of in[blockIdx << log_chunk .. +2^log_chunk), 1 <= log_chunk <= 9.
// If `mailbox` is set (single-workgroup launch), the root and then the sequence number are also written to
// pinned host memory so the host can pick the root up by polling instead of a stream synchronise.
// Multi-workgroup form of the same hand-off (batched commits: one root per workgroup): `out` is then the device view of
// pinned host memory, every workgroup stores its root there with system-scope stores, bumps `counter`, and the
// workgroup that brings it to `target` publishes `seq`.  (`counter` only ever grows; the host passes the running total.)
__global__ __launch_bounds__(QR_THREADS) void k_quad_reduce(const digest64* __restrict__ in, digest64* __restrict__ out, uint32_t log_chunk,
                                                            uint64_t* __restrict__ mailbox, uint32_t seq, uint32_t* __restrict__ counter, uint32_t target,
                                                            const uint32_t* __restrict__ d_abort) {
    __shared__ __align__(16) uint64_t sm[QR_LDS_U64];
    if (d_abort && __hip_atomic_load(d_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;     // the round was abandoned upstream: publish nothing
    const uint32_t tid = threadIdx.x;
    const uint32_t chunk = 1u << log_chunk;
    const QuadLane L = quad_lane_init(tid & 3);
    {
        const uint4* src = reinterpret_cast<const uint4*>(in + ((size_t)blockIdx.x << log_chunk));
        uint4* dst = reinterpret_cast<uint4*>(sm);
        for (uint32_t i = tid; i < chunk * 4; i += QR_THREADS) dst[(QR_DIGEST_OFF(i >> 2) >> 1) + (i & 3)] = src[i];
    }
    __syncthreads();
    const uint32_t s_off = quad_tree_levels(sm, 0, 256 * QR_PAIR_U64, chunk, log_chunk, tid, QR_THREADS, L);
    if (counter) {
        if (tid < 8) __hip_atomic_store(&out[blockIdx.x].h[tid], sm[s_off + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
        __syncthreads();
        if (tid == 0) {
            const uint32_t done = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_SYSTEM) + 1u;
            if (done == target) {
                __threadfence_system();
                __hip_atomic_store(reinterpret_cast<uint32_t*>(mailbox + 8), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        return;
    }
    if (tid < 8) {
        const uint64_t w = sm[s_off + tid];
        out[blockIdx.x].h[tid] = w;
        if (mailbox) {
            __hip_atomic_store(mailbox + tid, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __threadfence_system();
            if (tid == 0) __hip_atomic_store(reinterpret_cast<uint32_t*>(mailbox + 8), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

hipError_t launch_quad_reduce(const digest64* in, digest64* out, size_t n_in, uint32_t log_chunk, uint64_t* mailbox, uint32_t seq, hipStream_t s,
                              uint32_t* counter, uint32_t target, const uint32_t* d_abort) {
    if (log_chunk < 1 || log_chunk > 9 || n_in == 0 || (n_in & ((((size_t)1) << log_chunk) - 1))) return hipErrorInvalidValue;
    const size_t blocks = n_in >> log_chunk;
    if (mailbox && !counter && blocks != 1) return hipErrorInvalidValue;
    if (counter && !mailbox) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_quad_reduce, dim3((unsigned)blocks), dim3(QR_THREADS), 0, s, in, out, log_chunk, mailbox, seq, counter, target, d_abort);
    return hipGetLastError();
}
