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

// byte offsets (word index * 8) of the message words lane q needs in round r:
// [x_col, y_col, x_diag, y_diag] = sigma[r][2q], sigma[r][2q+1], sigma[r][8+2q], sigma[r][8+2q+1]
__device__ const uint32_t QSIG[4][12] = {
    {0x48400800u, 0x60085070u, 0x70504058u, 0x30104838u, 0x08700048u, 0x68206010u, 0x38002860u, 0x00285868u, 0x10607830u, 0x58781050u, 0x48400800u, 0x60085070u},
    {0x58501810u, 0x10004020u, 0x30180060u, 0x50280818u, 0x60583828u, 0x28385030u, 0x18307808u, 0x20787038u, 0x38684870u, 0x70482040u, 0x58501810u, 0x10004020u},
    {0x68602820u, 0x38587848u, 0x08381028u, 0x00206068u, 0x40302010u, 0x70785800u, 0x10486870u, 0x30400860u, 0x20081858u, 0x60183038u, 0x68602820u, 0x38587848u},
    {0x78703830u, 0x18283068u, 0x20486878u, 0x40787058u, 0x68187850u, 0x48081840u, 0x58405020u, 0x50104818u, 0x28504000u, 0x00682808u, 0x78703830u, 0x18283068u},
};

// lane q <- lane (q + S) & 3 of its quad
template <int S>
__device__ __forceinline__ uint64_t quad_rot(uint64_t v) {
    constexpr int ctrl = ((0 + S) & 3) | (((1 + S) & 3) << 2) | (((2 + S) & 3) << 4) | (((3 + S) & 3) << 6);
    int lo = (int)(uint32_t)v, hi = (int)(uint32_t)(v >> 32);
    lo = __builtin_amdgcn_mov_dpp(lo, ctrl, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, ctrl, 0xf, 0xf, true);
    return (uint64_t)(uint32_t)lo | ((uint64_t)(uint32_t)hi << 32);
}

// 64-bit rotate right as two v_alignbit_b32 on the halves (left to itself the compiler builds the rotates that feed
// DPP moves from 64-bit shifts and ORs: ~4 instructions each instead of 2, on a chain where every instruction is
// ~6 cycles of latency).
template <int N>
__device__ __forceinline__ uint64_t qrotr(uint64_t v) {
    const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    if (N == 32) return ((uint64_t)lo << 32) | hi;
    if (N < 32) {
        const uint32_t nl = __builtin_amdgcn_alignbit(hi, lo, N), nh = __builtin_amdgcn_alignbit(lo, hi, N);
        return ((uint64_t)nh << 32) | nl;
    }
    const uint32_t nl = __builtin_amdgcn_alignbit(lo, hi, N - 32), nh = __builtin_amdgcn_alignbit(hi, lo, N - 32);
    return ((uint64_t)nh << 32) | nl;
}

#define QG(a, b, c, d, x, y)                      \
    do {                                          \
        a = a + b + (x); d = qrotr<32>(d ^ a);    \
        c = c + d;       b = qrotr<24>(b ^ c);    \
        a = a + b + (y); d = qrotr<16>(d ^ a);    \
        c = c + d;       b = qrotr<63>(b ^ c);    \
    } while (0)

#define QR_PAIR_U64 18
#define QR_DIGEST_OFF(d) (((d) >> 1) * QR_PAIR_U64 + ((d) & 1) * 8)
#define QR_THREADS 512
#define QR_QUADS (QR_THREADS / 4)

// out[blockIdx] = Merkle root of in[blockIdx << log_chunk .. +2^log_chunk), 1 <= log_chunk <= 9.
// If `mailbox` is set (single-workgroup launch), the root and then the sequence number are also written to
// pinned host memory so the host can pick the root up by polling instead of a stream synchronise.
// Multi-workgroup form of the same hand-off (batched commits: one root per workgroup): `out` is then the device view of
// pinned host memory, every workgroup stores its root there with system-scope stores, bumps `counter`, and the
// workgroup that brings it to `target` publishes `seq`.  (`counter` only ever grows; the host passes the running total.)
__global__ __launch_bounds__(QR_THREADS) void k_quad_reduce(const digest64* __restrict__ in, digest64* __restrict__ out, uint32_t log_chunk,
                                                            uint64_t* __restrict__ mailbox, uint32_t seq, uint32_t* __restrict__ counter, uint32_t target) {
    // Level buffers hold digest PAIRS (= one node's 128-byte message) at a stride of 144 bytes: with the natural 128 the
    // same message word of all 16 quads of a wave falls in one LDS bank (rocprof: SQ_LDS_BANK_CONFLICT = 70 % of
    // LDS-active cycles); 16 bytes of padding rotate consecutive pairs by 4 banks and keep 16-byte alignment.
    __shared__ __align__(16) uint64_t sm[(256 + 128) * QR_PAIR_U64];
    const uint32_t tid = threadIdx.x, q = tid & 3, quad = tid >> 2;
    const uint32_t chunk = 1u << log_chunk;

    uint32_t sig[12];
#pragma unroll
    for (int r = 0; r < 12; ++r) sig[r] = QSIG[q][r];
    // per-lane initial state / feed-forward constants (t = 128, final block)
    const uint64_t iv_lo = q == 0 ? B2B_IV0 : q == 1 ? B2B_IV1 : q == 2 ? B2B_IV2 : B2B_IV3;
    const uint64_t iv_hi = q == 0 ? B2B_IV4 : q == 1 ? B2B_IV5 : q == 2 ? B2B_IV6 : B2B_IV7;
    const uint64_t a0 = q == 0 ? B2B_H0 : iv_lo, b0 = iv_hi, c0 = iv_lo;
    const uint64_t d0 = q == 0 ? (iv_hi ^ 128ULL) : q == 2 ? ~iv_hi : iv_hi;

    {
        const uint4* src = reinterpret_cast<const uint4*>(in + ((size_t)blockIdx.x << log_chunk));
        uint4* dst = reinterpret_cast<uint4*>(sm);
        for (uint32_t i = tid; i < chunk * 4; i += QR_THREADS) dst[(QR_DIGEST_OFF(i >> 2) >> 1) + (i & 3)] = src[i];
    }
    __syncthreads();

    uint32_t s_off = 0, d_off = 256 * QR_PAIR_U64;
#pragma unroll 1
    for (uint32_t n = chunk >> 1; n >= 1; n >>= 1) {
#pragma unroll 1
        for (uint32_t node = quad; node < n; node += QR_QUADS) {
            const unsigned char* m = reinterpret_cast<const unsigned char*>(sm + s_off + node * QR_PAIR_U64);
            uint64_t a = a0, b = b0, c = c0, d = d0;
            uint64_t mw[12][4];                                          // all 48 message words up front: LDS latency paid once
#pragma unroll
            for (int r = 0; r < 12; ++r) {
                const uint32_t sg = sig[r];
                mw[r][0] = *reinterpret_cast<const uint64_t*>(m + (sg & 0xff));
                mw[r][1] = *reinterpret_cast<const uint64_t*>(m + ((sg >> 8) & 0xff));
                mw[r][2] = *reinterpret_cast<const uint64_t*>(m + ((sg >> 16) & 0xff));
                mw[r][3] = *reinterpret_cast<const uint64_t*>(m + (sg >> 24));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 12; ++r) {
                QG(a, b, c, d, mw[r][0], mw[r][1]);                       // column step
                b = quad_rot<1>(b); c = quad_rot<2>(c); d = quad_rot<3>(d);
                QG(a, b, c, d, mw[r][2], mw[r][3]);                       // diagonal step
                b = quad_rot<3>(b); c = quad_rot<2>(c); d = quad_rot<1>(d);
            }
            uint64_t* o = sm + d_off + QR_DIGEST_OFF(node);
            o[q] = a0 ^ a ^ c;
            o[4 + q] = b0 ^ b ^ d;
        }
        __syncthreads();
        const uint32_t t = s_off; s_off = d_off; d_off = t;
    }
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
                              uint32_t* counter, uint32_t target) {
    if (log_chunk < 1 || log_chunk > 9 || n_in == 0 || (n_in & ((((size_t)1) << log_chunk) - 1))) return hipErrorInvalidValue;
    const size_t blocks = n_in >> log_chunk;
    if (mailbox && !counter && blocks != 1) return hipErrorInvalidValue;
    if (counter && !mailbox) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_quad_reduce, dim3((unsigned)blocks), dim3(QR_THREADS), 0, s, in, out, log_chunk, mailbox, seq, counter, target);
    return hipGetLastError();
}
