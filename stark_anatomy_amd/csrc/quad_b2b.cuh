// quad_b2b.cuh — one Blake2b-512 compression shared by FOUR lanes (a "quad"): the latency form used wherever a Merkle
// level has fewer nodes than the chip has lanes (merkle.rs:48-60 top levels; the small FRI rounds).
//
// Lane q of a quad owns column q of the 4x4 state (a = v[q], b = v[4+q], c = v[8+q], d = v[12+q]); the four column
// G's run in parallel, the diagonal step rotates b/c/d by 1/2/3 lanes with DPP quad_perm moves.  The 128-byte message
// (two child digests, or a zero-padded leaf value) is read from LDS with per-lane sigma offsets.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "blake2b.cuh"

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

// Level buffers hold digest PAIRS (= one node's 128-byte message) at a stride of 144 bytes: with the natural 128 the
// same message word of all 16 quads of a wave falls in one LDS bank; 16 bytes of padding rotate consecutive pairs by
// 4 banks and keep 16-byte alignment.
#define QR_PAIR_U64 18
#define QR_PAIR_BYTES (QR_PAIR_U64 * 8)
#define QR_DIGEST_OFF(d) (((d) >> 1) * QR_PAIR_U64 + ((d) & 1) * 8)      // u64 index of digest d in a level buffer

// Per-lane constants of a quad: initial state / feed-forward words and the packed sigma offsets.
struct QuadLane {
    uint64_t a0, b0, c0, d0_node, d0_leaf;
    uint32_t sig[12];
};
// `sigtab`: the 4 x 12 table above, or a copy of it in LDS
__device__ __forceinline__ QuadLane quad_lane_init(uint32_t q, const uint32_t* sigtab = &QSIG[0][0]) {
    QuadLane L;
#pragma unroll
    for (int r = 0; r < 12; ++r) L.sig[r] = sigtab[q * 12 + r];
    const uint64_t iv_lo = q == 0 ? B2B_IV0 : q == 1 ? B2B_IV1 : q == 2 ? B2B_IV2 : B2B_IV3;
    const uint64_t iv_hi = q == 0 ? B2B_IV4 : q == 1 ? B2B_IV5 : q == 2 ? B2B_IV6 : B2B_IV7;
    L.a0 = q == 0 ? B2B_H0 : iv_lo;
    L.b0 = iv_hi;
    L.c0 = iv_lo;
    L.d0_node = q == 0 ? (iv_hi ^ 128ULL) : q == 2 ? ~iv_hi : iv_hi;     // t = 128, final block   [merkle.rs:52-54]
    L.d0_leaf = q == 0 ? (iv_hi ^ 16ULL) : q == 2 ? ~iv_hi : iv_hi;      // t = 16, final block    [SURVEY.md §8c G2]
    return L;
}

// One compression of the 128-byte message at LDS address `m` by the calling quad; lane q returns words q and 4+q of the
// digest.  d0 = L.d0_node for H(left || right), L.d0_leaf for a leaf value zero-padded to 128 bytes.
__device__ __forceinline__ void quad_compress(const unsigned char* m, const QuadLane& L, uint64_t d0, uint64_t& out_lo, uint64_t& out_hi) {
    uint64_t a = L.a0, b = L.b0, c = L.c0, d = d0;
    uint64_t mw[12][4];                                          // all 48 message words up front: LDS latency paid once
#pragma unroll
    for (int r = 0; r < 12; ++r) {
        const uint32_t sg = L.sig[r];
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
    out_lo = L.a0 ^ a ^ c;
    out_hi = L.b0 ^ b ^ d;
}

// `levels` levels of a forest whose n_in digests (a multiple of 2^levels) sit in the level buffer at u64 offset s_off of `sm`
// (digest d at QR_DIGEST_OFF(d)), ping-ponging with the buffer at d_off; all `nthreads` threads of the workgroup call it.
// Returns the u64 offset of the buffer that holds the n_in >> levels results (again at QR_DIGEST_OFF).  sm holds
// (256 + 128) * QR_PAIR_U64 words: a buffer of 256 pairs at offset 0 and one of 128 pairs at 256 * QR_PAIR_U64, so
// n_in <= 512 starting at 0, n_in <= 256 starting in the second buffer.
#define QR_LDS_U64 ((256 + 128) * QR_PAIR_U64)
__device__ __forceinline__ uint32_t quad_tree_levels(uint64_t* sm, uint32_t s_off, uint32_t d_off, uint32_t n_in, uint32_t levels, uint32_t tid, uint32_t nthreads,
                                                     const QuadLane& L) {
    const uint32_t q = tid & 3, quad = tid >> 2, nquads = nthreads >> 2;
    uint32_t n = n_in >> 1;
#pragma unroll 1
    for (uint32_t lv = 0; lv < levels; ++lv, n >>= 1) {
#pragma unroll 1
        for (uint32_t node = quad; node < n; node += nquads) {
            uint64_t lo, hi;
            quad_compress(reinterpret_cast<const unsigned char*>(sm + s_off + node * QR_PAIR_U64), L, L.d0_node, lo, hi);
            uint64_t* o = sm + d_off + QR_DIGEST_OFF(node);
            o[q] = lo;
            o[4 + q] = hi;
        }
        __syncthreads();
        const uint32_t t = s_off; s_off = d_off; d_off = t;
    }
    return s_off;
}

// Hand `nroots` finished roots (in sm at s_off, QR_DIGEST_OFF layout) to the host: system-scope stores into the round's
// pinned record at root index root0 + i, then the sequence word — directly when the launch is a single workgroup, else through
// `counter` (only ever grows; `target` is the running total that this launch completes): the workgroup that brings it to
// `target` publishes.  All threads of the workgroup call it.
__device__ __forceinline__ void quad_publish_roots(const uint64_t* sm, uint32_t s_off, uint32_t nroots, size_t root0, uint64_t* __restrict__ rec_roots,
                                                   uint32_t* __restrict__ rec_seq, uint32_t seq, uint32_t* __restrict__ counter, uint32_t target, uint32_t tid, uint32_t nthreads) {
    for (uint32_t i = tid; i < nroots * 8; i += nthreads)
        __hip_atomic_store(rec_roots + (root0 << 3) + i, sm[s_off + QR_DIGEST_OFF(i >> 3) + (i & 7)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
        bool last = true;
        if (counter) last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_SYSTEM) + 1u == target;
        if (last) {
            __threadfence_system();
            __hip_atomic_store(rec_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
