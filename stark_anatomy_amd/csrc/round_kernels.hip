// round_kernels.hip — kernels of the FRI commit's serial chain root_r -> challenge_r -> fold.
//
// The host enqueues the kernels of ALL rounds before the first root exists.  A round that needs the challenge of the
// round before it starts with k_wait_c0: one lane polls a sequence word in pinned host memory until the host — which
// still owns the transcript (proof_stream.rs:25-27, 51-56) — has published that round's fold constants there, and
// copies them into device memory for the kernels that follow on the stream.  The root -> host -> challenge hand-off
// then costs two PCIe hops and a SHA-256, not a stream synchronise plus two or three kernel launches.
//
// Every spin is bounded: the waiting lane gives up after `timeout_ticks` of the 100 MHz wall clock, or as soon as the host
// publishes an abort value (transcript callback failed), and raises the device abort word so that the remaining rounds'
// wait kernels return at once.
#include "kernels.h"
#include "quad_b2b.cuh"
#include "chain_wait.cuh"

// the device's 100 MHz wall clock into pinned host memory: pairs it with the host's clock (debug runs only)
__global__ void k_clock_probe(uint64_t* h_out) { __hip_atomic_store(h_out, (uint64_t)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
hipError_t launch_clock_probe(uint64_t* h_out, hipStream_t s) {
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(1), 0, s, h_out);
    return hipGetLastError();
}
// h_seq: pinned host word.  value & 0x7fffffff == gen means "published"; bit 31 = abort.
__global__ __launch_bounds__(1024) void k_wait_c0(const uint32_t* __restrict__ h_seq, uint32_t gen, const fp* __restrict__ h_c0, fp* __restrict__ d_c0,
                                                  uint32_t B, uint32_t* __restrict__ d_abort, uint32_t* __restrict__ h_abort, uint64_t timeout_ticks) {
    __shared__ uint32_t s_state;
    const ChainWait w{h_seq, gen, d_abort, h_abort, timeout_ticks};
    if (!chain_wait_wg(w, &s_state, false)) return;
    for (uint32_t b = threadIdx.x; b < B; b += blockDim.x) {
        const uint64_t* src = reinterpret_cast<const uint64_t*>(h_c0 + b);
        const uint64_t lo = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const uint64_t hi = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        d_c0[b] = fp_from_u64x2(lo, hi);
    }
}

// B digests that are already roots (trees of one subtree digest) -> the round's record in pinned host memory, then seq.
__global__ __launch_bounds__(256) void k_publish_digests(const digest64* __restrict__ in, uint32_t B, uint64_t* __restrict__ h_roots, uint32_t* __restrict__ h_seq,
                                                         uint32_t seq, const uint32_t* __restrict__ d_abort) {
    if (d_abort && __hip_atomic_load(d_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    for (uint32_t i = threadIdx.x; i < B * 8; i += blockDim.x)
        __hip_atomic_store(h_roots + i, in[i >> 3].h[i & 7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(h_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---------------------------------------------------------------------------------------------------------------------
// k_round_wg — one whole small FRI round per launch (outputs B·half <= 2^15): wait for the challenge, fold, store the
// folded codeword, hash it and build the bottom min(7, log2 half) Merkle levels in LDS, four lanes per compression
// throughout (leaves too: a leaf is the same compression over a zero-padded 16-byte message).  A workgroup covers 128
// consecutive outputs of the B folded codewords laid back to back, so every level of its tree is ONE compression deep:
// the round costs (1 + levels) dependent compressions, ~1.4 us each, instead of three or four launches.
// If a tree fits its workgroup (half <= 128) the roots go straight to the host record; else the workgroup's sub-root
// goes to `sub_out` and k_quad_reduce finishes the <= 9 remaining levels.  FOLD = false is round 0: leaves = cw itself.
struct RoundArgs {
    const fp* cw; fp* out;
    uint32_t log_half, total, B;                // outputs per codeword (log2); outputs over all codewords
    const uint32_t* h_seq; uint32_t gen; const fp* h_c0; const fp* d_c0;   // h_seq != null: wait in the kernel (one codeword); else c0 = d_c0[b]
    uint32_t* d_abort; uint32_t* h_abort; uint64_t timeout_ticks;
    const fp* tw_inv; uint32_t log_n_table, tw_shift;
    digest64* sub_out;                         // per-workgroup results when the trees continue above the workgroup
    uint64_t* rec_roots; uint32_t* rec_seq; uint32_t* counter; uint32_t target;   // host record when they end in it
};
#define RW_THREADS 512
#define RW_CHUNK_LOG 7
#define RW_CHUNK (1u << RW_CHUNK_LOG)
template <bool FOLD>
__global__ __launch_bounds__(RW_THREADS) void k_round_wg(const RoundArgs a) {
    __shared__ __align__(16) uint64_t sm[QR_LDS_U64];
    __shared__ uint32_t s_state, s_sig[48];
    __shared__ fp s_c0;
    const uint32_t tid = threadIdx.x;
    const uint32_t j0 = blockIdx.x * RW_CHUNK;
    const uint32_t active = a.total - j0 < RW_CHUNK ? a.total - j0 : RW_CHUNK;
    // while lane 0 waits: the sigma table, and zeros behind the 16 message bytes of every leaf slot (slot i = pair slot i of
    // the level buffer at offset 0: 144 bytes, the value in words 0-1)
    if (tid >= 64 && tid < 112) s_sig[tid - 64] = (&QSIG[0][0])[tid - 64];
    for (uint32_t i = tid; i < RW_CHUNK * 16; i += RW_THREADS) sm[(i >> 4) * QR_PAIR_U64 + 2 + (i & 15)] = 0;
    // Everything of the fold that does not need the challenge happens BEFORE the wait: the loads of the two inputs and of the twiddle
    // (L2 / HBM latency), (x + y)/2 and (x - y)·w^-i.  out = (x + y)/2 + ((x - y)·w^-i)·c0 — exact modular arithmetic, so the order of the
    // two multiplies does not change a bit of the result.
    fp pre_s = fp_zero(), pre_u = fp_zero();
    if (FOLD && tid < active) {
        const uint32_t j = j0 + tid, half = 1u << a.log_half, b = j >> a.log_half, w = j & (half - 1);
        const fp* lo = a.cw + ((size_t)b << (a.log_half + 1)) + w;
        fp tw = a.tw_inv[((size_t)w << a.tw_shift) & ((1u << NTT_TW_LO_BITS) - 1)];
        if (a.log_n_table > NTT_TW_LO_BITS) tw = mont_mul(tw, a.tw_inv[(1u << NTT_TW_LO_BITS) + (((size_t)w << a.tw_shift) >> NTT_TW_LO_BITS)]);
        const fp x = lo[0], y = lo[half];
        pre_s = fp_half(fp_add(x, y));
        pre_u = mont_mul(fp_sub(x, y), tw);
    }
    if (FOLD && a.h_seq) {                                                   // every workgroup waits for the challenge itself (chain_wait.cuh)
        const ChainWait w{a.h_seq, a.gen, a.d_abort, a.h_abort, a.timeout_ticks};
        if (tid == 0) {
            const uint32_t st = chain_wait_poll(w, true);
            if (!st && a.B == 1) s_c0 = load_fp_sys(a.h_c0);                     // one codeword: one lane fetches the constant
            s_state = st;
        }
        __syncthreads();
        if (s_state) return;
    } else if (FOLD) {
        if (__hip_atomic_load(a.d_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;      // raised by the k_wait_c0 in front of this launch
    }
    if (tid < active) {
        const uint32_t j = j0 + tid;
        fp v;
        if (FOLD) {
            const uint32_t b = j >> a.log_half;
            const fp c0 = a.h_seq ? (a.B == 1 ? s_c0 : load_fp_sys(a.h_c0 + b)) : a.d_c0[b];
            v = fp_add(pre_s, mont_mul(pre_u, c0));                                            // G4, as fold_one (hash_kernels.hip)
            a.out[j] = v;
        } else v = a.cw[j];
        sm[tid * QR_PAIR_U64] = fp_lo64(v);
        sm[tid * QR_PAIR_U64 + 1] = fp_hi64(v);
    }
    __syncthreads();
    const QuadLane L = quad_lane_init(tid & 3, s_sig);
    const uint32_t q = tid & 3, quad = tid >> 2;
    const uint32_t D = 256 * QR_PAIR_U64;                                           // leaf digests -> the second level buffer
    if (quad < active) {                                                            // leaf = Blake2b512(value.to_le_bytes())   [SURVEY.md §8c G2]
        uint64_t lo, hi;
        quad_compress(reinterpret_cast<const unsigned char*>(sm + quad * QR_PAIR_U64), L, L.d0_leaf, lo, hi);
        uint64_t* o = sm + D + QR_DIGEST_OFF(quad);
        o[q] = lo;
        o[4 + q] = hi;
    }
    __syncthreads();
    const uint32_t levels = a.log_half < RW_CHUNK_LOG ? a.log_half : RW_CHUNK_LOG;  // active is a multiple of 2^levels (launcher)
    const uint32_t s_off = quad_tree_levels(sm, D, 0, active, levels, tid, RW_THREADS, L);
    const uint32_t nout = active >> levels;
    if (levels == a.log_half && (gridDim.x == 1 || a.counter))
        quad_publish_roots(sm, s_off, nout, j0 >> a.log_half, a.rec_roots, a.rec_seq, a.gen, gridDim.x > 1 ? a.counter : nullptr, a.target, tid, RW_THREADS);
    else                                           // sub-roots for the reduction above — or finished roots (several workgroups, no counter) for k_publish_digests
        for (uint32_t i = tid; i < nout * 8; i += RW_THREADS) a.sub_out[(size_t)blockIdx.x * nout + (i >> 3)].h[i & 7] = sm[s_off + QR_DIGEST_OFF(i >> 3) + (i & 7)];
}

uint32_t round_wg_max_log() {
    static const uint32_t v = [] { const char* e = getenv("STARK_HIP_ROUND_WG_MAX_LOG"); const int x = e ? atoi(e) : 0; return (uint32_t)(x >= 7 && x <= ROUND_WG_MAX_LOG_LIMIT ? x : ROUND_WG_MAX_LOG_TOTAL); }();
    return v;
}
// Returns hipErrorInvalidValue unless B·2^log_half <= 2^round_wg_max_log() (2^15 by default).
hipError_t launch_round_wg(bool fold, const fp* cw, fp* out, uint32_t log_half, size_t B, const uint32_t* h_seq, uint32_t gen, const fp* h_c0, const fp* d_c0, uint32_t* d_abort,
                           uint32_t* h_abort, uint64_t timeout_ticks, const fp* tw_inv, uint32_t log_n_table, uint32_t tw_shift, digest64* sub_out,
                           uint64_t* rec_roots, uint32_t* rec_seq, uint32_t* counter, uint32_t target, hipStream_t s) {
    const size_t total = B << log_half;
    if (!B || total > ((size_t)1 << round_wg_max_log()) || log_half == 0 || log_half > round_wg_max_log()) return hipErrorInvalidValue;
    if (log_half > RW_CHUNK_LOG && !sub_out) return hipErrorInvalidValue;
    if (log_half <= RW_CHUNK_LOG && total > RW_CHUNK && !counter && !sub_out) return hipErrorInvalidValue;    // roots of several workgroups need one of the two ways out
    const unsigned blocks = (unsigned)((total + RW_CHUNK - 1) / RW_CHUNK);
    // h_seq: lane 0 of EVERY workgroup polls it.  Fine for a record on the card; for a record in pinned host memory every poll is a
    // PCIe read, so the caller keeps that form to single-workgroup launches (api_fri.hip).
    if (fold && h_seq && !h_c0) return hipErrorInvalidValue;
    if (fold && !h_seq && !d_c0) return hipErrorInvalidValue;
    RoundArgs a{cw, out, log_half, (uint32_t)total, (uint32_t)B, h_seq, gen, h_c0, d_c0, d_abort, h_abort, timeout_ticks, tw_inv, log_n_table, tw_shift, sub_out, rec_roots, rec_seq, counter, target};
    if (fold) hipLaunchKernelGGL(k_round_wg<true>, dim3(blocks), dim3(RW_THREADS), 0, s, a);
    else hipLaunchKernelGGL(k_round_wg<false>, dim3(blocks), dim3(RW_THREADS), 0, s, a);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_copy_to_host(const fp* __restrict__ src, fp* __restrict__ h_dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) h_dst[i] = src[i];
}

hipError_t launch_copy_to_host(const fp* src, fp* h_dst, size_t n, hipStream_t s) {
    if (!n) return hipSuccess;
    size_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_copy_to_host, dim3((unsigned)blocks), dim3(256), 0, s, src, h_dst, n);
    return hipGetLastError();
}
hipError_t launch_wait_c0(const uint32_t* h_seq, uint32_t gen, const fp* h_c0, fp* d_c0, size_t B, uint32_t* d_abort, uint32_t* h_abort, uint64_t timeout_ticks,
                          hipStream_t s) {
    if (!B || B > 0xffffffffu) return hipErrorInvalidValue;
    const unsigned threads = B >= 1024 ? 1024u : (unsigned)((B + 63) & ~(size_t)63);
    hipLaunchKernelGGL(k_wait_c0, dim3(1), dim3(threads), 0, s, h_seq, gen, h_c0, d_c0, (uint32_t)B, d_abort, h_abort, timeout_ticks);
    return hipGetLastError();
}
hipError_t launch_publish_digests(const digest64* in, size_t B, uint64_t* h_roots, uint32_t* h_seq, uint32_t seq, hipStream_t s, const uint32_t* d_abort) {
    if (!B || B > (1u << 24)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_publish_digests, dim3(1), dim3(256), 0, s, in, (uint32_t)B, h_roots, h_seq, seq, d_abort);
    return hipGetLastError();
}
