// hash_kernels.hip — Blake2b-512 leaf hashing, Merkle construction and the fused FRI round on gfx950.
//
// Replaces MerkleTree::commit's strictly sequential loop (merkle.rs:48-60: one thread, L-1
// compressions) and the per-element field divisions of a naive split-and-fold (field.rs:97-114).
//
// Structure.  Blake2b is 64-bit add/xor/rotate work on a 32-bit VALU (~2.3k VALU ops per
// compression) — the commit is integer-ALU bound, not HBM bound.  What matters is
// keeping every lane busy: a level-by-level tree inside a wave/workgroup idles half the lanes per
// level.  So each LANE owns a whole 2^K-leaf subtree and walks it depth-first with a K-deep stack
// of pending left siblings held in registers: 2^(K+1)-1 compressions per lane, no LDS, no
// barriers, no idle lanes.  Each kernel contains exactly one unrolled leaf compression and one
// unrolled node compression (the loops around them are rolled) so the instruction stream stays
// inside the instruction cache.  Only the last <= 512 nodes are reduced level by level in LDS by
// a single workgroup (k_merkle_tail).
#include <stdlib.h>
#include "kernels.h"
#include "chain_wait.cuh"

#define SUBTREE_KMAX 4
#ifndef SUBTREE_WAVES
#define SUBTREE_WAVES 1
#endif

// Depth-first reduction of 2^K items produced by `leaf(i)` (i ascending) to one digest, K <= KMAX <= 4.
// Every loop bound and stack index is wave-uniform.  The stack of pending left siblings is KMAX digests of 16 VGPRs each: a
// kernel whose K is fixed says so (KMAX = K) and does not carry the registers of levels it never reaches.
template <int KMAX = SUBTREE_KMAX, class LeafFn>
__device__ __forceinline__ digest64 subtree_reduce(uint32_t K, LeafFn leaf) {
    static_assert(KMAX >= 1 && KMAX <= 4, "subtree stack depth");
    digest64 lv0, lv1, lv2, lv3, cur;
    const uint32_t count = 1u << K;
#pragma unroll 1
    for (uint32_t i = 0; i < count; ++i) {
        cur = leaf(i);
        uint32_t j = 0;
#pragma unroll 1
        for (; (i >> j) & 1u; ++j) {
            digest64 left;
            if (j == 0 || KMAX == 1) left = lv0;
            else if (j == 1 || KMAX == 2) left = lv1;
            else if (j == 2 || KMAX == 3) left = lv2;
            else left = lv3;
            cur = b2b_node(left, cur);                   // merkle.rs:52-54: H(left || right)
        }
        if (j < K) {                                     // j == K: the subtree's root, nothing waits for it
            if (j == 0 || KMAX == 1) lv0 = cur;
            else if (j == 1 || KMAX == 2) lv1 = cur;
            else if (j == 2 || KMAX == 3) lv2 = cur;
            else lv3 = cur;
        }
    }
    return cur;
}

__device__ __forceinline__ void store_digest(digest64* __restrict__ p, const digest64& d) {
    uint4* q = reinterpret_cast<uint4*>(p);
    const uint32_t* w = reinterpret_cast<const uint32_t*>(d.h);
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
__device__ __forceinline__ digest64 load_digest(const digest64* __restrict__ p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    digest64 d;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint4 v = q[i];
        d.h[2 * i] = (uint64_t)v.x | ((uint64_t)v.y << 32);
        d.h[2 * i + 1] = (uint64_t)v.z | ((uint64_t)v.w << 32);
    }
    return d;
}

__global__ __launch_bounds__(256) void k_hash_leaves(const fp* __restrict__ elems, digest64* __restrict__ out, size_t L) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
#pragma unroll 1
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < L; i += stride) {
        fp e = elems[i];
        store_digest(out + i, b2b_leaf(fp_lo64(e), fp_hi64(e)));
    }
}

// KMAX = 3 for K <= 3 (every launch of the commit), 4 only for the K = 4 that sh_dev_merkle_subtrees also accepts: one stack level
// fewer is 16 VGPRs fewer, and with <= 128 a SIMD holds four of these waves instead of three.
template <int KMAX>
__global__ __launch_bounds__(256, SUBTREE_WAVES) void k_leaf_subtree(const fp* __restrict__ elems, digest64* __restrict__ out, size_t n_out, uint32_t K) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    const fp* base = elems + (i << K);
    digest64 d = subtree_reduce<KMAX>(K, [&](uint32_t q) {
        fp e = base[q];
        return b2b_leaf(fp_lo64(e), fp_hi64(e));
    });
    store_digest(out + i, d);
}

template <int KMAX>
__global__ __launch_bounds__(256, SUBTREE_WAVES) void k_node_subtree(const digest64* __restrict__ in, digest64* __restrict__ out, size_t n_out, uint32_t K) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    const digest64* base = in + (i << K);
    digest64 d = subtree_reduce<KMAX>(K, [&](uint32_t q) { return load_digest(base + q); });
    store_digest(out + i, d);
}

// Top of the tree: n_in <= 512 nodes -> root, built level by level in LDS by one workgroup.
__global__ __launch_bounds__(256) void k_merkle_tail(const digest64* __restrict__ in, digest64* __restrict__ root, uint32_t n_in) {
    __shared__ digest64 sm[MERKLE_TAIL_MAX];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < n_in; i += blockDim.x) sm[i] = load_digest(in + i);
    __syncthreads();
#pragma unroll 1
    for (uint32_t n = n_in; n > 1; n >>= 1) {
        const uint32_t half = n >> 1;
        digest64 r;
        const bool act = tid < half;
        if (act) r = b2b_node(sm[2 * tid], sm[2 * tid + 1]);
        __syncthreads();
        if (act) sm[tid] = r;
        __syncthreads();
    }
    if (tid == 0) store_digest(root, sm[0]);
}

// ---------------------------------------------------------------------------------------------
// FRI split-and-fold:
//   out[i] = 2^-1·((1 + a/(off·w^i))·x + (1 − a/(off·w^i))·y),  x = cw[i], y = cw[L/2+i]
//          = (x + y)/2 + (a·2^-1·off^-1)·w^-i·(x − y)            (same canonical residue)
// One Montgomery multiply per output plus the twiddle; no per-element division (field.rs:97-114).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ fp fold_one(fp x, fp y, fp t_mont) {     // t_mont = c0·w^-i in Montgomery form
    return fp_add(fp_half(fp_add(x, y)), mont_mul(fp_sub(x, y), t_mont));
}

#define FOLD_PER_THREAD 4
// `lo`/`hi` are the two halves (separate pointers: a rank of the distributed commit holds a slice of each).
// c0_dev != null: the fold constant comes from device memory (a wait kernel put it there) instead of `c0_mont`; i_base: global
// index of lo[0] in the round's codeword (a rank's slice of a sharded commit); d_abort as in kernels.h.
__global__ __launch_bounds__(256) void k_fri_fold(const fp* __restrict__ lo, const fp* __restrict__ hi, size_t half, fp c0_mont, fp omega_inv,
                                                  fp* __restrict__ out, const fp* __restrict__ c0_dev, size_t i_base, const uint32_t* __restrict__ d_abort) {
    if (d_abort && __hip_atomic_load(d_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    const size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * FOLD_PER_THREAD;
    if (i0 >= half) return;
    // t = c0 · omega_inv^(i_base + i0) (Montgomery form), by square-and-multiply; then stepped by omega_inv
    fp step = fp_to_mont(omega_inv), base = step, t = c0_dev ? *c0_dev : c0_mont;
    for (size_t e = i_base + i0; e; e >>= 1) {
        if (e & 1) t = mont_mul(t, base);
        base = mont_mul(base, base);
    }
#pragma unroll 1
    for (int q = 0; q < FOLD_PER_THREAD && i0 + q < half; ++q) {
        out[i0 + q] = fold_one(lo[i0 + q], hi[i0 + q], t);
        t = mont_mul(t, step);
    }
}

__device__ __forceinline__ fp load_tw(const fp* __restrict__ tw, uint32_t log_n, size_t e) {
    // tw = [lo: w^i, i < min(n,4096) | hi: w^(4096·i)], Montgomery form
    fp lo = tw[e & ((1u << NTT_TW_LO_BITS) - 1)];
    if (log_n <= NTT_TW_LO_BITS) return lo;
    fp hi = tw[(1u << NTT_TW_LO_BITS) + (e >> NTT_TW_LO_BITS)];
    return mont_mul(lo, hi);
}

// Fused FRI round: fold, write the folded codeword, and hash it into 2^K-leaf subtree digests.
// Batched form: B codewords of length 2*half back to back; lane i owns outputs [i<<K, (i+1)<<K) of the
// concatenated folded codewords (2^K <= half, so a lane never straddles two codewords).  The per-codeword
// constant c0 = alpha/(2*offset) comes from `c0_arr[b]` (device, Montgomery form) or, when null, `c0_single`.
template <int KMAX>
__global__ __launch_bounds__(256, SUBTREE_WAVES) void k_fold_leaf_subtree(const fp* __restrict__ cw, uint32_t log_half, const fp* __restrict__ c0_arr,
                                                           fp c0_single, const fp* __restrict__ tw_inv, uint32_t log_n_table, uint32_t tw_shift,
                                                           fp* __restrict__ out, digest64* __restrict__ digests, size_t n_out, uint32_t K,
                                                           const uint32_t* __restrict__ d_abort, size_t w_base, const ChainWait wait) {
    __shared__ uint32_t s_wait;
    if (d_abort && __hip_atomic_load(d_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t j0 = i << K, half = (size_t)1 << log_half;
    const size_t b = j0 >> log_half, w0 = j0 & (half - 1);
    const fp* lo = cw + (b << (log_half + 1)) + w0;
    const fp* hi = lo + half;
    // a kernel that waits for the challenge itself takes its first output as far as it goes without it before the wait: the loads
    // and (x + y)/2, (x - y)·w^-i are off the serial chain (exact arithmetic: the order of the two multiplies changes no bit)
    fp pre_s = fp_zero(), pre_u = fp_zero();
    if (wait.seq) {
        if (i < n_out) {
            const fp x = lo[0], y = hi[0];
            pre_s = fp_half(fp_add(x, y));
            pre_u = mont_mul(fp_sub(x, y), load_tw(tw_inv, log_n_table, (w_base + w0) << tw_shift));
        }
        if (!chain_wait_wg(wait, &s_wait, false)) return;
    }
    if (i >= n_out) return;
    const fp c0_mont = c0_arr ? (wait.seq ? load_fp_sys(c0_arr + b) : c0_arr[b]) : c0_single;
    digest64 d = subtree_reduce<KMAX>(K, [&](uint32_t q) {
        fp v;
        if (wait.seq && q == 0) v = fp_add(pre_s, mont_mul(pre_u, c0_mont));
        else {
            fp t = mont_mul(load_tw(tw_inv, log_n_table, (w_base + w0 + q) << tw_shift), c0_mont);
            v = fold_one(lo[q], hi[q], t);
        }
        out[j0 + q] = v;
        return b2b_leaf(fp_lo64(v), fp_hi64(v));
    });
    store_digest(digests + i, d);
}

// The same kernel with HBM access staged through LDS (K = 1, 2, 3).  In the plain form lane i reads lo[2^K·i+q], hi[2^K·i+q] and
// writes out[2^K·i+q] one q at a time, 2^K·16 bytes apart from its neighbours: every 128-byte line is touched up to 8 times over
// the ~100 us a subtree takes, the lines in flight per XCD (~6 MB) overflow its 4 MB L2, and rocprof shows 4x the algorithmic
// read traffic at K = 3 (2x for writes), 1.24x at K = 1, 2.  Here a wave first FOLDS its 64·2^K outputs in coalesced order —
// lane l takes outputs l, 64+l, ...: one coalesced 1 KB read of each half, the fold, one coalesced 1 KB write of the result per
// step — and parks the results in LDS (rows of 2^K padded by one so the later per-lane reads spread over the banks); then every
// lane hashes its own 2^K consecutive results out of LDS.  <= 9.2 KB of LDS per wave, so the kernel runs as many waves as its
// registers allow; the first staged form kept both input halves in LDS (18.4 KB per wave), ran 2 waves per SIMD and was ~10 %
// slower for it.
template <int K>
__global__ __launch_bounds__(256, 2) void k_fold_leaf_subtree_staged(const fp* __restrict__ cw, uint32_t log_half, const fp* __restrict__ c0_arr, fp c0_single,
                                                                     const fp* __restrict__ tw_inv, uint32_t log_n_table, uint32_t tw_shift,
                                                                     fp* __restrict__ out, digest64* __restrict__ digests, size_t n_out,
                                                                     const uint32_t* __restrict__ d_abort, size_t w_base, const ChainWait wait) {
    constexpr uint32_t ROW = (1u << K) + 1, WAVE_ELEMS = 64 * ROW;
    __shared__ fp s_v[4 * WAVE_ELEMS];
    __shared__ uint32_t s_wait;
    if (d_abort && __hip_atomic_load(d_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    if (wait.seq && !chain_wait_wg(wait, &s_wait, false)) return;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;          // n_out is a multiple of 256: whole workgroups are in or out
    if (i >= n_out) return;
    const size_t half = (size_t)1 << log_half;
    const size_t wj0 = (i - lane) << K;                                        // first output of this wave
    const size_t b = wj0 >> log_half, ww0 = wj0 & (half - 1);
    const fp c0_mont = c0_arr ? (wait.seq ? load_fp_sys(c0_arr + b) : c0_arr[b]) : c0_single;
    const fp* lo = cw + (b << (log_half + 1)) + ww0;
    const fp* hi = lo + half;
    fp* mine = s_v + wave * WAVE_ELEMS;
#pragma unroll 2
    for (uint32_t t = 0; t < (1u << K); ++t) {
        const uint32_t e = t * 64 + lane;
        const fp tw = mont_mul(load_tw(tw_inv, log_n_table, (w_base + ww0 + e) << tw_shift), c0_mont);
        const fp v = fold_one(lo[e], hi[e], tw);
        out[wj0 + e] = v;
        mine[e + (e >> K)] = v;
    }
    __syncthreads();
    digest64 d = subtree_reduce<K>(K, [&](uint32_t q) {
        const fp v = mine[lane * ROW + q];
        return b2b_leaf(fp_lo64(v), fp_hi64(v));
    });
    store_digest(digests + i, d);
}

// The same staging for the first round's leaf-only kernel: one coalesced read of the wave's 64·2^K elements instead of 2^K
// strided passes over them (rocprof at K = 3: 341 MB fetched per 268 MB of codeword in the plain form).
template <int K>
__global__ __launch_bounds__(256, SUBTREE_WAVES) void k_leaf_subtree_staged(const fp* __restrict__ elems, digest64* __restrict__ out, size_t n_out) {
    constexpr uint32_t ROW = (1u << K) + 1, WAVE_ELEMS = 64 * ROW;
    __shared__ fp s_el[4 * WAVE_ELEMS];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    const fp* base = elems + ((i - lane) << K);
    fp* mine = s_el + wave * WAVE_ELEMS;
#pragma unroll
    for (uint32_t t = 0; t < (1u << K); ++t) {
        const uint32_t e = t * 64 + lane;
        mine[e + (e >> K)] = base[e];
    }
    __syncthreads();
    digest64 d = subtree_reduce<K>(K, [&](uint32_t q) {
        fp e = mine[lane * ROW + q];
        return b2b_leaf(fp_lo64(e), fp_hi64(e));
    });
    store_digest(out + i, d);
}

// ---------------------------------------------------------------------------------------------
static inline unsigned grid_for(size_t n, int block) { return (unsigned)((n + block - 1) / block); }

hipError_t launch_hash_leaves(const fp* elems, digest64* out, size_t L, hipStream_t s) {
    if (!L) return hipSuccess;
    size_t blocks = (L + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_hash_leaves, dim3((unsigned)blocks), dim3(256), 0, s, elems, out, L);
    return hipGetLastError();
}
hipError_t launch_leaf_subtree(const fp* elems, digest64* out, size_t n_out, uint32_t K, hipStream_t s) {
    if (!n_out) return hipSuccess;
    if (K > SUBTREE_KMAX) return hipErrorInvalidValue;
    static const bool no_stage = getenv("STARK_NO_STAGE") != nullptr;
    if (K >= 1 && K <= 3 && (n_out & 255) == 0 && !no_stage) {
        const dim3 grid((unsigned)(n_out >> 8));
        if (K == 3) hipLaunchKernelGGL(k_leaf_subtree_staged<3>, grid, dim3(256), 0, s, elems, out, n_out);
        else if (K == 2) hipLaunchKernelGGL(k_leaf_subtree_staged<2>, grid, dim3(256), 0, s, elems, out, n_out);
        else hipLaunchKernelGGL(k_leaf_subtree_staged<1>, grid, dim3(256), 0, s, elems, out, n_out);
        return hipGetLastError();
    }
    if (K <= 3) hipLaunchKernelGGL(k_leaf_subtree<3>, dim3(grid_for(n_out, 256)), dim3(256), 0, s, elems, out, n_out, K);
    else hipLaunchKernelGGL(k_leaf_subtree<4>, dim3(grid_for(n_out, 256)), dim3(256), 0, s, elems, out, n_out, K);
    return hipGetLastError();
}
hipError_t launch_node_subtree(const digest64* in, digest64* out, size_t n_out, uint32_t K, hipStream_t s) {
    if (!n_out) return hipSuccess;
    if (K < 1 || K > SUBTREE_KMAX) return hipErrorInvalidValue;
    if (K <= 3) hipLaunchKernelGGL(k_node_subtree<3>, dim3(grid_for(n_out, 256)), dim3(256), 0, s, in, out, n_out, K);
    else hipLaunchKernelGGL(k_node_subtree<4>, dim3(grid_for(n_out, 256)), dim3(256), 0, s, in, out, n_out, K);
    return hipGetLastError();
}
hipError_t launch_merkle_tail(const digest64* in, digest64* root, size_t n_in, hipStream_t s) {
    if (n_in == 0 || n_in > MERKLE_TAIL_MAX || (n_in & (n_in - 1))) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_merkle_tail, dim3(1), dim3(256), 0, s, in, root, (uint32_t)n_in);
    return hipGetLastError();
}
hipError_t launch_fri_fold(const fp* lo, const fp* hi, size_t half, fp c0_mont, fp omega_inv, fp* out, hipStream_t s, const fp* c0_dev, size_t i_base,
                           const uint32_t* d_abort) {
    if (!half) return hipErrorInvalidValue;
    const size_t threads = (half + FOLD_PER_THREAD - 1) / FOLD_PER_THREAD;
    hipLaunchKernelGGL(k_fri_fold, dim3(grid_for(threads, 256)), dim3(256), 0, s, lo, hi, half, c0_mont, omega_inv, out, c0_dev, i_base, d_abort);
    return hipGetLastError();
}
hipError_t launch_fold_leaf_subtree(const fp* cw, size_t L, size_t B, const fp* c0_arr, fp c0_single, const fp* tw_inv, uint32_t log_n_table,
                                    uint32_t tw_shift, fp* out, digest64* digests, uint32_t K, hipStream_t s, const uint32_t* d_abort, size_t w_base, const ChainWait* wait) {
    const size_t half = L / 2;
    if (!half || (half & (half - 1)) || !B || K > SUBTREE_KMAX || (half >> K) == 0) return hipErrorInvalidValue;
    const ChainWait cwt = wait ? *wait : ChainWait{nullptr, 0, nullptr, nullptr, 0};
    if (cwt.seq && (!c0_arr || !cwt.d_abort || !cwt.h_abort)) return hipErrorInvalidValue;
    uint32_t log_half = 0;
    while (((size_t)1 << log_half) < half) ++log_half;
    const size_t n_out = (B * half) >> K;
    static const bool no_stage = getenv("STARK_NO_STAGE") != nullptr;
    if (K >= 1 && K <= 3 && (n_out & 255) == 0 && log_half >= 6 + K && !no_stage) {          // whole 256-lane workgroups, waves inside one codeword
        const dim3 grid((unsigned)(n_out >> 8));
        if (K == 3) hipLaunchKernelGGL(k_fold_leaf_subtree_staged<3>, grid, dim3(256), 0, s, cw, log_half, c0_arr, c0_single, tw_inv, log_n_table, tw_shift, out, digests, n_out, d_abort, w_base, cwt);
        else if (K == 2) hipLaunchKernelGGL(k_fold_leaf_subtree_staged<2>, grid, dim3(256), 0, s, cw, log_half, c0_arr, c0_single, tw_inv, log_n_table, tw_shift, out, digests, n_out, d_abort, w_base, cwt);
        else hipLaunchKernelGGL(k_fold_leaf_subtree_staged<1>, grid, dim3(256), 0, s, cw, log_half, c0_arr, c0_single, tw_inv, log_n_table, tw_shift, out, digests, n_out, d_abort, w_base, cwt);
        return hipGetLastError();
    }
    if (K <= 3) hipLaunchKernelGGL(k_fold_leaf_subtree<3>, dim3(grid_for(n_out, 256)), dim3(256), 0, s, cw, log_half, c0_arr, c0_single, tw_inv,
                                   log_n_table, tw_shift, out, digests, n_out, K, d_abort, w_base, cwt);
    else hipLaunchKernelGGL(k_fold_leaf_subtree<4>, dim3(grid_for(n_out, 256)), dim3(256), 0, s, cw, log_half, c0_arr, c0_single, tw_inv,
                            log_n_table, tw_shift, out, digests, n_out, K, d_abort, w_base, cwt);
    return hipGetLastError();
}
