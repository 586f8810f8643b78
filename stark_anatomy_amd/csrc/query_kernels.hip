// query_kernels.hip — authentication paths for the FRI query phase (merkle.rs:70-108 `open`, for ~3t indices per tree of a
// proof at once), served from what the commit left on the device (CommitKeep, api_internal.h) instead of rebuilding trees.
//
// A commit's kernels write a digest array only every few levels (a lane's 2^K-leaf subtree, a workgroup's 2^9-digest
// chunk); the levels in between never reach memory.  A path needs the sibling at EVERY level.  The siblings between two
// arrays — levels lo .. hi-1 — all sit inside the one aligned block of 2^(hi-lo) level-lo nodes around the index, so a
// workgroup loads that block (or hashes the 2^hi codeword elements under it when there is no array below), rebuilds its
// levels in LDS four lanes per compression, and hands out one sibling per level as it goes.
// One workgroup per (tree, index, segment between arrays); all trees of a proof in one launch.
#include "kernels.h"
#include "quad_b2b.cuh"

#define OP_THREADS 512
__global__ __launch_bounds__(OP_THREADS) void k_open_paths(const OpenTree* __restrict__ trees, uint32_t n_trees, const uint64_t* __restrict__ idx,
                                                            digest64* __restrict__ paths, fp* __restrict__ vals) {
    __shared__ __align__(16) uint64_t sm[QR_LDS_U64];
    __shared__ uint32_t s_sig[48];
    const uint32_t tid = threadIdx.x;
    uint32_t t = 0;
    while (t + 1 < n_trees && trees[t + 1].wg_off <= blockIdx.x) ++t;
    const OpenTree T = trees[t];
    const uint32_t from_leaves = (T.n_stored == 0 || T.level[0] > 0) ? 1u : 0u, n_seg = T.n_stored + from_leaves;
    const uint32_t local = blockIdx.x - T.wg_off, j = local / n_seg, sg = local - j * n_seg;
    const uint64_t i = idx[T.idx_off + j];
    digest64* out = paths + T.path_off + (size_t)j * T.log_L;            // out[l] = sibling at level l (merkle.rs:104-107 order: leaf level first)
    if (tid < 48) s_sig[tid] = (&QSIG[0][0])[tid];
    uint32_t lo, hi, s_off, d_off;
    if (from_leaves && sg == 0) {                                        // 2^hi codeword elements -> leaf hashes = level 0
        lo = 0; hi = T.n_stored ? T.level[0] : T.log_L;
        const uint32_t n_in = 1u << hi;
        const uint64_t base = (i >> hi) << hi;
        if (tid == 0 && vals) vals[T.idx_off + j] = T.cw[i];
        for (uint32_t e = tid; e < n_in * 16; e += OP_THREADS) sm[(e >> 4) * QR_PAIR_U64 + 2 + (e & 15)] = 0;
        if (tid < n_in) {
            const fp v = T.cw[base + tid];
            sm[tid * QR_PAIR_U64] = fp_lo64(v);
            sm[tid * QR_PAIR_U64 + 1] = fp_hi64(v);
        }
        __syncthreads();
        const QuadLane L = quad_lane_init(tid & 3, s_sig);
        const uint32_t q = tid & 3, quad = tid >> 2;
        s_off = 256 * QR_PAIR_U64; d_off = 0;
        if (quad < n_in) {                                               // leaf = Blake2b512(value.to_le_bytes())   [SURVEY.md §8c G2]
            uint64_t dl, dh;
            quad_compress(reinterpret_cast<const unsigned char*>(sm + quad * QR_PAIR_U64), L, L.d0_leaf, dl, dh);
            uint64_t* o = sm + s_off + QR_DIGEST_OFF(quad);
            o[q] = dl;
            o[4 + q] = dh;
        }
        __syncthreads();
    } else {                                                             // the block of 2^(hi-lo) digests of the array at level lo
        const uint32_t k = sg - from_leaves;
        lo = T.level[k]; hi = (k + 1 < T.n_stored) ? T.level[k + 1] : T.log_L;
        const uint32_t n_in = 1u << (hi - lo);
        const digest64* src = T.stored[k] + ((i >> hi) << (hi - lo));
        if (!from_leaves && sg == 0 && tid == 0 && vals) vals[T.idx_off + j] = T.cw[i];
        s_off = 0; d_off = 256 * QR_PAIR_U64;
        for (uint32_t e = tid; e < n_in * 8; e += OP_THREADS) sm[QR_DIGEST_OFF(e >> 3) + (e & 7)] = src[e >> 3].h[e & 7];
        __syncthreads();
    }
    const QuadLane L = quad_lane_init(tid & 3, s_sig);
    uint32_t n = 1u << (hi - lo);
    for (uint32_t l = lo; l < hi; ++l, n >>= 1) {                        // level l of the block is at s_off: n digests
        const uint32_t sib = (uint32_t)(((i >> l) ^ 1) & (n - 1));
        if (tid < 8) out[l].h[tid] = sm[s_off + QR_DIGEST_OFF(sib) + tid];
        if (l + 1 < hi) {
            const uint32_t r = quad_tree_levels(sm, s_off, d_off, n, 1, tid, OP_THREADS, L);     // ends with a barrier
            d_off = s_off; s_off = r;
        }
    }
}

// Host-side checks: every segment between stored levels must fit the kernel's LDS buffers.
hipError_t launch_open_paths(const OpenTree* d_trees, const OpenTree* h_trees, uint32_t n_trees, uint32_t total_wgs, const uint64_t* d_idx, digest64* d_paths, fp* d_vals, hipStream_t s) {
    if (!n_trees || !total_wgs) return hipSuccess;
    for (uint32_t t = 0; t < n_trees; ++t) {
        const OpenTree& T = h_trees[t];
        const bool from_leaves = T.n_stored == 0 || T.level[0] > 0;
        if (!T.log_L || T.n_stored > OPEN_MAX_STORED || (!T.cw && (from_leaves || d_vals))) return hipErrorInvalidValue;
        for (uint32_t q = 0; q < T.n_stored; ++q) {
            if (!T.stored[q] || T.level[q] >= T.log_L || (q && T.level[q] <= T.level[q - 1])) return hipErrorInvalidValue;
            const uint32_t hi = q + 1 < T.n_stored ? T.level[q + 1] : T.log_L;
            if (hi - T.level[q] > 9) return hipErrorInvalidValue;        // 2^9 digests: the first level buffer
        }
        if ((T.n_stored ? T.level[0] : T.log_L) > 7) return hipErrorInvalidValue;   // 2^7 leaf message slots
    }
    hipLaunchKernelGGL(k_open_paths, dim3(total_wgs), dim3(OP_THREADS), 0, s, d_trees, n_trees, d_idx, d_paths, d_vals);
    return hipGetLastError();
}
