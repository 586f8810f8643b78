// api_merkle.hip — Merkle drivers behind include/stark_hip.h: digest reduction (wide levels, quad levels, host mailbox),
// commit / leaf hashing / open entry points.
#include "api_internal.h"

// ------------------------------------------------------------------ Merkle
// Reduce `count` (power of two) digests at `cur` to the root in c->d_root (and the host mailbox).  `cur` is not
// modified.  Wide levels: one 2^K-node subtree per lane (throughput).  From 2^17 nodes down: four lanes per
// compression, levels in LDS (latency), in at most two launches.
int32_t reduce_digests(sh_ctx* c, const digest64* cur, size_t count) {
    uint32_t lg = ilog2(count);
    c->mail_pending = false;
    if (lg == 0) {
        HIPCHK(hipMemcpyAsync(c->d_root, cur, sizeof(digest64), hipMemcpyDeviceToDevice, c->stream));
        return SH_OK;
    }
    bool use_b = true;                         // ping-pong SCR_DIG_B / SCR_DIG_A; the caller's input may live in SCR_DIG_A
    auto next_buf = [&](size_t n_out, digest64** out) -> int32_t {
        void* b;
        SHCHK(ensure_scratch(c, use_b ? SCR_DIG_B : SCR_DIG_A, n_out * sizeof(digest64), &b));
        use_b = !use_b;
        *out = (digest64*)b;
        return SH_OK;
    };
    while (lg > QUAD_MAX_LOG_COUNT) {
        const uint32_t K = lg - QUAD_MAX_LOG_COUNT < 3 ? lg - QUAD_MAX_LOG_COUNT : 3;
        digest64* out;
        SHCHK(next_buf(count >> K, &out));
        HIPCHK(launch_node_subtree(cur, out, count >> K, K, c->stream));
        cur = out; count >>= K; lg -= K;
    }
    if (lg > QUAD_MAX_LOG_CHUNK) {
        const uint32_t first = (lg + 1) / 2;
        digest64* out;
        SHCHK(next_buf(count >> first, &out));
        HIPCHK(launch_quad_reduce(cur, out, count, first, nullptr, 0, c->stream));
        cur = out; count >>= first; lg -= first;
    }
    if (++c->mail_seq == 0) c->mail_seq = 1;
    HIPCHK(launch_quad_reduce(cur, c->d_root, count, lg, c->d_mail, c->mail_seq, c->stream));
    c->mail_pending = true;
    return SH_OK;
}
uint32_t leaf_subtree_k(uint32_t log_L) {
    // enough lanes to fill the chip first (>= 2^18 lanes), then deeper per-lane subtrees
    uint32_t K = log_L > 18 ? log_L - 18 : 0;
    return K > 3 ? 3 : K;
}
// root of merkle(leafhash(elems[0..L))) -> c->d_root (enqueued, not synchronised)
int32_t merkle_root_from_elems(sh_ctx* c, const fp* d_elems, size_t L) {
    if (L == 1) { c->mail_pending = false; HIPCHK(launch_hash_leaves(d_elems, c->d_root, 1, c->stream)); return SH_OK; }
    const uint32_t K = leaf_subtree_k(ilog2(L));
    void* dig;
    SHCHK(ensure_scratch(c, SCR_DIG_A, (L >> K) * sizeof(digest64), &dig));
    HIPCHK(launch_leaf_subtree(d_elems, (digest64*)dig, L >> K, K, c->stream));
    return reduce_digests(c, (const digest64*)dig, L >> K);
}
// Bring the root of the last reduction to the host.  Fast path: the last kernel wrote root + sequence number
// into pinned host memory; poll for it (no interrupt-driven stream synchronise on the serial FRI chain).
// poll the mailbox flag for c->mail_seq; *got = false if the stream drained without it (caller takes a copy path)
int32_t wait_mailbox(sh_ctx* c, bool* got) {
    const uint32_t* flag = reinterpret_cast<const uint32_t*>(c->h_mail + 64);
    uint64_t spins = 0;
    *got = false;
    for (;;) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == c->mail_seq) { *got = true; return SH_OK; }
        if ((++spins & 0xfff) == 0) {
            hipError_t e = hipStreamQuery(c->stream);
            if (e == hipSuccess) { *got = (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == c->mail_seq); return SH_OK; }
            if (e != hipErrorNotReady) return sh_failf(SH_EHIP, "stream failed while waiting for a Merkle root: %s", hipGetErrorString(e));
        }
        __builtin_ia32_pause();
    }
}
int32_t fetch_root(sh_ctx* c, uint8_t root[64]) {
    if (c->mail_pending) {
        c->mail_pending = false;
        bool got = false;
        SHCHK(wait_mailbox(c, &got));
        if (got) { memcpy(root, c->h_mail, 64); return SH_OK; }
        ++c->mail_fallbacks;                   // stream drained but the mailbox write is not visible: take the copy path
    }
    HIPCHK(hipMemcpyAsync(c->h_pinned, c->d_root, 64, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy(root, c->h_pinned, 64);
    return SH_OK;
}

extern "C" {

int32_t sh_dev_merkle_commit(sh_ctx* c, const uint8_t* d_leaves, size_t L, uint8_t root[64]) {
    CTXCHK(c);
    if (!d_leaves || !root) return sh_fail(SH_EINVAL, "sh_dev_merkle_commit: null pointer");
    if (!is_pow2(L)) return sh_fail(SH_EINVAL, "sh_merkle_commit: leaf count must be a power of two (merkle.rs:36)");
    if (L == 1) {                                                  // merkle.rs:48,62: root == leaf
        HIPCHK(hipMemcpyAsync(c->h_pinned, d_leaves, 64, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        memcpy(root, c->h_pinned, 64);
        return SH_OK;
    }
    SHCHK(reduce_digests(c, (const digest64*)d_leaves, L));
    return fetch_root(c, root);
}
int32_t sh_merkle_commit(sh_ctx* c, const uint8_t* leaves, size_t L, uint8_t root[64]) {
    CTXCHK(c);
    if (!leaves || !root) return sh_fail(SH_EINVAL, "sh_merkle_commit: null pointer");
    if (!is_pow2(L)) return sh_fail(SH_EINVAL, "sh_merkle_commit: leaf count must be a power of two (merkle.rs:36)");
    void* d;
    SHCHK(ensure_scratch(c, SCR_IO_A, L * 64, &d));
    HIPCHK(hipMemcpyAsync(d, leaves, L * 64, hipMemcpyHostToDevice, c->stream));
    return sh_dev_merkle_commit(c, (const uint8_t*)d, L, root);
}
int32_t sh_dev_hash_leaves(sh_ctx* c, const uint64_t* d_elems, size_t L, uint8_t* d_out) {
    CTXCHK(c);
    if (L == 0) return SH_OK;
    if (!d_elems || !d_out) return sh_fail(SH_EINVAL, "sh_dev_hash_leaves: null pointer");
    HIPCHK(launch_hash_leaves((const fp*)d_elems, (digest64*)d_out, L, c->stream));
    return SH_OK;
}
int32_t sh_hash_leaves(sh_ctx* c, const uint64_t* elems, size_t L, uint8_t* out) {
    CTXCHK(c);
    if (L == 0) return SH_OK;
    if (!elems || !out) return sh_fail(SH_EINVAL, "sh_hash_leaves: null pointer");
    void *din, *dout;
    SHCHK(ensure_scratch(c, SCR_IO_A, L * 16, &din));
    SHCHK(ensure_scratch(c, SCR_IO_B, L * 64, &dout));
    HIPCHK(hipMemcpyAsync(din, elems, L * 16, hipMemcpyHostToDevice, c->stream));
    HIPCHK(launch_hash_leaves((const fp*)din, (digest64*)dout, L, c->stream));
    HIPCHK(hipMemcpyAsync(out, dout, L * 64, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SH_OK;
}
int32_t sh_dev_merkle_subtrees(sh_ctx* c, const uint64_t* d_elems, size_t n_out, uint32_t K, uint8_t* d_out) {
    CTXCHK(c);
    if (n_out == 0) return SH_OK;
    if (!d_elems || !d_out) return sh_fail(SH_EINVAL, "sh_dev_merkle_subtrees: null pointer");
    if (K > 4) return sh_fail(SH_EINVAL, "sh_dev_merkle_subtrees: K must be <= 4");
    HIPCHK(launch_leaf_subtree((const fp*)d_elems, (digest64*)d_out, n_out, K, c->stream));
    return SH_OK;
}
int32_t sh_dev_merkle_commit_elems(sh_ctx* c, const uint64_t* d_elems, size_t L, uint8_t root[64]) {
    CTXCHK(c);
    if (!d_elems || !root) return sh_fail(SH_EINVAL, "sh_dev_merkle_commit_elems: null pointer");
    if (!is_pow2(L)) return sh_fail(SH_EINVAL, "sh_merkle_commit_elems: element count must be a power of two (merkle.rs:36)");
    SHCHK(merkle_root_from_elems(c, (const fp*)d_elems, L));
    return fetch_root(c, root);
}
int32_t sh_merkle_commit_elems(sh_ctx* c, const uint64_t* elems, size_t L, uint8_t root[64]) {
    CTXCHK(c);
    if (!elems || !root) return sh_fail(SH_EINVAL, "sh_merkle_commit_elems: null pointer");
    if (!is_pow2(L)) return sh_fail(SH_EINVAL, "sh_merkle_commit_elems: element count must be a power of two (merkle.rs:36)");
    void* d;
    SHCHK(ensure_scratch(c, SCR_IO_A, L * 16, &d));
    HIPCHK(hipMemcpyAsync(d, elems, L * 16, hipMemcpyHostToDevice, c->stream));
    return sh_dev_merkle_commit_elems(c, (const uint64_t*)d, L, root);
}
// MerkleTree::open, merkle.rs:70-108: one index through the batched path (api_query.hip).
int32_t sh_merkle_open(sh_ctx* c, const uint8_t* leaves, size_t L, size_t index, uint8_t* path, size_t* path_len) {
    CTXCHK(c);
    if (!leaves || !path || !path_len) return sh_fail(SH_EINVAL, "sh_merkle_open: null pointer");
    if (!is_pow2(L)) return sh_fail(SH_EINVAL, "sh_merkle_open: leaf count must be a power of two (merkle.rs:72)");
    if (index >= L) return sh_fail(SH_EBOUNDS, "sh_merkle_open: index out of bounds (merkle.rs:75)");
    if (L < 2) return sh_fail(SH_EBOUNDS, "sh_merkle_open: a one-leaf tree has no path (merkle.rs:104)");
    const uint64_t idx = index;
    SHCHK(sh_merkle_open_batch(c, leaves, L, &idx, 1, path));
    *path_len = ilog2(L);
    return SH_OK;
}

// ---- device-resident pieces of a sharded commit: nothing here synchronises or touches the host
// root of merkle(leafhash(d_elems[0..L))) -> 64 bytes at d_root
int32_t sh_dev_merkle_root_elems(sh_ctx* c, const uint64_t* d_elems, size_t L, uint8_t* d_root) {
    CTXCHK(c);
    if (!d_elems || !d_root) return sh_fail(SH_EINVAL, "sh_dev_merkle_root_elems: null pointer");
    if (!is_pow2(L)) return sh_fail(SH_EINVAL, "sh_dev_merkle_root_elems: element count must be a power of two (merkle.rs:36)");
    SHCHK(merkle_root_from_elems(c, (const fp*)d_elems, L));
    c->mail_pending = false;                                       // nobody waits for this root on the host
    HIPCHK(hipMemcpyAsync(d_root, c->d_root, 64, hipMemcpyDeviceToDevice, c->stream));
    return SH_OK;
}
// root of the 2P-leaf tree over the all-gathered sub-root pairs of a sharded round: d_pairs = (P, 2) digests, rank s's
// roots over half-slices s and P + s; leaf order = half-slice order.  One host touch: the root.
int32_t sh_dev_merkle_commit_pairs(sh_ctx* c, const uint8_t* d_pairs, size_t P, uint8_t root[64]) {
    CTXCHK(c);
    if (!d_pairs || !root) return sh_fail(SH_EINVAL, "sh_dev_merkle_commit_pairs: null pointer");
    if (!is_pow2(P) || P > ((size_t)1 << 20)) return sh_fail(SH_EINVAL, "sh_dev_merkle_commit_pairs: rank count must be a power of two");
    void* lv;
    SHCHK(ensure_scratch(c, SCR_ROOTS, 2 * P * sizeof(digest64), &lv));
    HIPCHK(launch_pair_leaves((const digest64*)d_pairs, (digest64*)lv, P, c->stream));
    SHCHK(reduce_digests(c, (const digest64*)lv, 2 * P));
    return fetch_root(c, root);
}
int32_t sh_dev_zero(sh_ctx* c, void* d, size_t bytes) {
    CTXCHK(c);
    if (!bytes) return SH_OK;
    if (!d) return sh_fail(SH_EINVAL, "sh_dev_zero: null pointer");
    HIPCHK(hipMemsetAsync(d, 0, bytes, c->stream));
    return SH_OK;
}
int32_t sh_dev_transpose(sh_ctx* c, const uint64_t* d_in, size_t rows, size_t cols, uint64_t* d_out) {
    CTXCHK(c);
    if (!rows || !cols) return SH_OK;
    if (!d_in || !d_out || d_in == d_out) return sh_fail(SH_EINVAL, "sh_dev_transpose: null or aliased pointer");
    HIPCHK(launch_transpose16((const fp*)d_in, (fp*)d_out, rows, cols, c->stream));
    return SH_OK;
}
int32_t sh_dev_interleave_blocks(sh_ctx* c, const uint64_t* d_in, size_t P, size_t b, size_t a, uint64_t* d_out) {
    CTXCHK(c);
    if (!P || !b || !a) return SH_OK;
    if (!d_in || !d_out || d_in == d_out) return sh_fail(SH_EINVAL, "sh_dev_interleave_blocks: null or aliased pointer");
    HIPCHK(launch_interleave_blocks((const fp*)d_in, (fp*)d_out, P, b, a, c->stream));
    return SH_OK;
}

}  // extern "C"
