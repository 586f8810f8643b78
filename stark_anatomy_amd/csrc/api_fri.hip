// api_fri.hip — FRI drivers behind include/stark_hip.h: split-and-fold entry points, the commit loop (single codeword and
// lock-step batches), the library-side transcript callback.
#include "api_internal.h"
#include <unistd.h>

// ------------------------------------------------------------------ FRI
int32_t check_params(const sh_fri_params* p, fp* offset) {
    if (!p) return sh_fail(SH_EINVAL, "fri: null params");
    if (p->expansion == 0) return sh_fail(SH_EINVAL, "fri: expansion must be >= 1");
    *offset = fp_load_host(p->offset);
    if (!fp_is_canonical(*offset) || fp_is_zero(*offset)) return sh_fail(SH_EINVAL, "fri: offset must be a non-zero canonical field element");
    return SH_OK;
}
// ------------------------------------------------------------------ the commit chain (ChainLayout, ChainGuard: api_internal.h)
// Is `p` mapped into this process?  Asked of the kernel (a pipe write from an unmapped address fails with EFAULT) rather than found out
// by a store that would take the process down.
static bool host_can_read(const void* p) {
    int fd[2];
    if (pipe(fd) != 0) return false;
    const ssize_t n = write(fd[1], p, 4);
    close(fd[0]); close(fd[1]);
    return n == 4;
}
int32_t ensure_chain(sh_ctx* c, size_t bytes, size_t c0_end) {
    if (c->chain_bytes < bytes) {
        if (c->h_chain) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipHostFree(c->h_chain)); c->h_chain = nullptr; c->d_chain = nullptr; c->chain_bytes = 0; }
        const size_t want = (bytes + 4095) & ~(size_t)4095;
        HIPCHK(hipHostMalloc((void**)&c->h_chain, want, hipHostMallocMapped | hipHostMallocCoherent));
        memset(c->h_chain, 0, want);
        HIPCHK(hipHostGetDevicePointer((void**)&c->d_chain, c->h_chain, 0));
        c->chain_bytes = want;
    }
    // The c0 records: on the card when the host can store into device memory (large BAR).  The GPU then polls its own memory instead
    // of reading host memory over PCIe for every poll, and reads the constants locally once they are there: ~1.5 us less per
    // hand-off (tools/bar_probe.hip: host store -> GPU sees it -> GPU store -> host sees it, 2.43 us with a pinned host word, 1.91 us
    // with an uncached device word; the constants' own PCIe read on top in the pinned form).
    static const bool pinned_c0 = getenv("STARK_HIP_C0_PINNED") != nullptr;
    int large_bar = 0;
    if (!pinned_c0) (void)hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, c->device);
    if (large_bar && c->c0_bytes < c0_end) {
        if (c->c0_buf) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(c->c0_buf)); c->c0_buf = nullptr; c->c0_bytes = 0; }
        const size_t want = (c0_end + 65535) & ~(size_t)65535;
        if (hipExtMallocWithFlags((void**)&c->c0_buf, want, hipDeviceMallocUncached) == hipSuccess) {
            HIPCHK(hipMemsetAsync(c->c0_buf, 0, want, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            if (host_can_read(c->c0_buf)) c->c0_bytes = want;
            else { (void)hipFree(c->c0_buf); c->c0_buf = nullptr; }     // the attribute promised a mapping that is not there: stay with pinned records
        } else { (void)hipGetLastError(); c->c0_buf = nullptr; }
    }
    if (large_bar && c->c0_buf) c->c0_host = c->c0_dev = c->c0_buf;
    else { c->c0_host = c->h_chain; c->c0_dev = c->d_chain; }
    return SH_OK;
}
// Wait for a root record.  NO HIP call on this path: while the host sits here the next round's kernel may already be spinning
// on the GPU for the challenge this thread is about to produce, and a runtime call can block behind another thread's call
// that itself waits for GPU work queued behind that spinning kernel (seen with two contexts in two threads).  So: poll the
// record, the abort word that a wait kernel raises when its bounded spin runs out, and a wall clock.
// Returns SH_OK, WAIT_RETRY (a wait kernel gave up: the caller re-enqueues from this round) or an error.
int32_t wait_seq(sh_ctx* c, const uint32_t* flag, uint32_t gen, uint32_t round, double deadline_s, const ChainEnqueuer* helper) {
    const volatile uint32_t* hd = reinterpret_cast<volatile uint32_t*>(c->h_chain);
    const auto t0 = std::chrono::steady_clock::now();
    uint64_t spins = 0;
    for (;;) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == gen) return SH_OK;
        if ((++spins & 0xff) == 0) {
            if (hd[0]) return __atomic_load_n(flag, __ATOMIC_ACQUIRE) == gen ? SH_OK : WAIT_RETRY;
            if (helper && helper->failed()) return __atomic_load_n(flag, __ATOMIC_ACQUIRE) == gen ? SH_OK : WAIT_ENQUEUE_FAILED;
            if ((spins & 0x3ff) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > deadline_s) {
                const hipError_t e = hipStreamQuery(c->stream);
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == gen) return SH_OK;
                return sh_failf(SH_EHIP, "fri_commit: no Merkle root for round %u after %.0f s (stream: %s)", round, deadline_s,
                                e == hipSuccess ? "drained" : hipGetErrorString(e));
            }
        }
        __builtin_ia32_pause();
    }
}
// header word 3 of a give-up = (address of the sequence word - address of the header) mod 2^31: back to the word's layout offset,
// whichever buffer the c0 records are in
static uint32_t c0_offset_of(const sh_ctx* c, uint32_t word3) {
    const uint32_t delta = (uint32_t)((uintptr_t)c->c0_dev - (uintptr_t)c->d_chain);
    return ((word3 & 0x7fffffffu) - delta) & 0x7fffffffu;
}
// what the wait kernel that gave up last left in the chain header (round_kernels.hip)
static int32_t fail_gave_up(sh_ctx* c, uint32_t round, uint32_t gen, uint32_t retries) {
    const volatile uint32_t* hd = reinterpret_cast<volatile uint32_t*>(c->h_chain);
    return sh_failf(SH_EHIP, "fri_commit: round %u: the GPU still does not see its challenge after %u re-launches (gen %u; kernel waited for gen %u, record held %u, chain offset 0x%x%s, "
                    "block %u of %u, %u polls, xcc %u; the host sees %u there now)", round, retries, gen, hd[1], hd[2], c0_offset_of(c, hd[3]), (hd[3] >> 31) ? " [k_round_wg]" : " [k_wait_c0]",
                    hd[4] & 0xffff, hd[4] >> 16, hd[5], hd[6], *reinterpret_cast<volatile uint32_t*>(c->c0_host + c0_offset_of(c, hd[3])));
}

// How many levels the next launch of a forest reduction takes off B trees of 2^lg digests each, and with which kernel.
// Wide levels: one 2^step-node subtree per lane; from 2^17 digests down: four lanes per compression, levels in LDS —
// full workgroups (2^9 digests) while there is work, balanced launches once it is latency only.
uint32_t forest_step(uint32_t lg, size_t B, bool* quad) {
    const size_t total = B << lg;
    static const uint32_t quad_max = [] { const char* e = getenv("STARK_HIP_QUAD_MAX_LOG"); const int v = e ? atoi(e) : 0; return (uint32_t)(v >= 10 && v <= 22 ? v : QUAD_MAX_LOG_COUNT); }();
    if (total > ((size_t)1 << quad_max)) {
        const uint32_t over = ilog2(total) - quad_max;
        uint32_t step = over < 3 ? over : 3;
        *quad = false;
        return step > lg ? lg : step;
    }
    *quad = true;
    return lg <= QUAD_MAX_LOG_CHUNK ? lg : (total >= ((size_t)1 << 14) ? QUAD_MAX_LOG_CHUNK : (lg + 1) / 2);
}
// digests that the launches of a forest reduction write below the roots (what a LevelStore must hold for it)
static size_t forest_store_need(uint32_t lg, size_t B) {
    size_t need = 0;
    while (lg) { bool q; lg -= forest_step(lg, B, &q); if (lg) need += B << lg; }
    return need;
}
// Keeps what the commit's kernels write on the way up a tree (CommitKeep, api_internal.h): every round owns a fixed range of
// one arena, so a round that is enqueued a second time writes to the same places.
struct LevelStore {
    digest64* base = nullptr;
    std::vector<size_t> round_off;
    size_t cur = 0;
    CommitKeep::Tree* tree = nullptr;
    void begin_round(uint32_t r, CommitKeep::Tree* t) { cur = round_off[r]; tree = t; tree->stored.clear(); }
    digest64* take(size_t count, uint32_t level) { digest64* p = base + cur; cur += count; tree->stored.push_back({level, p}); return p; }
};

// Forest reduction: B trees of 2^lg digests each, back to back at `cur` -> B roots in the round's record + sequence word.
// Without a store the launches ping-pong between `other` and `self` (the buffer `cur` lives in); with one, every launch's
// output stays: `level0` = log2 of the leaves under one digest of `cur`.
static int32_t enqueue_forest_reduce(sh_ctx* c, const digest64* cur, digest64* other, digest64* self, uint32_t lg, size_t B, uint8_t* d_rec, uint32_t gen, const uint32_t* d_abort,
                                     LevelStore* store = nullptr, uint32_t level0 = 0, digest64* d_roots = nullptr) {
    uint64_t* rec_roots = reinterpret_cast<uint64_t*>(d_rec);
    uint32_t* rec_seq = reinterpret_cast<uint32_t*>(d_rec + B * 64);
    if (lg == 0) { HIPCHK(launch_publish_digests(cur, B, rec_roots, rec_seq, gen, c->stream, d_abort)); return SH_OK; }
    while (lg > 0) {
        const size_t total = B << lg;
        bool quad;
        const uint32_t step = forest_step(lg, B, &quad);
        level0 += step;
        if (!quad) {
            if (step == lg) {                                      // B > 2^17 trees: finish them lane-per-subtree, then publish
                HIPCHK(launch_node_subtree(cur, other, total >> step, step, c->stream));
                HIPCHK(launch_publish_digests(other, B, rec_roots, rec_seq, gen, c->stream, d_abort));
                return SH_OK;
            }
            digest64* out = store ? store->take(total >> step, level0) : other;
            HIPCHK(launch_node_subtree(cur, out, total >> step, step, c->stream));
            cur = out; other = self; self = out; lg -= step;
            continue;
        }
        if (step == lg) {
            if (B == 1) HIPCHK(launch_quad_reduce(cur, c->d_root, total, step, rec_roots, gen, c->stream, nullptr, 0, d_abort));
            else if (d_roots) {                                    // B roots to device memory, then one small kernel hands them to the host
                HIPCHK(launch_quad_reduce(cur, d_roots, total, step, nullptr, 0, c->stream, nullptr, 0, d_abort));
                HIPCHK(launch_publish_digests(d_roots, B, rec_roots, rec_seq, gen, c->stream, d_abort));
            } else {
                c->batch_done_total += (uint32_t)B;
                HIPCHK(launch_quad_reduce(cur, reinterpret_cast<digest64*>(d_rec), total, step, reinterpret_cast<uint64_t*>(d_rec + B * 64 - 64), gen, c->stream,
                                          c->d_flag + 16, c->batch_done_total, d_abort));
            }
            return SH_OK;
        }
        digest64* out = store ? store->take(total >> step, level0) : other;
        HIPCHK(launch_quad_reduce(cur, out, total, step, nullptr, 0, c->stream, nullptr, 0, d_abort));
        cur = out; other = self; self = out; lg -= step;
    }
    return SH_OK;
}

// Per round the host side gets the B roots and, when `want`, returns the B challenges (canonical).
typedef std::function<int32_t(uint32_t r, const uint8_t* roots, bool want, fp* alpha_out)> RoundFn;

// The commit of B equal-length codewords stored back to back, rounds in lock step (B = 1: one codeword).
// `keep`: when set (B == 1), every round's codeword stays resident (round r >= 1 at offset N - N/2^(r-1) of one N-element
// arena, instead of the two ping-pong buffers), and so does every digest array that the round's kernels write on the way up
// its tree (SCR_LEVELS instead of the two ping-pong digest buffers) — the query phase opens paths from them afterwards.
// `r0_leaves_hashed`: the subtree digests of round 0 already sit in SCR_DIG_A (the host-buffer entry points hash every chunk of the
// codeword as it lands): the first enqueue of round 0 skips its leaf kernel (a re-enqueue after an abandoned round does not).
static int32_t fri_commit_core(sh_ctx* c, const fp* d_cws, size_t B, size_t N, const sh_fri_params* prm, const RoundFn& fn,
                               uint64_t* last_out, size_t* last_len, CommitKeep* keep, bool r0_leaves_hashed = false) {
    const uint32_t log_N = ilog2(N);
    fp off;
    SHCHK(check_params(prm, &off));
    const uint32_t R = sh_fri_num_rounds(N, prm->expansion, prm->num_colinearity_tests);
    if (keep) { keep->cws.assign(1, d_cws); keep->trees.assign(R ? R : 1, CommitKeep::Tree()); keep->trees[0].lg = log_N; }
    if (R == 0) {
        HIPCHK(hipMemcpyAsync(last_out, d_cws, B * N * sizeof(fp), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        *last_len = N;
        return SH_OK;
    }
    // ---- everything that allocates or synchronises happens before the first kernel is enqueued
    const size_t L_last = N >> (R - 1);
    const size_t last_bytes = B * L_last * sizeof(fp);
    const bool stage_last = last_bytes <= ((size_t)64 << 20);
    const ChainLayout lay(B, R, stage_last ? last_bytes : 0);
    SHCHK(ensure_chain(c, lay.total, lay.off_last));
    const fp* tw_inv;
    SHCHK(get_twiddles(c, log_N, true, &tw_inv));
    void *bufX = nullptr, *bufY = nullptr, *arena = nullptr, *digA, *digB, *d_c0;
    if (keep) { if (R > 1) SHCHK(ensure_scratch(c, SCR_ARENA, N * sizeof(fp), &arena)); }
    else {
        if (R > 1) SHCHK(ensure_scratch(c, SCR_IO_A, B * (N / 2) * sizeof(fp), &bufX));
        if (R > 2) SHCHK(ensure_scratch(c, SCR_IO_B, B * (N / 4) * sizeof(fp), &bufY));
    }
    auto leaf_k = [&](size_t len) {                 // per-lane subtree depth: fill the chip first, never cross a codeword
        uint32_t K = leaf_subtree_k(ilog2(B * len));
        const uint32_t lg = ilog2(len);
        return K < lg ? K : lg;
    };
    static const bool use_round_wg = getenv("STARK_NO_ROUND_WG") == nullptr;
    // tree of round r: 2^lg_tree leaves per codeword; `small`: the fused one-launch round (k_round_wg); K: leaves under one
    // digest of the round's first kernel (== lg_tree when the tree ends in it)
    struct RoundShape { uint32_t lg_tree; bool small; uint32_t K; size_t first; };
    auto shape_of = [&](uint32_t r) {
        RoundShape sh;
        sh.lg_tree = log_N - r;
        sh.small = use_round_wg && (B << sh.lg_tree) <= ((size_t)1 << round_wg_max_log()) && sh.lg_tree >= 1;
        sh.K = sh.small ? (sh.lg_tree < 7 ? sh.lg_tree : 7) : leaf_k((size_t)1 << sh.lg_tree);
        sh.first = sh.K < sh.lg_tree ? (B << sh.lg_tree) >> sh.K : 0;
        return sh;
    };
    const size_t dig0 = (B * N) >> leaf_k(N);
    LevelStore store;
    if (keep) {
        size_t need = 0;
        for (uint32_t r = 0; r < R; ++r) {
            const RoundShape sh = shape_of(r);
            store.round_off.push_back(need);
            need += sh.first + forest_store_need(sh.lg_tree - sh.K, B);
        }
        void* lv;
        SHCHK(ensure_scratch(c, SCR_LEVELS, (need ? need : 1) * sizeof(digest64), &lv));
        store.base = (digest64*)lv;
        digA = digB = nullptr;
    } else {
        SHCHK(ensure_scratch(c, SCR_DIG_A, dig0 * sizeof(digest64), &digA));
        SHCHK(ensure_scratch(c, SCR_DIG_B, (dig0 / 2 ? dig0 / 2 : 1) * sizeof(digest64), &digB));
    }
    SHCHK(ensure_scratch(c, SCR_C0, B * sizeof(fp), &d_c0));
    void* d_roots_v = nullptr;
    static const bool publish_kernel = getenv("STARK_HIP_COUNTER_PUBLISH") == nullptr;
    if (B > 1 && publish_kernel) SHCHK(ensure_scratch(c, SCR_ROOTS, B * sizeof(digest64), &d_roots_v));
    digest64* d_roots = (digest64*)d_roots_v;
    uint32_t gen = (c->chain_gen + 1) & 0x7fffffffu;
    if (gen == 0) gen = 1;
    c->chain_gen = gen;
    // A wait kernel spins for at most `spin_us`, then abandons its round; the host re-enqueues from that round (below).  The
    // wall-clock `timeout_s` bounds the whole wait for one root.
    const double timeout_s = [] { const char* e = getenv("STARK_HIP_CHAIN_TIMEOUT_S"); double v = e ? atof(e) : 0; return v > 0 ? v : 20.0; }();
    const double spin_us = [] { const char* e = getenv("STARK_HIP_CHAIN_SPIN_US"); double v = e ? atof(e) : 0; return v > 0 ? v : 500.0; }();
    const uint64_t timeout_ticks = (uint64_t)(spin_us * 100.0) + 1;                   // s_memrealtime: 100 MHz
    uint32_t* d_abort = c->d_flag + 32;
    uint32_t* h_abort = reinterpret_cast<uint32_t*>(c->h_chain);
    uint32_t* hd_abort = reinterpret_cast<uint32_t*>(c->d_chain);
    for (int i = 0; i < 16; ++i) h_abort[i] = 0;
    // The layout depends on (B, R): a sequence word of this commit may lie where an earlier commit of another shape left a
    // root, a fold constant or a staged codeword element — bytes that can equal `gen`.  The stream is drained (every commit
    // ends with a synchronise), so nothing reads these words now: lower them all.
    for (uint32_t r = 0; r < R; ++r) {
        *reinterpret_cast<volatile uint32_t*>(c->h_chain + lay.root_seq(r)) = 0;
        *reinterpret_cast<volatile uint32_t*>(c->c0_host + lay.c0_seq(r)) = 0;
    }
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    c0_flush();
    // one memset: the workgroup counter of the counter-published launches (d_flag[16]; a commit that failed half way leaves it short of
    // the host's running total) and the abort word (d_flag[32])
    c->batch_done_total = 0;
    HIPCHK(hipMemsetAsync(c->d_flag + 16, 0, 17 * 4, c->stream));
    const bool dbg = getenv("STARK_HIP_DEBUG") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    // debug runs pair the device's 100 MHz clock with the host's once per commit (a tiny kernel stamps pinned memory, the host
    // polls for it: good to a few us), so that a wait kernel's post-mortem stamps can be laid beside the host's own.
    double cal_host_us = 0; uint64_t cal_ticks = 0;
    if (dbg) {
        volatile uint64_t* h_clk = reinterpret_cast<volatile uint64_t*>(h_abort + 12);
        HIPCHK(launch_clock_probe(reinterpret_cast<uint64_t*>(hd_abort + 12), c->stream));
        const double t_give_up = now() + 1e6;
        while (!(cal_ticks = *h_clk) && now() < t_give_up) __builtin_ia32_pause();
        cal_host_us = now();
        HIPCHK(hipStreamSynchronize(c->stream));
        *h_clk = 0;
    }
    const double t_start = now();

    // ---- the kernels of ALL rounds are enqueued before the chain is served; round r+1's kernels start by waiting for
    // challenge r.  Round r >= 1 reads codeword r-1 (length N >> (r-1)) and writes codeword r; both are functions of r alone,
    // so that a round can be enqueued a second time after its wait kernel gave up.
    auto cw_of = [&](uint32_t r) -> fp* {          // codeword r, r >= 1: arena offset N - N/2^(r-1), or the ping-pong buffers
        return keep ? (fp*)arena + (N - (N >> (r - 1))) : (fp*)(((r - 1) & 1) ? bufY : bufX);
    };
    if (keep) for (uint32_t r = 1; r < R; ++r) { keep->cws.push_back(cw_of(r)); keep->trees[r].lg = log_N - r; }
    // With the c0 records on the card (ensure_chain) every kernel that needs a challenge waits for it itself — lane 0 of each workgroup
    // polls the record's sequence word in local memory — and the k_wait_c0 launch in front of every round is gone (one launch boundary
    // per round off the serial chain).  Records in pinned host memory keep the wait kernel: one poller, one PCIe read per poll.
    static const bool wait_kernel = getenv("STARK_HIP_WAIT_KERNEL") != nullptr;              // A/B
    // One codeword only: with B > 1 every lane would fetch its codeword's constant from the uncached record (128 x 2^14: 0.950 ms
    // against 0.937 with the wait kernel, which copies the B constants to cached memory once).
    const bool direct_wait = B == 1 && c->c0_buf != nullptr && c->c0_dev == c->c0_buf && !wait_kernel;
    uint32_t enq = 0;                              // rounds enqueued so far
    auto enqueue_round = [&]() -> int32_t {
        const uint32_t r = enq;
        uint8_t* rec = c->d_chain + lay.roots(r);
        const RoundShape sh = shape_of(r);
        LevelStore* st = keep ? &store : nullptr;
        digest64 *first = (digest64*)digA, *other = (digest64*)digB;
        if (keep) { store.begin_round(r, &keep->trees[r]); first = sh.first ? store.take(sh.first, sh.K) : nullptr; other = nullptr; }
        // small rounds: the whole round (wait, fold, leaf hash, bottom 7 levels) is one launch, plus one for the levels above
        auto small_round = [&](bool fold, const fp* in, fp* out, uint32_t lg_tree) -> int32_t {
            const size_t blocks = ((B << lg_tree) + 127) / 128;
            uint32_t target = 0;
            const bool roots_via_kernel = lg_tree <= 7 && blocks > 1 && d_roots;      // finished roots of several workgroups: to device memory, then k_publish_digests
            if (lg_tree <= 7 && blocks > 1 && !roots_via_kernel) { c->batch_done_total += (uint32_t)blocks; target = c->batch_done_total; }
            const bool poll_inside = fold && (direct_wait || (B == 1 && blocks == 1));   // records in pinned host memory: a wait kernel first unless one workgroup is all there is
            if (fold && !poll_inside)
                HIPCHK(launch_wait_c0(reinterpret_cast<const uint32_t*>(c->c0_dev + lay.c0_seq(r - 1)), gen, reinterpret_cast<const fp*>(c->c0_dev + lay.c0(r - 1)),
                                      (fp*)d_c0, B, d_abort, hd_abort, timeout_ticks, c->stream));
            HIPCHK(launch_round_wg(fold, in, out, lg_tree, B, poll_inside ? reinterpret_cast<const uint32_t*>(c->c0_dev + lay.c0_seq(r - 1)) : nullptr, gen,
                                   poll_inside ? reinterpret_cast<const fp*>(c->c0_dev + lay.c0(r - 1)) : nullptr, (const fp*)d_c0, d_abort, hd_abort, timeout_ticks, tw_inv, log_N,
                                   fold ? r - 1 : 0, roots_via_kernel ? d_roots : first, reinterpret_cast<uint64_t*>(rec), reinterpret_cast<uint32_t*>(rec + B * 64),
                                   roots_via_kernel ? nullptr : c->d_flag + 16, target, c->stream));
            if (roots_via_kernel) HIPCHK(launch_publish_digests(d_roots, B, reinterpret_cast<uint64_t*>(rec), reinterpret_cast<uint32_t*>(rec + B * 64), gen, c->stream, d_abort));
            if (lg_tree > 7) SHCHK(enqueue_forest_reduce(c, first, other, first, lg_tree - 7, B, rec, gen, d_abort, st, 7, d_roots));
            return SH_OK;
        };
        if (r == 0) {
            if (sh.small) SHCHK(small_round(false, d_cws, nullptr, log_N));
            else {
                const uint32_t K = sh.K;
                digest64* dst = first ? first : c->d_root;              // a tree that ends in its lanes (N <= 2^K): one digest, the root
                if (r0_leaves_hashed && !keep) r0_leaves_hashed = false;                  // consumed: dst == SCR_DIG_A holds them
                else HIPCHK(launch_leaf_subtree(d_cws, dst, (B * N) >> K, K, c->stream));
                SHCHK(enqueue_forest_reduce(c, dst, other, dst, log_N - K, B, rec, gen, d_abort, st, K, d_roots));
            }
        } else {
            const size_t L = N >> (r - 1), half = L / 2;
            const fp* in = r == 1 ? d_cws : cw_of(r - 1);
            fp* next = cw_of(r);
            if (sh.small) SHCHK(small_round(true, in, next, ilog2(half)));
            else {
                const uint32_t K = sh.K;
                digest64* dst = first ? first : c->d_root;
                if (direct_wait) {                                 // the fold kernel waits itself and reads the constants out of the record
                    const ChainWait cwt{reinterpret_cast<const uint32_t*>(c->c0_dev + lay.c0_seq(r - 1)), gen, d_abort, hd_abort, timeout_ticks};
                    HIPCHK(launch_fold_leaf_subtree(in, L, B, reinterpret_cast<const fp*>(c->c0_dev + lay.c0(r - 1)), fp_zero(), tw_inv, log_N, r - 1, next, dst, K, c->stream, d_abort, 0,
                                                    &cwt));
                } else {
                    HIPCHK(launch_wait_c0(reinterpret_cast<const uint32_t*>(c->c0_dev + lay.c0_seq(r - 1)), gen, reinterpret_cast<const fp*>(c->c0_dev + lay.c0(r - 1)),
                                          (fp*)d_c0, B, d_abort, hd_abort, timeout_ticks, c->stream));
                    HIPCHK(launch_fold_leaf_subtree(in, L, B, (const fp*)d_c0, fp_zero(), tw_inv, log_N, r - 1, next, dst, K, c->stream, d_abort));
                }
                SHCHK(enqueue_forest_reduce(c, dst, other, dst, ilog2(half) - K, B, rec, gen, d_abort, st, K, d_roots));
            }
        }
        ++enq;
        if (enq == R && stage_last) HIPCHK(launch_copy_to_host(R > 1 ? cw_of(R - 1) : d_cws, reinterpret_cast<fp*>(c->d_chain + lay.off_last), B * L_last, c->stream));
        return SH_OK;
    };
    // Who makes the launches.  While a wait kernel spins on the GPU for a challenge, the thread that will produce that challenge
    // must not be inside the HIP runtime: a launch can block behind another host thread's hipFree / context teardown, which waits
    // for the device to go idle, i.e. for that very kernel — a cycle that only the kernel's bounded spin breaks (measured: the
    // "stale" sequence words of round 2 were records the blocked host thread had not yet written).  So the
    // caller enqueues round 0 (no wait in it) and the context's enqueue thread, which nothing on the GPU waits for, the rest;
    // the caller then only reads and writes pinned memory.  STARK_HIP_CHAIN_INLINE_ENQUEUE=1 keeps round 2's scheme (the
    // caller enqueues one round ahead between two challenges) for A/B runs.
    static const bool inline_enq = getenv("STARK_HIP_CHAIN_INLINE_ENQUEUE") != nullptr;
    ChainEnqueuer* helper = nullptr;
    if (R >= 2 && !inline_enq) {
        if (!c->enqueuer) { try { c->enqueuer = new ChainEnqueuer(c->device); } catch (const std::system_error&) { c->enqueuer = nullptr; } }
        helper = c->enqueuer;                      // no thread to be had: the caller enqueues as in round 2
    }
    std::atomic<int> gate{0};                      // 0: round 0 not enqueued yet, 1: go, 2: round 0 failed
    auto enqueue_rest = [&]() -> int32_t { while (enq < R) SHCHK(enqueue_round()); return SH_OK; };
    std::string helper_msg;
    ChainGuard guard{c, &lay, gen};                // declared after everything a posted job refers to
    guard.waits = R - 1;
    guard.helper = helper;
    guard.armed = true;
    if (helper) {
        helper->post([&]() -> int32_t {            // woken now, so that it is awake by the time round 0 is on the stream
            int g;
            uint32_t spins = 0;
            while ((g = gate.load(std::memory_order_acquire)) == 0) { if (++spins > 20000) std::this_thread::yield(); else __builtin_ia32_pause(); }
            return g == 1 ? enqueue_rest() : SH_OK;
        });
        const int32_t rc0 = enqueue_round();
        gate.store(rc0 == SH_OK ? 1 : 2, std::memory_order_release);
        if (rc0 != SH_OK) return rc0;
    } else {
        SHCHK(enqueue_round());
        if (R > 1) SHCHK(enqueue_round());
    }

    // ---- serve the chain: root_r -> transcript -> challenge_r -> fold constants.  No HIP call in here (helper mode).
    const fp inv2 = fp_inv(fp_make(2, 0, 0, 0));
    fp off_inv = fp_inv(off);                      // (off^(2^r))^-1 = (off^-1)^(2^r): one inversion for the whole commit
    std::vector<fp> alpha(B);
    std::vector<double> t_pub(R, 0.0), t_enq0(R, 0.0), t_enq1(R, 0.0);    // debug: when c0(r) was published; the caller's enqueue after it
    double t_wait = 0, t_host = 0, t_enq = 0;
    uint32_t retries = 0;
    const uint32_t max_retries = 4 * R + 32;
    for (uint32_t r = 0; r < R; ++r) {
        const double t0 = now();
        for (;;) {
            const int32_t w = wait_seq(c, reinterpret_cast<const uint32_t*>(c->h_chain + lay.root_seq(r)), gen, r, timeout_s, helper);
            if (w == SH_OK) break;
            if (w == WAIT_ENQUEUE_FAILED) { const int32_t rc = helper->wait(&helper_msg); return sh_fail(rc, helper_msg.c_str()); }
            if (w != WAIT_RETRY) return w;
            // The wait kernel of round r (or of r+1, which then found root r published) ran out of spin before the challenge was
            // there: the transcript callback was slower than the budget, or (inline mode) this thread sat in a launch that was
            // blocked behind another thread's hipFree.  Everything behind that kernel on the stream has returned at sight of the
            // abort word, so no codeword or root of round >= r was written: drain, lower the flags, enqueue again from round r.
            if (dbg) {
                const volatile uint32_t* hd = h_abort;
                const uint64_t tk1 = *reinterpret_cast<const volatile uint64_t*>(h_abort + 8), tk0 = *reinterpret_cast<const volatile uint64_t*>(h_abort + 10);
                const double g1 = cal_host_us + ((double)(int64_t)(tk1 - cal_ticks)) / 100.0 - t_start, g0 = cal_host_us + ((double)(int64_t)(tk0 - cal_ticks)) / 100.0 - t_start;
                const uint32_t off = c0_offset_of(c, hd[3]);
                const uint32_t k = (uint32_t)((off - lay.off_c0) / lay.rec_c0);
                fprintf(stderr, "[stark_hip] GAVE UP: %s waited for c0(%u) (gen %u) from t=%.1f to t=%.1f us (%u polls, xcc %u); it read %u (polling load), %u (non-temporal load), neighbour word %u.\n"
                                "[stark_hip]   host: c0(%u) %s; now t=%.1f, serving round %u; this thread's last enqueue: round-%u slot, t=%.1f .. %.1f us%s\n",
                        (hd[3] >> 31) ? "k_round_wg" : "k_wait_c0", k, hd[1], g0, g1, hd[5], hd[6], hd[2], hd[7], hd[14], k,
                        k < R && t_pub[k] > 0 ? "published at" : "NOT published yet", now() - t_start, r, r ? r - 1 : 0, r ? t_enq0[r - 1] - t_start : 0.0, r ? t_enq1[r - 1] - t_start : 0.0,
                        helper ? " (enqueue thread in use)" : "");
                if (k < R && t_pub[k] > 0) fprintf(stderr, "[stark_hip]   c0(%u) published at t=%.1f us: %s the kernel gave up\n", k, t_pub[k] - t_start, t_pub[k] - t_start > g1 ? "AFTER" : "BEFORE");
            }
            if (++retries > max_retries) return fail_gave_up(c, r, gen, retries - 1);
            if (helper) { const int32_t rc = helper->wait(&helper_msg); if (rc != SH_OK) return sh_fail(rc, helper_msg.c_str()); }
            HIPCHK(hipStreamSynchronize(c->stream));
            if (__atomic_load_n(reinterpret_cast<const uint32_t*>(c->h_chain + lay.root_seq(r)), __ATOMIC_ACQUIRE) == gen) {
                // root r did arrive: it was round r+1's wait that gave up; serve root r first, recover on the next wait
                break;
            }
            for (int i = 0; i < 12; ++i) __atomic_store_n(h_abort + i, 0u, __ATOMIC_RELAXED);
            c->batch_done_total = 0;
            HIPCHK(hipMemsetAsync(c->d_flag + 16, 0, 17 * 4, c->stream));
            enq = r;
            if (helper) helper->post(enqueue_rest);
            else {
                SHCHK(enqueue_round());
                if (enq < R) SHCHK(enqueue_round());
            }
            if (dbg) fprintf(stderr, "[stark_hip] round %u re-enqueued (%u so far)\n", r, retries);
        }
        const double t1 = now();
        const bool want = r + 1 < R;
        SHCHK(fn(r, c->h_chain + lay.roots(r), want, alpha.data()));
        if (want) {
            fp* h_c0 = reinterpret_cast<fp*>(c->c0_host + lay.c0(r));
            const fp k_rr = fp_to_mont(fp_to_mont(fp_mul(inv2, off_inv)));            // 2^-1 · offset_r^-1 · R^2: one multiply per codeword below
            for (size_t b = 0; b < B; ++b) {
                if (!fp_is_canonical(alpha[b])) return sh_fail(SH_EINVAL, "fri_commit: callback returned a non-canonical alpha");
                h_c0[b] = mont_mul(alpha[b], k_rr);                                   // c0 = alpha · 2^-1 · offset_r^-1, in Montgomery form
            }
            c0_flush();                                                               // the constants before the sequence word, also through a write-combining BAR mapping
            __atomic_store_n(reinterpret_cast<uint32_t*>(c->c0_host + lay.c0_seq(r)), gen, __ATOMIC_RELEASE);
            c0_flush();
            if (dbg) t_pub[r] = now();
            guard.next_c0 = r + 1;
            off_inv = fp_mul(off_inv, off_inv);
        }
        const double t2 = now();
        if (!helper && enq < R) SHCHK(enqueue_round());
        if (dbg) {
            const double t3 = now();
            t_enq0[r] = t2; t_enq1[r] = t3;
            t_wait += t1 - t0; t_host += t2 - t1; t_enq += t3 - t2;
            fprintf(stderr, "[stark_hip] round %2u L=2^%-2u  wait root %6.1f us | transcript + c0 %5.1f us | enqueue ahead %5.1f us\n", r, ilog2(N >> r), t1 - t0, t2 - t1, t3 - t2);
        }
    }
    if (helper) { const int32_t rc = helper->wait(&helper_msg); if (rc != SH_OK) return sh_fail(rc, helper_msg.c_str()); }
    c->chain_retries += retries;
    const fp* cur = R > 1 ? cw_of(R - 1) : d_cws;
    const size_t L = L_last;
    guard.armed = false;
    if (!stage_last) HIPCHK(hipMemcpyAsync(last_out, cur, last_bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (*h_abort) return sh_fail(SH_EHIP, "fri_commit: a round timed out on the GPU waiting for its challenge");
    if (stage_last) memcpy(last_out, c->h_chain + lay.off_last, last_bytes);
    *last_len = L;
    if (dbg) fprintf(stderr, "[stark_hip] commit B=%zu N=2^%u: waited for roots %.0f us, host transcripts %.0f us, enqueue %.0f us over %u rounds%s\n", B, log_N, t_wait, t_host, t_enq, R,
                     helper ? " (enqueue thread)" : "");
    return SH_OK;
}

// ------------------------------------------------------------------ host buffer -> device, pipelined
// A pageable host buffer reaches the device through the runtime's own staging at ~20-29 GB/s (2^24 elements: 9-13 ms) and nothing
// overlaps it; the caller's buffer cannot be pinned in place (hipHostRegister is refused for an ordinary user on the box).  So:
// a ring of pinned slots per context; the context's worker threads copy chunk i+1 of the caller's buffer into a slot (memcpy
// split over the threads) while the copy engine moves chunk i out of its slot on a stream of its own; `after_chunk` puts kernels
// on the context's stream once a chunk has landed.
// The two streams are ordered against each other by the HOST only (hipEventSynchronize on a chunk's copy before its kernels are
// launched; the context's stream is drained before the first copy).  Stream-side waits (hipStreamWaitEvent in both directions) were
// the first form, and hung: streams share the device's few hardware queues (GPU_MAX_HW_QUEUES = 4) once a process has more
// streams than that, a barrier packet holds up everything behind it in its hardware queue whichever stream that belongs to, and a
// barrier that waits for a copy queued behind it in the same hardware queue never ends (native stacks of the stuck process:
// every thread inside a HIP wait; tools/run_with_dump.py).
#define STAGE_SLOTS 4
#define STAGE_SLOT_BYTES ((size_t)8 << 20)
int32_t upload_pipelined(sh_ctx* c, const uint64_t* host, size_t n_elems, fp* d_dst, const std::function<int32_t(size_t, size_t)>& after_chunk) {
    const size_t bytes = n_elems * sizeof(fp);
    static const bool plain = getenv("STARK_HIP_PLAIN_UPLOAD") != nullptr;
    if (plain || bytes < ((size_t)1 << 20)) {                            // small: one copy
        HIPCHK(hipMemcpyAsync(d_dst, host, bytes, hipMemcpyHostToDevice, c->stream));
        return after_chunk ? after_chunk(0, n_elems) : SH_OK;
    }
    if (!c->h_stage) {
        HIPCHK(hipHostMalloc((void**)&c->h_stage, STAGE_SLOTS * STAGE_SLOT_BYTES, hipHostMallocDefault));
        for (int i = 0; i < STAGE_SLOTS; ++i) HIPCHK(hipEventCreateWithFlags(&c->stage_ev[i], hipEventDisableTiming));
        HIPCHK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    }
    if (!c->pool) {
        unsigned hw = std::thread::hardware_concurrency();
        if (hw > 8) hw = 8;
        try { c->pool = new SpinPool(hw > 1 ? hw - 1 : 0); } catch (const std::system_error&) { c->pool = new SpinPool(0); }
    }
    static const unsigned copy_threads = [] { const char* e = getenv("STARK_HIP_UPLOAD_THREADS"); const int v = e ? atoi(e) : 0; return (unsigned)(v >= 1 && v <= 8 ? v : 6); }();
    HIPCHK(hipStreamSynchronize(c->stream));                              // the destination may still be read by work enqueued before
    c->pool->wake(copy_threads);
    struct Parker { SpinPool* p; ~Parker() { p->park(); } } parker{c->pool};
    const size_t chunk_bytes = bytes < 4 * STAGE_SLOT_BYTES ? ((bytes / 4 + 4095) & ~(size_t)4095) : STAGE_SLOT_BYTES;   // a short buffer still pipelines
    const size_t chunk_elems = chunk_bytes / sizeof(fp);
    const uint8_t* src = reinterpret_cast<const uint8_t*>(host);
    size_t slot = 0, used = 0, prev_e0 = 0, prev_ne = 0, prev_slot = 0;
    bool have_prev = false;
    auto land_prev = [&]() -> int32_t {                                   // the previous chunk's copy has finished: its kernels may go
        if (!have_prev) return SH_OK;
        HIPCHK(hipEventSynchronize(c->stage_ev[prev_slot]));
        have_prev = false;
        return after_chunk ? after_chunk(prev_e0, prev_ne) : SH_OK;
    };
    for (size_t e0 = 0; e0 < n_elems; e0 += chunk_elems, slot = (slot + 1) % STAGE_SLOTS, ++used) {
        const size_t ne = n_elems - e0 < chunk_elems ? n_elems - e0 : chunk_elems, nb = ne * sizeof(fp);
        uint8_t* stage = c->h_stage + slot * STAGE_SLOT_BYTES;
        if (used >= STAGE_SLOTS) HIPCHK(hipEventSynchronize(c->stage_ev[slot]));          // the copy that last used this slot has left it
        const uint8_t* from = src + e0 * sizeof(fp);
        c->pool->parallel((nb + 4095) >> 12, [&](size_t p0, size_t p1) {                  // 4 KiB pages, one contiguous slice per thread
            const size_t b0 = p0 << 12, b1 = (p1 << 12) < nb ? (p1 << 12) : nb;
            if (b0 < b1) memcpy(stage + b0, from + b0, b1 - b0);
        });
        HIPCHK(hipMemcpyAsync(d_dst + e0, stage, nb, hipMemcpyHostToDevice, c->copy_stream));
        HIPCHK(hipEventRecord(c->stage_ev[slot], c->copy_stream));
        SHCHK(land_prev());                                              // chunk i-1 was in flight while chunk i was staged
        prev_e0 = e0; prev_ne = ne; prev_slot = slot; have_prev = true;
    }
    return land_prev();                                                  // returns with every copy complete: nothing is left on the copy stream
}
// Upload of ONE codeword for a commit, with round 0's leaf hashing started on every chunk as it lands (the subtree digests of a
// chunk are independent of the rest).  *hashed tells the driver that SCR_DIG_A already holds round 0's digests.
static int32_t upload_codeword_hashing(sh_ctx* c, const uint64_t* host, size_t N, fp* d_dst, bool* hashed) {
    *hashed = false;
    const uint32_t lg = ilog2(N);
    uint32_t K = leaf_subtree_k(lg);
    if (K > lg) K = lg;
    static const bool use_round_wg = getenv("STARK_NO_ROUND_WG") == nullptr;
    const bool small = use_round_wg && N <= ((size_t)1 << round_wg_max_log()) && lg >= 1;      // the fused one-launch round hashes its own leaves
    void* dig = nullptr;
    const bool can = !small && K < lg && (N * sizeof(fp)) >= ((size_t)1 << 20);
    if (can) SHCHK(ensure_scratch(c, SCR_DIG_A, (N >> K) * sizeof(digest64), &dig));
    SHCHK(upload_pipelined(c, host, N, d_dst, [&](size_t e0, size_t ne) -> int32_t {
        if (!can) return SH_OK;
        if ((e0 | ne) & (((size_t)256 << K) - 1)) return sh_fail(SH_EHIP, "upload: chunk not aligned to whole workgroups of subtrees");   // sizes are powers of two: cannot happen
        HIPCHK(launch_leaf_subtree(d_dst + e0, (digest64*)dig + (e0 >> K), ne >> K, K, c->stream));
        return SH_OK;
    }));
    *hashed = can;
    return SH_OK;
}

int32_t fri_commit_dev(sh_ctx* c, const fp* d_cw, size_t N, const sh_fri_params* prm, sh_transcript_cb cb, void* user,
                       uint8_t* roots_out, uint64_t* last_out, size_t* last_len, CommitKeep* keep, bool r0_leaves_hashed) {
    CTXCHK(c);
    if (!d_cw || !cb || !roots_out || !last_out || !last_len) return sh_fail(SH_EINVAL, "fri_commit: null pointer");
    if (!is_pow2(N)) return sh_fail(SH_EINVAL, "fri_commit: N must be a power of two");
    if (ilog2(N) > SH_MAX_LOG_N) return sh_failf(SH_EINVAL, "fri_commit: N > 2^%d", SH_MAX_LOG_N);
    try {
        RoundFn fn = [&](uint32_t r, const uint8_t* roots, bool want, fp* alpha_out) -> int32_t {
            memcpy(roots_out + 64 * (size_t)r, roots, 64);
            uint64_t alpha_w[2] = {0, 0};
            if (cb(user, r, roots_out + 64 * (size_t)r, want ? 1 : 0, alpha_w) != 0) return sh_failf(SH_ECALLBACK, "fri_commit: transcript callback failed in round %u", r);
            if (want) alpha_out[0] = fp_load_host(alpha_w);
            return SH_OK;
        };
        return fri_commit_core(c, d_cw, 1, N, prm, fn, last_out, last_len, keep, r0_leaves_hashed);
    } catch (const std::bad_alloc&) { return sh_fail(SH_ENOMEM, "fri_commit: out of host memory");
    } catch (...) { return sh_fail(SH_ENOMEM, "fri_commit: unexpected exception on the host side"); }
}

// ------------------------------------------------------------------ batched FRI commit (lock-step rounds)
// B independent codewords of equal length, back to back.  Every kernel of a round covers all B codewords, so the
// serial root -> challenge -> fold chain is paid once per round for the whole batch instead of
// once per commit: this is what makes small commits (config 4: 1024 x 2^14) throughput-bound.  Each codeword has its
// own ProofStream; a spin pool advances the B transcripts of a round in parallel.
static int32_t fri_commit_batch_dev(sh_ctx* c, const fp* d_cws, size_t B, size_t N, const sh_fri_params* prm, sh_proofstream** streams,
                                    uint8_t* roots_out, uint64_t* last_out, size_t* last_len) {
    CTXCHK(c);
    if (!d_cws || !streams || !roots_out || !last_out || !last_len) return sh_fail(SH_EINVAL, "fri_commit_batch: null pointer");
    if (B == 0) return sh_fail(SH_EINVAL, "fri_commit_batch: empty batch");
    for (size_t b = 0; b < B; ++b) if (!streams[b]) return sh_fail(SH_EINVAL, "fri_commit_batch: null proof stream");
    if (!is_pow2(N)) return sh_fail(SH_EINVAL, "fri_commit_batch: N must be a power of two");
    const uint32_t log_N = ilog2(N);
    if (log_N > SH_MAX_LOG_N || ilog2(B) + log_N > 30) return sh_fail(SH_EINVAL, "fri_commit_batch: batch too large (B*N must be <= 2^30)");
    if (!prm) return sh_fail(SH_EINVAL, "fri: null params");
    const uint32_t R = sh_fri_num_rounds(N, prm->expansion, prm->num_colinearity_tests);
    const size_t L_last = N >> (R ? R - 1 : 0);
    try {
        // Two lock-step groups on two streams (the context's own and that of an internal second context), each served by its
        // own host thread, for large batches: the groups drift out of phase, so one group's latency-bound stretches (tree
        // tops, root hand-off, transcripts) overlap the other's kernels.  Measured: 1024 x 2^14 6.75 -> 6.3 ms; at 128 x 2^14
        // the two groups stay in phase and the second thread only costs (1.25 -> 1.35 ms), so small batches keep one group.
        static const bool one_group = getenv("STARK_HIP_BATCH_ONE_GROUP") != nullptr;
        const size_t G = (!one_group && B >= 2 && R >= 2 && (B << log_N) >= ((size_t)1 << 23)) ? 2 : 1;   // a few large codewords too: 8 x 2^20, 2 x 2^24
        if (G == 2 && !c->sub) SHCHK(sh_ctx_create(c->device, nullptr, &c->sub));
        unsigned T = std::thread::hardware_concurrency();
        if (T > 8) T = 8;
        if (T > B / 16) T = (unsigned)(B / 16);
        const unsigned Tg = G == 2 ? (T / 2 ? T / 2 : 1) : T;
        auto commit_group = [&](sh_ctx* cx, size_t b_first, size_t Bg, SpinPool* pool) -> int32_t {
            RoundFn fn = [&, b_first, Bg, pool](uint32_t r, const uint8_t* roots, bool want, fp* alpha_out) -> int32_t {
                std::atomic<int> failed{0};
                auto per_codeword = [&](size_t b0, size_t b1) {          // transcripts are independent: one ProofStream per codeword
                    for (size_t b = b0; b < b1; ++b) {
                        const uint8_t* root = roots + 64 * b;
                        sh_proofstream* ps = streams[b_first + b];
                        memcpy(roots_out + 64 * ((b_first + b) * R + r), root, 64);
                        if (sh_proofstream_push_merkle_root(ps, root) != SH_OK) { failed = 1; return; }      // proof_stream.rs:25
                        if (want) {
                            uint8_t dig[32];
                            uint64_t aw[2];
                            if (sh_proofstream_prover_fiat_shamir(ps, dig) != SH_OK) { failed = 1; return; } // proof_stream.rs:51
                            if (sh_sample_field(dig, aw) != SH_OK) { failed = 1; return; }                   // G3
                            alpha_out[b] = fp_load_host(aw);
                        }
                    }
                };
                pool->parallel(Bg, per_codeword);
                return failed ? sh_fail(SH_ECALLBACK, "fri_commit_batch: transcript update failed") : SH_OK;
            };
            size_t ll = 0;
            SHCHK(fri_commit_core(cx, d_cws + b_first * N, Bg, N, prm, fn, last_out + 2 * b_first * L_last, &ll, nullptr));
            return SH_OK;
        };
        // the context's transcript workers (created once, asleep between calls: creating seven threads costs ~120 us per call)
        auto pool_of = [&](sh_ctx* cx) -> SpinPool* {
            if (!cx->pool) {
                unsigned hw = std::thread::hardware_concurrency();
                if (hw > 8) hw = 8;
                try { cx->pool = new SpinPool(hw > 1 ? hw - 1 : 0); } catch (const std::system_error&) { cx->pool = new SpinPool(0); }   // no threads to be had: the caller does it all
            }
            cx->pool->wake(Tg);
            return cx->pool;
        };
        struct Parker { SpinPool* p = nullptr; ~Parker() { if (p) p->park(); } } park0, park1;
        const auto t_core0 = std::chrono::steady_clock::now();
        SpinPool* pool0 = pool_of(c);
        park0.p = pool0;
        if (getenv("STARK_HIP_DEBUG")) fprintf(stderr, "[stark_hip] batch: B=%zu N=2^%u, %zu group(s), %u transcript thread(s) per group\n", B, log_N, G, Tg);
        if (G == 1) SHCHK(commit_group(c, 0, B, pool0));
        else {
            const size_t B0 = B / 2, B1 = B - B0;
            // The second stream starts after whatever produced the codewords — ordered by the host, not by a stream-side wait on an
            // event: streams share hardware queues, and a barrier packet of one stream in front of the other's packets can wait forever.
            HIPCHK(hipStreamSynchronize(c->stream));
            SpinPool* pool1 = pool_of(c->sub);
            park1.p = pool1;
            int32_t rc1 = SH_OK;
            std::string msg1;
            bool threaded = true;
            std::thread second;
            struct JoinOnExit { std::thread& t; ~JoinOnExit() { if (t.joinable()) t.join(); } } join_on_exit{second};   // also when this thread unwinds
            try {
                second = std::thread([&] {
                    try {
                        if (hipSetDevice(c->device) != hipSuccess) { rc1 = SH_EHIP; msg1 = "hipSetDevice failed in the second group's thread"; return; }
                        rc1 = commit_group(c->sub, B0, B1, pool1);
                        if (rc1 != SH_OK) msg1 = sh_last_error();
                    } catch (...) { rc1 = SH_ENOMEM; msg1 = "fri_commit_batch: unexpected exception in the second group's thread"; }
                });
            } catch (const std::system_error&) { threaded = false; }
            const int32_t rc0 = commit_group(c, 0, B0, pool0);
            if (threaded) second.join();
            else rc1 = commit_group(c->sub, B0, B1, pool0);           // no thread to be had: one group after the other
            if (rc0 != SH_OK) return rc0;
            if (rc1 != SH_OK) return threaded ? sh_fail(rc1, msg1.c_str()) : rc1;
        }
        *last_len = L_last;
        const auto t_tail0 = std::chrono::steady_clock::now();
        std::atomic<int> failed{0};
        pool0->parallel(B, [&](size_t b0, size_t b1) { for (size_t b = b0; b < b1; ++b) if (sh_proofstream_push_codeword(streams[b], last_out + 2 * b * L_last, L_last) != SH_OK) failed = 1; });
        if (failed) return sh_fail(SH_ECALLBACK, "fri_commit_batch: pushing the last codeword failed");
        if (getenv("STARK_HIP_DEBUG")) fprintf(stderr, "[stark_hip] batch: %zu group(s), rounds %.0f us, pushing the last codewords %.0f us\n", G,
                                               std::chrono::duration<double, std::micro>(t_tail0 - t_core0).count(),
                                               std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_tail0).count());
        return SH_OK;
    } catch (const std::bad_alloc&) { return sh_fail(SH_ENOMEM, "fri_commit_batch: out of host memory");
    } catch (...) { return sh_fail(SH_ENOMEM, "fri_commit_batch: unexpected exception on the host side"); }
}

int32_t ps_callback(void* user, uint32_t, const uint8_t root[64], int32_t want_alpha, uint64_t alpha_out[2]) {
    PsCbState* st = (PsCbState*)user;
    if (sh_proofstream_push_merkle_root(st->ps, root) != SH_OK) return 1;        // proof_stream.rs:25
    if (want_alpha) {
        uint8_t dig[32];
        if (sh_proofstream_prover_fiat_shamir(st->ps, dig) != SH_OK) return 1;    // proof_stream.rs:51
        if (sh_sample_field(dig, alpha_out) != SH_OK) return 1;                  // G3
    }
    return 0;
}

extern "C" {

int32_t sh_dev_fri_fold(sh_ctx* c, const uint64_t* d_cw, size_t L, const uint64_t alpha[2], const uint64_t offset[2],
                        const uint64_t omega[2], uint64_t* d_out) {
    CTXCHK(c);
    if (!d_cw || !alpha || !offset || !omega || !d_out) return sh_fail(SH_EINVAL, "sh_dev_fri_fold: null pointer");
    if (!is_pow2(L) || L < 2) return sh_fail(SH_EINVAL, "sh_fri_fold: L must be a power of two >= 2");
    const fp a = fp_load_host(alpha), off = fp_load_host(offset), w = fp_load_host(omega);
    if (!fp_is_canonical(a) || !fp_is_canonical(off) || !fp_is_canonical(w)) return sh_fail(SH_EINVAL, "sh_fri_fold: non-canonical scalar");
    if (fp_is_zero(off) || fp_is_zero(w)) return sh_fail(SH_EDIVZERO, "Cannot divide by zero");      // alpha/(offset·omega^i), field.rs:102
    const fp c0 = fp_mul(fp_mul(a, fp_inv(fp_make(2, 0, 0, 0))), fp_inv(off));
    HIPCHK(launch_fri_fold((const fp*)d_cw, (const fp*)d_cw + L / 2, L / 2, fp_to_mont(c0), fp_inv(w), (fp*)d_out, c->stream));
    return SH_OK;
}
int32_t sh_dev_fri_fold_range(sh_ctx* c, const uint64_t* d_lo, const uint64_t* d_hi, size_t count, uint64_t i0, const uint64_t alpha[2],
                              const uint64_t offset[2], const uint64_t omega[2], uint64_t* d_out) {
    CTXCHK(c);
    if (!count) return SH_OK;
    if (!d_lo || !d_hi || !alpha || !offset || !omega || !d_out) return sh_fail(SH_EINVAL, "sh_dev_fri_fold_range: null pointer");
    const fp a = fp_load_host(alpha), off = fp_load_host(offset), w = fp_load_host(omega);
    if (!fp_is_canonical(a) || !fp_is_canonical(off) || !fp_is_canonical(w)) return sh_fail(SH_EINVAL, "sh_fri_fold_range: non-canonical scalar");
    if (fp_is_zero(off) || fp_is_zero(w)) return sh_fail(SH_EDIVZERO, "Cannot divide by zero");      // field.rs:102
    const fp winv = fp_inv(w);
    const fp c0 = fp_mul(fp_mul(fp_mul(a, fp_inv(fp_make(2, 0, 0, 0))), fp_inv(off)), fp_pow(winv, i0));
    HIPCHK(launch_fri_fold((const fp*)d_lo, (const fp*)d_hi, count, fp_to_mont(c0), winv, (fp*)d_out, c->stream));
    return SH_OK;
}
int32_t sh_dev_scale_powers(sh_ctx* c, uint64_t* d_data, size_t count, uint64_t j0, const uint64_t base[2]) {
    CTXCHK(c);
    if (!count) return SH_OK;
    if (!d_data || !base) return sh_fail(SH_EINVAL, "sh_dev_scale_powers: null pointer");
    const fp b = fp_load_host(base);
    if (!fp_is_canonical(b)) return sh_fail(SH_EINVAL, "sh_dev_scale_powers: non-canonical base");
    HIPCHK(launch_scale_powers((fp*)d_data, count, fp_to_mont(fp_pow(b, j0)), b, c->stream));
    return SH_OK;
}
int32_t sh_fri_fold(sh_ctx* c, const uint64_t* cw, size_t L, const uint64_t alpha[2], const uint64_t offset[2],
                    const uint64_t omega[2], uint64_t* out) {
    CTXCHK(c);
    if (!cw || !out) return sh_fail(SH_EINVAL, "sh_fri_fold: null pointer");
    if (!is_pow2(L) || L < 2) return sh_fail(SH_EINVAL, "sh_fri_fold: L must be a power of two >= 2");
    void *din, *dout;
    SHCHK(ensure_scratch(c, SCR_IO_A, L * 16, &din));
    SHCHK(ensure_scratch(c, SCR_IO_B, L * 8, &dout));
    HIPCHK(hipMemcpyAsync(din, cw, L * 16, hipMemcpyHostToDevice, c->stream));
    SHCHK(sh_dev_fri_fold(c, (const uint64_t*)din, L, alpha, offset, omega, (uint64_t*)dout));
    HIPCHK(hipMemcpyAsync(out, dout, L * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SH_OK;
}

uint32_t sh_fri_num_rounds(size_t N, uint32_t expansion, uint32_t num_colinearity_tests) {
    uint32_t r = 0;
    size_t len = N;
    while (len > expansion && 4 * (size_t)num_colinearity_tests < len) { len /= 2; ++r; }
    return r;
}
int32_t sh_dev_fri_commit(sh_ctx* c, const uint64_t* d_codeword, size_t N, const sh_fri_params* params, sh_transcript_cb cb, void* user,
                          uint8_t* roots_out, uint64_t* last_codeword_out, size_t* last_len) {
    return fri_commit_dev(c, (const fp*)d_codeword, N, params, cb, user, roots_out, last_codeword_out, last_len);
}
int32_t sh_fri_commit(sh_ctx* c, const uint64_t* codeword, size_t N, const sh_fri_params* params, sh_transcript_cb cb, void* user,
                      uint8_t* roots_out, uint64_t* last_codeword_out, size_t* last_len) {
    CTXCHK(c);
    if (!codeword) return sh_fail(SH_EINVAL, "sh_fri_commit: null pointer");
    if (!is_pow2(N) || ilog2(N) > SH_MAX_LOG_N) return sh_fail(SH_EINVAL, "sh_fri_commit: N must be a power of two <= 2^28");
    void* d;
    SHCHK(ensure_scratch(c, SCR_IO_C, N * 16, &d));
    bool hashed = false;
    SHCHK(upload_codeword_hashing(c, codeword, N, (fp*)d, &hashed));
    return fri_commit_dev(c, (const fp*)d, N, params, cb, user, roots_out, last_codeword_out, last_len, nullptr, hashed);
}
int32_t sh_dev_fri_commit_proofstream(sh_ctx* c, const uint64_t* d_codeword, size_t N, const sh_fri_params* params, sh_proofstream* ps,
                                      uint8_t* roots_out, uint64_t* last_codeword_out, size_t* last_len) {
    if (!ps) return sh_fail(SH_EINVAL, "sh_fri_commit_proofstream: null proof stream");
    PsCbState st{ps};
    SHCHK(fri_commit_dev(c, (const fp*)d_codeword, N, params, ps_callback, &st, roots_out, last_codeword_out, last_len));
    return sh_proofstream_push_codeword(ps, last_codeword_out, *last_len);
}
int32_t sh_fri_commit_proofstream(sh_ctx* c, const uint64_t* codeword, size_t N, const sh_fri_params* params, sh_proofstream* ps,
                                  uint8_t* roots_out, uint64_t* last_codeword_out, size_t* last_len) {
    if (!ps) return sh_fail(SH_EINVAL, "sh_fri_commit_proofstream: null proof stream");
    PsCbState st{ps};
    SHCHK(sh_fri_commit(c, codeword, N, params, ps_callback, &st, roots_out, last_codeword_out, last_len));
    return sh_proofstream_push_codeword(ps, last_codeword_out, *last_len);
}

int32_t sh_dev_fri_commit_batch(sh_ctx* c, const uint64_t* d_codewords, size_t B, size_t N, const sh_fri_params* params, sh_proofstream** streams,
                                uint8_t* roots_out, uint64_t* last_codewords_out, size_t* last_len) {
    return fri_commit_batch_dev(c, (const fp*)d_codewords, B, N, params, streams, roots_out, last_codewords_out, last_len);
}
int32_t sh_fri_commit_batch(sh_ctx* c, const uint64_t* codewords, size_t B, size_t N, const sh_fri_params* params, sh_proofstream** streams,
                            uint8_t* roots_out, uint64_t* last_codewords_out, size_t* last_len) {
    CTXCHK(c);
    if (!codewords) return sh_fail(SH_EINVAL, "sh_fri_commit_batch: null pointer");
    if (B == 0 || !is_pow2(N) || ilog2(N) > SH_MAX_LOG_N || ilog2(B) + ilog2(N) > 30) return sh_fail(SH_EINVAL, "sh_fri_commit_batch: bad batch shape");
    void* d;
    SHCHK(ensure_scratch(c, SCR_IO_C, B * N * 16, &d));
    SHCHK(upload_pipelined(c, codewords, B * N, (fp*)d, nullptr));
    return fri_commit_batch_dev(c, (const fp*)d, B, N, params, streams, roots_out, last_codewords_out, last_len);
}

}  // extern "C"
