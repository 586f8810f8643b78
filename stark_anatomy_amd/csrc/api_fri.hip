// api_fri.hip — FRI drivers behind include/stark_hip.h: split-and-fold entry points, the commit loop (single codeword and
// lock-step batches), the library-side transcript callback.
#include "api_internal.h"

// ------------------------------------------------------------------ FRI
int32_t check_params(const sh_fri_params* p, fp* offset) {
    if (!p) return sh_fail(SH_EINVAL, "fri: null params");
    if (p->expansion == 0) return sh_fail(SH_EINVAL, "fri: expansion must be >= 1");
    *offset = fp_load_host(p->offset);
    if (!fp_is_canonical(*offset) || fp_is_zero(*offset)) return sh_fail(SH_EINVAL, "fri: offset must be a non-zero canonical field element");
    return SH_OK;
}
// `keep`: when set, every round's codeword stays resident (round r >= 1 at offset N - N/2^(r-1) of one N-element arena,
// instead of the two ping-pong buffers)  and its device pointer is recorded — the query phase opens them afterwards.
int32_t fri_commit_dev(sh_ctx* c, const fp* d_cw, size_t N, const sh_fri_params* prm, sh_transcript_cb cb, void* user,
                              uint8_t* roots_out, uint64_t* last_out, size_t* last_len, std::vector<const fp*>* keep) {
    CTXCHK(c);
    if (!d_cw || !cb || !roots_out || !last_out || !last_len) return sh_fail(SH_EINVAL, "fri_commit: null pointer");
    if (!is_pow2(N)) return sh_fail(SH_EINVAL, "fri_commit: N must be a power of two");
    const uint32_t log_N = ilog2(N);
    if (log_N > SH_MAX_LOG_N) return sh_failf(SH_EINVAL, "fri_commit: N > 2^%d", SH_MAX_LOG_N);
    fp off;
    SHCHK(check_params(prm, &off));
    const uint32_t R = sh_fri_num_rounds(N, prm->expansion, prm->num_colinearity_tests);
    const fp* cur = d_cw;
    size_t L = N;
    if (R > 0) {
        const fp* tw_inv;
        SHCHK(get_twiddles(c, log_N, true, &tw_inv));
        void *bufX = nullptr, *bufY = nullptr, *arena = nullptr;
        if (keep) { keep->clear(); keep->push_back(d_cw); if (R > 1) SHCHK(ensure_scratch(c, SCR_ARENA, N * sizeof(fp), &arena)); }
        else {
            if (R > 1) SHCHK(ensure_scratch(c, SCR_IO_A, (N / 2) * sizeof(fp), &bufX));
            if (R > 2) SHCHK(ensure_scratch(c, SCR_IO_B, (N / 4) * sizeof(fp), &bufY));
        }
        const fp inv2 = fp_inv(fp_make(2, 0, 0, 0));
        fp off_inv = fp_inv(off);                  // (off^(2^r))^-1 = (off^-1)^(2^r): one inversion for the whole commit
        uint8_t root[64];
        SHCHK(merkle_root_from_elems(c, cur, L));
        SHCHK(fetch_root(c, root));
        const bool dbg = getenv("STARK_HIP_DEBUG") != nullptr;     // per-round host timing: callback / enqueue / wait for the root
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        for (uint32_t r = 0; r < R; ++r) {
            memcpy(roots_out + 64 * (size_t)r, root, 64);
            const int32_t want = (r + 1 < R) ? 1 : 0;
            uint64_t alpha_w[2] = {0, 0};
            const auto t0 = now();
            if (cb(user, r, root, want, alpha_w) != 0) return sh_failf(SH_ECALLBACK, "fri_commit: transcript callback failed in round %u", r);
            if (!want) break;
            const fp alpha = fp_load_host(alpha_w);
            if (!fp_is_canonical(alpha)) return sh_fail(SH_EINVAL, "fri_commit: callback returned a non-canonical alpha");
            // c0 = alpha · 2^-1 · offset^-1, kept in Montgomery form for the kernel
            const fp c0 = fp_mul(fp_mul(alpha, inv2), off_inv);
            const size_t half = L / 2;
            fp* next = keep ? (fp*)arena + (N - L) : (fp*)((r & 1) ? bufY : bufX);         // arena: round r+1 at offset N - N/2^r (L = N/2^r)
            if (keep) keep->push_back(next);
            uint32_t K = leaf_subtree_k(ilog2(half));
            void* dig;
            SHCHK(ensure_scratch(c, SCR_DIG_A, (half >> K) * sizeof(digest64), &dig));
            const auto t1 = now();
            HIPCHK(launch_fold_leaf_subtree(cur, L, 1, nullptr, fp_to_mont(c0), tw_inv, log_N, r, next, (digest64*)dig, K, c->stream));
            const auto t2 = now();
            SHCHK(reduce_digests(c, (const digest64*)dig, half >> K));
            const auto t3 = now();
            SHCHK(fetch_root(c, root));
            if (dbg) fprintf(stderr, "[stark_hip] round %2u L=2^%-2u  cb+c0 %5.1f us | launch fold %5.1f | launch reduce %5.1f | wait root %6.1f\n",
                             r, ilog2(L), us(t0, t1), us(t1, t2), us(t2, t3), us(t3, now()));
            cur = next; L = half;
            off_inv = fp_mul(off_inv, off_inv);
        }
    }
    HIPCHK(hipMemcpyAsync(last_out, cur, L * sizeof(fp), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    *last_len = L;
    return SH_OK;
}

// Worker threads that live for one batched commit and are handed one slice of the B transcripts per round.  They spin
// (then yield) between rounds: a round's hand-off must cost well under the ~10 us a thread spawn does.
class SpinPool {
  public:
    explicit SpinPool(unsigned workers) {
        for (unsigned w = 0; w < workers; ++w) th_.emplace_back([this, w] { run(w); });
    }
    ~SpinPool() {
        stop_.store(true, std::memory_order_relaxed);
        gen_.fetch_add(1, std::memory_order_release);
        for (auto& t : th_) t.join();
    }
    // f(b0, b1) over [0, n) split into workers + 1 contiguous slices; the caller takes the first
    template <class F>
    void parallel(size_t n, F&& f) {
        const size_t T = th_.size() + 1, chunk = (n + T - 1) / T;
        std::function<void(size_t, size_t)> fn = f;
        fn_ = &fn; n_ = n; chunk_ = chunk;
        pending_.store((int)th_.size(), std::memory_order_relaxed);
        gen_.fetch_add(1, std::memory_order_release);
        fn(0, chunk < n ? chunk : n);
        while (pending_.load(std::memory_order_acquire) != 0) __builtin_ia32_pause();
    }
  private:
    void run(unsigned w) {
        uint64_t seen = 0;
        for (;;) {
            uint32_t spins = 0;
            while (gen_.load(std::memory_order_acquire) == seen) { if (++spins > 4000) std::this_thread::yield(); else __builtin_ia32_pause(); }
            ++seen;
            if (stop_.load(std::memory_order_relaxed)) return;
            const size_t b0 = (size_t)(w + 1) * chunk_, b1 = b0 + chunk_;
            (*fn_)(b0 < n_ ? b0 : n_, b1 < n_ ? b1 : n_);
            pending_.fetch_sub(1, std::memory_order_release);
        }
    }
    std::vector<std::thread> th_;
    std::atomic<uint64_t> gen_{0};
    std::atomic<int> pending_{0};
    std::atomic<bool> stop_{false};
    const std::function<void(size_t, size_t)>* fn_ = nullptr;
    size_t n_ = 0, chunk_ = 0;
};

// ------------------------------------------------------------------ batched FRI commit (lock-step rounds)
// B independent codewords of equal length, back to back.  Every kernel of a round covers all B codewords, so the
// serial root -> challenge -> fold chain is paid once per round for the whole batch instead of
// once per commit: this is what makes small commits (config 4: 1024 x 2^14) throughput-bound.
// `via_mail`: when set and the last step is a quad reduction, the B roots go straight to the pinned host buffer
// (c->d_hbatch) and the mailbox sequence number announces them (*via_mail = true); otherwise they land in d_roots.
static int32_t reduce_digests_batch(sh_ctx* c, const digest64* cur, size_t count, size_t B, digest64* d_roots, bool* via_mail = nullptr) {
    uint32_t lg = ilog2(count);
    if (via_mail) *via_mail = false;
    if (lg == 0) {
        HIPCHK(hipMemcpyAsync(d_roots, cur, B * sizeof(digest64), hipMemcpyDeviceToDevice, c->stream));
        return SH_OK;
    }
    bool use_b = true;
    while (lg > 0) {
        const size_t total = B << lg;
        const bool wide = total > ((size_t)1 << QUAD_MAX_LOG_COUNT);
        uint32_t step;
        if (wide) {                            // lane-per-subtree levels only down to what the quad kernel takes (2^17 digests in all)
            const uint32_t over = ilog2(total) - QUAD_MAX_LOG_COUNT;
            step = over < 3 ? over : 3;
            if (step > lg) step = lg;
        } else step = lg < QUAD_MAX_LOG_CHUNK ? lg : QUAD_MAX_LOG_CHUNK;
        digest64* out = d_roots;
        if (lg != step) {
            void* b;
            SHCHK(ensure_scratch(c, use_b ? SCR_DIG_B : SCR_DIG_A, (total >> step) * sizeof(digest64), &b));
            use_b = !use_b;
            out = (digest64*)b;
        }
        if (wide) HIPCHK(launch_node_subtree(cur, out, total >> step, step, c->stream));
        else if (lg == step && via_mail && c->d_hbatch) {
            if (++c->mail_seq == 0) c->mail_seq = 1;
            c->batch_done_total += (uint32_t)B;
            HIPCHK(launch_quad_reduce(cur, (digest64*)c->d_hbatch, total, step, c->d_mail, c->mail_seq, c->stream, c->d_flag + 16, c->batch_done_total));
            *via_mail = true;
        } else HIPCHK(launch_quad_reduce(cur, out, total, step, nullptr, 0, c->stream));
        cur = out; lg -= step;
    }
    return SH_OK;
}

static int32_t fri_commit_batch_dev(sh_ctx* c, const fp* d_cws, size_t B, size_t N, const sh_fri_params* prm, sh_proofstream** streams,
                                    uint8_t* roots_out, uint64_t* last_out, size_t* last_len) {
    CTXCHK(c);
    if (!d_cws || !streams || !roots_out || !last_out || !last_len) return sh_fail(SH_EINVAL, "fri_commit_batch: null pointer");
    if (B == 0) return sh_fail(SH_EINVAL, "fri_commit_batch: empty batch");
    for (size_t b = 0; b < B; ++b) if (!streams[b]) return sh_fail(SH_EINVAL, "fri_commit_batch: null proof stream");
    if (!is_pow2(N)) return sh_fail(SH_EINVAL, "fri_commit_batch: N must be a power of two");
    const uint32_t log_N = ilog2(N);
    if (log_N > SH_MAX_LOG_N || ilog2(B) + log_N > 30) return sh_fail(SH_EINVAL, "fri_commit_batch: batch too large (B*N must be <= 2^30)");
    fp off;
    SHCHK(check_params(prm, &off));
    const uint32_t R = sh_fri_num_rounds(N, prm->expansion, prm->num_colinearity_tests);
    const fp* cur = d_cws;
    size_t L = N;
    if (R > 0) {
        const fp* tw_inv;
        SHCHK(get_twiddles(c, log_N, true, &tw_inv));
        void *bufX = nullptr, *bufY = nullptr, *d_roots, *d_c0;
        if (R > 1) SHCHK(ensure_scratch(c, SCR_IO_A, B * (N / 2) * sizeof(fp), &bufX));
        if (R > 2) SHCHK(ensure_scratch(c, SCR_IO_B, B * (N / 4) * sizeof(fp), &bufY));
        SHCHK(ensure_scratch(c, SCR_ROOTS, B * sizeof(digest64), &d_roots));
        SHCHK(ensure_scratch(c, SCR_C0, B * sizeof(fp), &d_c0));
        const size_t L_last = N >> (R - 1);
        const size_t need = B * 64 + B * 16 + B * L_last * sizeof(fp);     // roots | c0 | last codewords
        if (c->h_batch_bytes < need) {
            if (c->h_batch) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipHostFree(c->h_batch)); c->h_batch = nullptr; c->d_hbatch = nullptr; c->h_batch_bytes = 0; }
            HIPCHK(hipHostMalloc((void**)&c->h_batch, need, hipHostMallocMapped | hipHostMallocCoherent));
            HIPCHK(hipHostGetDevicePointer((void**)&c->d_hbatch, c->h_batch, 0));
            c->h_batch_bytes = need;
        }
        uint8_t* h_roots = c->h_batch;
        fp* h_c0 = reinterpret_cast<fp*>(c->h_batch + B * 64);
        const fp* d_c0_host = reinterpret_cast<const fp*>(c->d_hbatch + B * 64);   // the fold kernel reads c0[b] straight from pinned memory
        bool via_mail = false;
        (void)d_c0;
        const fp inv2 = fp_inv(fp_make(2, 0, 0, 0));
        fp off_inv = fp_inv(off);
        auto leaf_k = [&](size_t len) {                 // per-lane subtree depth: fill the chip first, never cross a codeword
            uint32_t K = leaf_subtree_k(ilog2(B * len));
            const uint32_t lg = ilog2(len);
            return K < lg ? K : lg;
        };
        {
            const uint32_t K = leaf_k(L);
            void* dig;
            SHCHK(ensure_scratch(c, SCR_DIG_A, ((B * L) >> K) * sizeof(digest64), &dig));
            HIPCHK(launch_leaf_subtree(cur, (digest64*)dig, (B * L) >> K, K, c->stream));
            SHCHK(reduce_digests_batch(c, (const digest64*)dig, L >> K, B, (digest64*)d_roots, &via_mail));
        }
        unsigned T = std::thread::hardware_concurrency();
        if (T > 8) T = 8;
        if (T > B / 16) T = (unsigned)(B / 16);
        SpinPool pool(T > 1 ? T - 1 : 0);
        const bool dbg = getenv("STARK_HIP_DEBUG") != nullptr;
        double t_wait = 0, t_host = 0, t_launch = 0;
        auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        for (uint32_t r = 0; r < R; ++r) {
            const double t0 = now();
            bool got = false;
            if (via_mail) SHCHK(wait_mailbox(c, &got));
            if (!got) {
                if (via_mail) ++c->mail_fallbacks;
                if (!via_mail) HIPCHK(hipMemcpyAsync(h_roots, d_roots, B * 64, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(hipStreamSynchronize(c->stream));          // (via_mail: the stream drained, so the roots are already in h_roots)
            }
            const double t1 = now();
            t_wait += t1 - t0;
            const bool want = r + 1 < R;
            std::atomic<int> failed{0};
            auto per_codeword = [&](size_t b0, size_t b1) {          // transcripts are independent: one ProofStream per codeword
                for (size_t b = b0; b < b1; ++b) {
                    const uint8_t* root = h_roots + 64 * b;
                    memcpy(roots_out + 64 * (b * R + r), root, 64);
                    if (sh_proofstream_push_merkle_root(streams[b], root) != SH_OK) { failed = 1; return; }      // proof_stream.rs:25
                    if (want) {
                        uint8_t dig[32];
                        uint64_t aw[2];
                        if (sh_proofstream_prover_fiat_shamir(streams[b], dig) != SH_OK) { failed = 1; return; } // proof_stream.rs:51
                        if (sh_sample_field(dig, aw) != SH_OK) { failed = 1; return; }                           // G3
                        h_c0[b] = fp_to_mont(fp_mul(fp_mul(fp_load_host(aw), inv2), off_inv));
                    }
                }
            };
            pool.parallel(B, per_codeword);
            if (failed) return sh_fail(SH_ECALLBACK, "fri_commit_batch: transcript update failed");
            t_host += now() - t1;
            if (!want) { if (dbg) fprintf(stderr, "[stark_hip] batch B=%zu N=%zu: wait %.0f us, host transcript %.0f us, launches %.0f us over %u rounds\n", B, N, t_wait, t_host, t_launch, R); break; }
            const double t2 = now();
            const size_t half = L / 2;
            fp* next = (fp*)((r & 1) ? bufY : bufX);
            const uint32_t K = leaf_k(half);
            void* dig;
            SHCHK(ensure_scratch(c, SCR_DIG_A, ((B * half) >> K) * sizeof(digest64), &dig));
            HIPCHK(launch_fold_leaf_subtree(cur, L, B, d_c0_host, fp_zero(), tw_inv, log_N, r, next, (digest64*)dig, K, c->stream));
            SHCHK(reduce_digests_batch(c, (const digest64*)dig, half >> K, B, (digest64*)d_roots, &via_mail));
            cur = next; L = half;
            off_inv = fp_mul(off_inv, off_inv);
            t_launch += now() - t2;
        }
    }
    const auto t_tail0 = std::chrono::steady_clock::now();
    if (R > 0 && c->h_batch && c->h_batch_bytes >= B * 80 + B * L * sizeof(fp)) {       // through pinned staging: a pageable D2H is several times slower
        uint8_t* stage = c->h_batch + B * 80;
        HIPCHK(hipMemcpyAsync(stage, cur, B * L * sizeof(fp), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        memcpy(last_out, stage, B * L * sizeof(fp));
    } else {
        HIPCHK(hipMemcpyAsync(last_out, cur, B * L * sizeof(fp), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    *last_len = L;
    {
        std::atomic<int> failed{0};
        auto push = [&](size_t b0, size_t b1) { for (size_t b = b0; b < b1; ++b) if (sh_proofstream_push_codeword(streams[b], last_out + 2 * b * L, L) != SH_OK) failed = 1; };
        unsigned T = std::thread::hardware_concurrency();
        if (T > 16) T = 16;
        if (T > B / 16) T = (unsigned)(B / 16);
        if (T <= 1) push(0, B);
        else {
            std::vector<std::thread> th;
            const size_t chunk = (B + T - 1) / T;
            for (unsigned t = 1; t < T; ++t) th.emplace_back(push, t * chunk < B ? t * chunk : B, (t + 1) * chunk < B ? (t + 1) * chunk : B);
            push(0, chunk < B ? chunk : B);
            for (auto& x : th) x.join();
        }
        if (failed) return sh_fail(SH_ECALLBACK, "fri_commit_batch: pushing the last codeword failed");
    }
    if (getenv("STARK_HIP_DEBUG")) fprintf(stderr, "[stark_hip] batch tail (last codewords to host + push): %.0f us\n",
                                           std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_tail0).count());
    return SH_OK;
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
    HIPCHK(hipMemcpyAsync(d, codeword, N * 16, hipMemcpyHostToDevice, c->stream));
    return fri_commit_dev(c, (const fp*)d, N, params, cb, user, roots_out, last_codeword_out, last_len);
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
    HIPCHK(hipMemcpyAsync(d, codewords, B * N * 16, hipMemcpyHostToDevice, c->stream));
    return fri_commit_batch_dev(c, (const fp*)d, B, N, params, streams, roots_out, last_codewords_out, last_len);
}

}  // extern "C"
