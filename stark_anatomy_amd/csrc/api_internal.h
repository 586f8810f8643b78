// api_internal.h — shared by the api_*.hip translation units (the host drivers behind include/stark_hip.h): the context,
// error macros, small host helpers and the internal entry points one unit borrows from another.  Internal.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <functional>
#include <map>
#include <memory>
#include <system_error>
#include <new>
#include <atomic>
#include <string>
#include <thread>
#include <vector>
#include "../../include/stark_hip.h"
#include "host_internal.h"
#include "kernels.h"


#define HIPCHK(expr)                                                                                           \
    do {                                                                                                       \
        hipError_t _e = (expr);                                                                                \
        if (_e != hipSuccess) return sh_failf(SH_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)
#define SHCHK(expr) do { int32_t _rc = (expr); if (_rc != SH_OK) return _rc; } while (0)

#define CTXCHK(c)                                                          \
    do {                                                                   \
        if (!(c)) return sh_fail(SH_EINVAL, "null sh_ctx");                \
        HIPCHK(hipSetDevice((c)->device));                                 \
    } while (0)

// Worker threads of a context that are handed one slice of the B transcripts of a batched commit per round.  Between wake(use)
// and park() the first use - 1 of them spin (then yield) between rounds: a round's hand-off must cost well under the ~10 us a
// wake-up does.  The others, and all of them between calls, sleep on a condition variable: a spinning thread that has no slice
// to work on only takes a core from the threads that serve the chains (the box gives a process 16; two lock-step groups with
// seven idle spinners each ran 2.4 - 7 ms per 8 x 2^20 batch, call by call).
class SpinPool {
  public:
    explicit SpinPool(unsigned workers) {
        for (unsigned w = 0; w < workers; ++w) th_.emplace_back([this, w] { run(w); });
    }
    ~SpinPool() {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; awake_.store(false, std::memory_order_release); }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    // `use` threads (the caller included) share the work until park()
    void wake(unsigned use) {
        {
            std::lock_guard<std::mutex> g(m_);
            use_ = use < 1 ? 1 : (use > th_.size() + 1 ? (unsigned)th_.size() + 1 : use);
            base_gen_ = gen_.load(std::memory_order_relaxed);
            awake_.store(true, std::memory_order_release);
        }
        if (use_ > 1) cv_.notify_all();
    }
    void park() { awake_.store(false, std::memory_order_release); }
    // f(b0, b1) over [0, n) split into `use` contiguous slices; the caller takes the first
    template <class F>
    void parallel(size_t n, F&& f) {
        const size_t T = use_;
        if (T <= 1) { f((size_t)0, n); return; }
        const size_t chunk = (n + T - 1) / T;
        std::function<void(size_t, size_t)> fn = f;
        fn_ = &fn; n_ = n; chunk_ = chunk;
        pending_.store((int)T - 1, std::memory_order_relaxed);
        gen_.fetch_add(1, std::memory_order_release);
        fn(0, chunk < n ? chunk : n);
        while (pending_.load(std::memory_order_acquire) != 0) __builtin_ia32_pause();
    }
  private:
    void run(unsigned w) {
        for (;;) {
            uint64_t seen;
            {   // asleep until a call wants this worker; jobs are counted from the generation at that call's wake()
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [this, w] { return stop_ || (awake_.load(std::memory_order_acquire) && w + 1 < use_); });
                if (stop_) return;
                seen = base_gen_;
            }
            for (;;) {
                uint32_t spins = 0;
                bool parked = false;
                while (gen_.load(std::memory_order_acquire) == seen) {
                    if (!awake_.load(std::memory_order_acquire)) { parked = true; break; }
                    if (++spins > 4000) std::this_thread::yield();
                    else __builtin_ia32_pause();
                }
                if (parked) break;
                ++seen;
                const size_t b0 = (size_t)(w + 1) * chunk_, b1 = b0 + chunk_;
                (*fn_)(b0 < n_ ? b0 : n_, b1 < n_ ? b1 : n_);
                pending_.fetch_sub(1, std::memory_order_release);
            }
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_;
    std::atomic<uint64_t> gen_{0};
    uint64_t base_gen_ = 0;
    std::atomic<int> pending_{0};
    std::atomic<bool> awake_{false};
    bool stop_ = false;
    const std::function<void(size_t, size_t)>* fn_ = nullptr;
    size_t n_ = 0, chunk_ = 0;
    unsigned use_ = 1;
};

// The thread of a context that makes the HIP calls of a commit's rounds (api_fri.hip).  The caller's thread serves the
// root -> challenge chain and must never sit inside the runtime while a kernel on the GPU waits for it: a launch can block
// behind another thread's hipFree / context teardown, which itself waits for the device to go idle — i.e. for the very
// kernel that waits for this host thread.  So the launches go through this thread, which nothing on the
// GPU depends on.  Asleep on a condition variable between commits.
class ChainEnqueuer {
  public:
    explicit ChainEnqueuer(int device) : device_(device), th_([this] { run(); }) {}
    ~ChainEnqueuer() {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; }
        cv_.notify_all();
        th_.join();
    }
    // job() runs on the helper thread; one job at a time (post, then wait, then post again)
    void post(std::function<int32_t()> job) {
        { std::lock_guard<std::mutex> g(m_); job_ = std::move(job); done_.store(false, std::memory_order_relaxed); has_job_ = true; }
        cv_.notify_one();
    }
    bool done() const { return done_.load(std::memory_order_acquire); }
    bool failed() const { return done() && rc_ != SH_OK; }
    // blocks until the posted job has returned; its status, and its error text in *msg
    int32_t wait(std::string* msg) {
        uint32_t spins = 0;
        while (!done()) { if (++spins > 2000) std::this_thread::yield(); else __builtin_ia32_pause(); }
        if (rc_ != SH_OK && msg) *msg = msg_;
        return rc_;
    }
  private:
    void run() {
        (void)hipSetDevice(device_);
        for (;;) {
            std::function<int32_t()> job;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [this] { return stop_ || has_job_; });
                if (stop_) return;
                job = std::move(job_);
                has_job_ = false;
            }
            int32_t rc;
            try { rc = job(); } catch (...) { rc = sh_fail(SH_ENOMEM, "fri_commit: exception in the enqueue thread"); }
            rc_ = rc;
            if (rc != SH_OK) msg_ = sh_last_error();
            done_.store(true, std::memory_order_release);
        }
    }
    int device_;
    std::mutex m_;
    std::condition_variable cv_;
    std::function<int32_t()> job_;
    bool has_job_ = false, stop_ = false;
    std::atomic<bool> done_{true};
    int32_t rc_ = SH_OK;
    std::string msg_;
    std::thread th_;                             // last member: the thread starts with everything above constructed
};

// ------------------------------------------------------------------ context
#define NUM_SCRATCH 20
enum { SCR_IO_A = 0, SCR_IO_B = 1, SCR_IO_C = 2, SCR_WORK = 3, SCR_DIG_A = 4, SCR_DIG_B = 5, SCR_ROOTS = 6, SCR_C0 = 7,
       SCR_ARENA = 8, SCR_LEVELS = 9, SCR_IDX = 10, SCR_PATHS = 11, SCR_POLY_A = 12, SCR_POLY_B = 13, SCR_LDE_TAB = 14, SCR_BINV = 15,
       SCR_DIST_A = 16, SCR_DIST_B = 17, SCR_DIST_C = 18, SCR_DIST_SUB = 19 };
struct sh_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    void* scratch[NUM_SCRATCH] = {};
    size_t scratch_bytes[NUM_SCRATCH] = {};
    uint32_t* d_flag = nullptr;
    digest64* d_root = nullptr;
    uint8_t* h_pinned = nullptr;                 // 256 B of pinned host memory (staging for small D2H copies)
    uint8_t* h_mail = nullptr;                   // pinned, device-mapped mailbox: root[64] | seq (u32)
    uint64_t* d_mail = nullptr;                  // device view of h_mail
    uint32_t mail_seq = 0;                       // last sequence number handed to a kernel
    bool mail_pending = false;                   // the root of the last reduction arrives through the mailbox
    uint64_t mail_fallbacks = 0;
    std::map<uint32_t, fp*> tw_cache;            // key = log_n*2 + inverse
    // the fused LDE's three power tables (SCR_LDE_TAB) as last built: a prover extends polynomial after polynomial onto one coset, and the
    // three k_pow_table launches are a fifth of a 128 x 2^12 -> 2^14 batch's time
    struct LdeTabKey { uint64_t off[2]; size_t rows, chunks, cols; uint32_t log_row_stride, log_cols; void* ptr; } lde_tab_key = {};
    std::map<uint32_t, fp*> ptw_cache;           // direct inter-pass twiddle tables: key = (log_n*2 + inverse)*NTT_MAX_PASSES + pass
    uint8_t* h_batch = nullptr;                  // pinned, device-mapped staging for batched commits: roots (B*64) | c0 (B*16)
    uint8_t* d_hbatch = nullptr;                 // device view of h_batch
    size_t h_batch_bytes = 0;
    uint32_t batch_done_total = 0;               // running total of workgroups that have reported through d_flag[16]
    uint8_t* h_chain = nullptr;                  // pinned, device-mapped: root / fold-constant records of a commit's rounds (api_fri.hip)
    uint8_t* d_chain = nullptr;                  // device view of h_chain
    size_t chain_bytes = 0;
    // Where the fold-constant (c0) records live, at the chain layout's offsets: in device memory that the host writes through the
    // PCIe BAR (c0_buf: uncached allocation; the GPU's polls and its read of the constants stay on the card), or — no large BAR, or
    // STARK_HIP_C0_PINNED=1 — in the pinned chain buffer itself (c0_host == h_chain, c0_dev == d_chain).
    uint8_t* c0_buf = nullptr;
    size_t c0_bytes = 0;
    uint8_t *c0_host = nullptr, *c0_dev = nullptr;
    SpinPool* pool = nullptr;                    // transcript workers of batched commits (asleep between calls)
    uint8_t* h_stage = nullptr;                  // pinned staging ring of the host-buffer entry points (upload_pipelined, api_fri.hip)
    hipEvent_t stage_ev[8] = {};                 // slot i may be refilled once its last copy has left it
    hipStream_t copy_stream = nullptr;           // the staged copies run on a stream of their own; only the HOST waits across the two streams
    ChainEnqueuer* enqueuer = nullptr;           // makes the HIP calls of a commit's rounds while the caller serves the chain
    sh_ctx* sub = nullptr;                       // second stream + buffers for the second lock-step group of a batched commit
    uint8_t* h_query = nullptr;                  // pinned staging of the query phase (paths | values | indices | tree table)
    size_t h_query_bytes = 0;
    uint64_t chain_retries = 0;                  // rounds enqueued a second time because a wait kernel ran out of spin
    uint32_t chain_gen = 0;                      // generation number of the last commit (the records' sequence value)
};

static inline bool is_pow2(size_t x) { return x && !(x & (x - 1)); }
static inline uint32_t ilog2(size_t x) { uint32_t l = 0; while (((size_t)1 << l) < x) ++l; return l; }
static inline fp fp_load_host(const uint64_t v[2]) { return fp_from_u64x2(v[0], v[1]); }
static inline void fp_store_host(uint64_t v[2], fp a) { v[0] = fp_lo64(a); v[1] = fp_hi64(a); }
static inline bool fp_is_canonical(fp a) { return fp_to_u128(a) < fp_prime_u128(); }

// G1: g = 85408008396924667383611388730472331217 = 0x4040fbed12ee470f_b5038f9c18f6f7d1, order 2^119
static inline fp host_generator() { return fp_from_u64x2(0xb5038f9c18f6f7d1ULL, 0x4040fbed12ee470fULL); }
static inline fp host_root_of_unity(uint32_t log_n) {
    fp w = fp_to_mont(host_generator());
    for (uint32_t i = log_n; i < 119; ++i) w = mont_mul(w, w);
    return fp_from_mont(w);
}

#define QUAD_MAX_LOG_COUNT 17                  // from this many digests down a tree is reduced four-lanes-per-compression

// api.hip
int32_t ensure_scratch(sh_ctx* c, int slot, size_t bytes, void** out);
int32_t get_twiddles(sh_ctx* c, uint32_t log_n, bool inverse, const fp** out);
int32_t get_pass_tables(sh_ctx* c, const NttPlan& plan, bool inverse, const fp* tw, NttPassTables* out);
int32_t dev_ntt(sh_ctx* c, fp* d_data, uint32_t log_n, bool inverse, size_t batch = 1);   // natural order in and out, on the context's stream
// api_merkle.hip
int32_t reduce_digests(sh_ctx* c, const digest64* cur, size_t count);
uint32_t leaf_subtree_k(uint32_t log_L);
int32_t merkle_root_from_elems(sh_ctx* c, const fp* d_elems, size_t L);
int32_t wait_mailbox(sh_ctx* c, bool* got);
int32_t fetch_root(sh_ctx* c, uint8_t root[64]);
// api_noalloc.hip
int32_t noalloc_init_device(int device);
// api_fri.hip
// ---- the commit chain (api_fri.hip; shared with the sharded commit, api_dist.hip)
// One pinned, device-mapped buffer per context carries the serial chain of a commit between GPU and host:
//   header (64 B: word 0 = abort status written by a wait kernel that timed out)
//   R root records  : B roots (64 B each) | 64-byte line whose first word is the sequence number
//   R c0 records    : B fold constants (16 B each, Montgomery form), padded to 64 | sequence line
//   staging for the B last codewords
// A sequence word equal to the commit's generation number means "published"; bit 31 on a c0 record means "abort".
struct ChainLayout {
    size_t B, R, rec_root, rec_c0, off_root, off_c0, off_last, total;
    ChainLayout(size_t B_, size_t R_, size_t last_bytes) : B(B_), R(R_) {
        rec_root = B * 64 + 64;
        rec_c0 = ((B * 16 + 63) & ~(size_t)63) + 64;
        off_root = 64;
        off_c0 = off_root + R * rec_root;
        off_last = off_c0 + R * rec_c0;
        total = off_last + last_bytes;
    }
    size_t roots(size_t r) const { return off_root + r * rec_root; }
    size_t root_seq(size_t r) const { return roots(r) + B * 64; }
    size_t c0(size_t r) const { return off_c0 + r * rec_c0; }
    size_t c0_seq(size_t r) const { return c0(r) + rec_c0 - 64; }
};
// Releases the GPU on every exit path once kernels that wait for the host are in flight: publishes "abort" in every c0
// record not yet published and drains the stream.
// Stores into the c0 records may go through a write-combining mapping of the BAR: push them out, in order, now.
static inline void c0_flush() { __builtin_ia32_sfence(); }
struct ChainGuard {
    sh_ctx* c; const ChainLayout* lay; uint32_t gen; size_t next_c0 = 0, waits = 0; bool armed = false;   // waits: c0 records 0..waits-1 may have a kernel waiting
    ChainEnqueuer* helper = nullptr;               // a posted job refers to the caller's frame: it must have returned before the frame goes
    ~ChainGuard() {
        if (!armed) return;
        for (size_t r = next_c0; r < waits; ++r)
            __atomic_store_n(reinterpret_cast<uint32_t*>(c->c0_host + lay->c0_seq(r)), gen | 0x80000000u, __ATOMIC_RELEASE);
        c0_flush();
        if (helper) (void)helper->wait(nullptr);
        (void)hipStreamSynchronize(c->stream);
    }
};
// bytes: the whole layout (pinned buffer); c0_end: the layout's offset behind the last c0 record (what a device-side copy must hold)
int32_t ensure_chain(sh_ctx* c, size_t bytes, size_t c0_end);
// Wait for a sequence word of the chain to carry `gen` without any HIP call: SH_OK, WAIT_RETRY (a wait kernel gave up), WAIT_ENQUEUE_FAILED
// (the enqueue thread's job failed) or an error (timeout).
static const int32_t WAIT_RETRY = 1, WAIT_ENQUEUE_FAILED = 2;
int32_t wait_seq(sh_ctx* c, const uint32_t* flag, uint32_t gen, uint32_t round, double deadline_s, const ChainEnqueuer* helper);
int32_t check_params(const sh_fri_params* p, fp* offset);
uint32_t forest_step(uint32_t lg, size_t B, bool* quad);   // levels the next launch of a forest reduction takes off B trees of 2^lg digests; which kernel
// What a commit leaves resident on the device for the query phase (one codeword): the codeword of every round and, per
// round's tree, the digest arrays its kernels wrote on the way up — (level, array) with 2^level leaves under each digest,
// ascending.  The levels in between are recomputed per opened path (k_open_paths).
struct CommitKeep {
    std::vector<const fp*> cws;
    struct Tree { uint32_t lg = 0; std::vector<std::pair<uint32_t, const digest64*>> stored; };
    std::vector<Tree> trees;
};
int32_t fri_commit_dev(sh_ctx* c, const fp* d_cw, size_t N, const sh_fri_params* prm, sh_transcript_cb cb, void* user,
                       uint8_t* roots_out, uint64_t* last_out, size_t* last_len, CommitKeep* keep = nullptr, bool r0_leaves_hashed = false);
// Host buffer -> device through the context's pinned staging ring: worker threads copy chunk i+1 into a pinned slot while the copy
// engine moves chunk i; `after_chunk(first_elem, n_elems)` (may be empty) enqueues work on a chunk as soon as its copy is enqueued.
int32_t upload_pipelined(sh_ctx* c, const uint64_t* host, size_t n_elems, fp* d_dst, const std::function<int32_t(size_t, size_t)>& after_chunk);
struct PsCbState { sh_proofstream* ps; };
int32_t ps_callback(void* user, uint32_t round, const uint8_t root[64], int32_t want_alpha, uint64_t alpha_out[2]);
