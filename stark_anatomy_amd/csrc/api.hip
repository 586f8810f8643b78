// api.hip — the C ABI of include/stark_hip.h: context, device buffers, and the host drivers that
// sequence the gfx950 kernels (field batch ops, NTT/LDE, Merkle commit, FRI fold and FRI commit).
// No entry point here computes a result on the CPU: without a usable device they return SH_EHIP.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <functional>
#include <map>
#include <new>
#include <atomic>
#include <string>
#include <thread>
#include <vector>
#include "../../include/stark_hip.h"
#include "host_internal.h"
#include "kernels.h"

// ------------------------------------------------------------------ errors
static thread_local std::string g_last_error;
int32_t sh_fail(int32_t code, const char* msg) { g_last_error = msg ? msg : ""; return code; }
int32_t sh_failf(int32_t code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_last_error = buf;
    return code;
}
#define HIPCHK(expr)                                                                                           \
    do {                                                                                                       \
        hipError_t _e = (expr);                                                                                \
        if (_e != hipSuccess) return sh_failf(SH_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)
#define SHCHK(expr) do { int32_t _rc = (expr); if (_rc != SH_OK) return _rc; } while (0)

// ------------------------------------------------------------------ context
#define NUM_SCRATCH 12
enum { SCR_IO_A = 0, SCR_IO_B = 1, SCR_IO_C = 2, SCR_WORK = 3, SCR_DIG_A = 4, SCR_DIG_B = 5, SCR_ROOTS = 6, SCR_C0 = 7,
       SCR_ARENA = 8, SCR_LEVELS = 9, SCR_IDX = 10, SCR_PATHS = 11 };
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
    uint8_t* h_batch = nullptr;                  // pinned, device-mapped staging for batched commits: roots (B*64) | c0 (B*16)
    uint8_t* d_hbatch = nullptr;                 // device view of h_batch
    size_t h_batch_bytes = 0;
    uint32_t batch_done_total = 0;               // running total of workgroups that have reported through d_flag[16]
};

static int32_t ensure_scratch(sh_ctx* c, int slot, size_t bytes, void** out) {
    if (c->scratch_bytes[slot] < bytes) {
        if (c->scratch[slot]) {
            HIPCHK(hipStreamSynchronize(c->stream));
            HIPCHK(hipFree(c->scratch[slot]));
            c->scratch[slot] = nullptr; c->scratch_bytes[slot] = 0;
        }
        size_t want = (bytes + 255) & ~(size_t)255;
        HIPCHK(hipMalloc(&c->scratch[slot], want));
        c->scratch_bytes[slot] = want;
    }
    *out = c->scratch[slot];
    return SH_OK;
}

static inline bool is_pow2(size_t x) { return x && !(x & (x - 1)); }
static inline uint32_t ilog2(size_t x) { uint32_t l = 0; while (((size_t)1 << l) < x) ++l; return l; }
static inline fp fp_load_host(const uint64_t v[2]) { return fp_from_u64x2(v[0], v[1]); }
static inline void fp_store_host(uint64_t v[2], fp a) { v[0] = fp_lo64(a); v[1] = fp_hi64(a); }
static inline bool fp_is_canonical(fp a) { return fp_to_u128(a) < fp_prime_u128(); }

// G1: g = 85408008396924667383611388730472331217 = 0x4040fbed12ee470f_b5038f9c18f6f7d1, order 2^119
static inline fp host_generator() { return fp_from_u64x2(0xb5038f9c18f6f7d1ULL, 0x4040fbed12ee470fULL); }
static fp host_root_of_unity(uint32_t log_n) {
    fp w = fp_to_mont(host_generator());
    for (uint32_t i = log_n; i < 119; ++i) w = mont_mul(w, w);
    return fp_from_mont(w);
}

// [lo | hi] Montgomery-form power table of the (inverse) n-th root of unity, cached per context.
static int32_t get_twiddles(sh_ctx* c, uint32_t log_n, bool inverse, const fp** out) {
    const uint32_t key = log_n * 2 + (inverse ? 1 : 0);
    auto it = c->tw_cache.find(key);
    if (it != c->tw_cache.end()) { *out = it->second; return SH_OK; }
    const size_t n = (size_t)1 << log_n;
    const size_t lo_n = n < ((size_t)1 << NTT_TW_LO_BITS) ? n : ((size_t)1 << NTT_TW_LO_BITS);
    const size_t hi_n = (n >> NTT_TW_LO_BITS) ? (n >> NTT_TW_LO_BITS) : 1;
    std::vector<fp> t(lo_n + hi_n);
    fp w = host_root_of_unity(log_n);
    if (inverse) w = fp_inv(w);
    const fp wm = fp_to_mont(w);
    fp acc = fp_mont_one();
    for (size_t i = 0; i < lo_n; ++i) { t[i] = acc; acc = mont_mul(acc, wm); }
    fp big = wm;
    for (uint32_t i = 0; i < NTT_TW_LO_BITS; ++i) big = mont_mul(big, big);   // w^4096
    acc = fp_mont_one();
    for (size_t i = 0; i < hi_n; ++i) { t[lo_n + i] = acc; acc = mont_mul(acc, big); }
    fp* d = nullptr;
    HIPCHK(hipMalloc(&d, t.size() * sizeof(fp)));
    hipError_t e = hipMemcpyAsync(d, t.data(), t.size() * sizeof(fp), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { (void)hipFree(d); return sh_failf(SH_EHIP, "twiddle upload failed: %s", hipGetErrorString(e)); }
    c->tw_cache[key] = d;
    *out = d;
    return SH_OK;
}

extern "C" {

const char* sh_last_error(void) { return g_last_error.c_str(); }
const char* sh_version(void) { return "stark_hip 0.1 (gfx950)"; }

int32_t sh_ctx_create(int32_t device, void* hip_stream, sh_ctx** out) {
    if (!out) return sh_fail(SH_EINVAL, "sh_ctx_create: null out");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return sh_failf(SH_EHIP, "sh_ctx_create: no HIP device available (%s); libstark_hip has no CPU fallback",
                        e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device < 0 || device >= count) return sh_failf(SH_EINVAL, "sh_ctx_create: device %d out of range [0,%d)", device, count);
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return sh_failf(SH_EHIP, "sh_ctx_create: device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    sh_ctx* c = new (std::nothrow) sh_ctx();
    if (!c) return sh_fail(SH_ENOMEM, "sh_ctx_create: out of memory");
    c->device = device;
    if (hip_stream) { c->stream = (hipStream_t)hip_stream; c->own_stream = false; }
    else {
        e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete c; return sh_failf(SH_EHIP, "hipStreamCreate failed: %s", hipGetErrorString(e)); }
        c->own_stream = true;
    }
    e = hipMalloc(&c->d_flag, 256);
    if (e == hipSuccess) e = hipMemset(c->d_flag, 0, 256);
    if (e == hipSuccess) e = hipMalloc(&c->d_root, 256);
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_pinned, 256, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_mail, 256, hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) { memset(c->h_mail, 0, 256); e = hipHostGetDevicePointer((void**)&c->d_mail, c->h_mail, 0); }
    if (e != hipSuccess) { sh_ctx_destroy(c); return sh_failf(SH_EHIP, "context allocation failed: %s", hipGetErrorString(e)); }
    *out = c;
    return SH_OK;
}
int32_t sh_ctx_destroy(sh_ctx* c) {
    if (!c) return SH_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (getenv("STARK_HIP_DEBUG")) fprintf(stderr, "[stark_hip] ctx %p: mailbox roots %u, fallbacks %llu\n", (void*)c, c->mail_seq, (unsigned long long)c->mail_fallbacks);
    for (auto& kv : c->tw_cache) (void)hipFree(kv.second);
    for (int i = 0; i < NUM_SCRATCH; ++i) if (c->scratch[i]) (void)hipFree(c->scratch[i]);
    if (c->d_flag) (void)hipFree(c->d_flag);
    if (c->d_root) (void)hipFree(c->d_root);
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    if (c->h_mail) (void)hipHostFree(c->h_mail);
    if (c->h_batch) (void)hipHostFree(c->h_batch);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return SH_OK;
}
#define CTXCHK(c)                                                          \
    do {                                                                   \
        if (!(c)) return sh_fail(SH_EINVAL, "null sh_ctx");                \
        HIPCHK(hipSetDevice((c)->device));                                 \
    } while (0)

int32_t sh_ctx_sync(sh_ctx* c) { CTXCHK(c); HIPCHK(hipStreamSynchronize(c->stream)); return SH_OK; }
int32_t sh_ctx_device(sh_ctx* c, int32_t* device) { if (!c || !device) return sh_fail(SH_EINVAL, "null pointer"); *device = c->device; return SH_OK; }
int32_t sh_dev_alloc(sh_ctx* c, size_t bytes, void** dptr) {
    CTXCHK(c);
    if (!dptr) return sh_fail(SH_EINVAL, "sh_dev_alloc: null out");
    HIPCHK(hipMalloc(dptr, bytes ? bytes : 16));
    return SH_OK;
}
int32_t sh_dev_free(sh_ctx* c, void* dptr) { CTXCHK(c); if (dptr) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(dptr)); } return SH_OK; }
int32_t sh_dev_upload(sh_ctx* c, void* dptr, const void* host, size_t bytes) {
    CTXCHK(c);
    if (!bytes) return SH_OK;
    if (!dptr || !host) return sh_fail(SH_EINVAL, "sh_dev_upload: null pointer");
    HIPCHK(hipMemcpyAsync(dptr, host, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SH_OK;
}
int32_t sh_dev_download(sh_ctx* c, void* host, const void* dptr, size_t bytes) {
    CTXCHK(c);
    if (!bytes) return SH_OK;
    if (!dptr || !host) return sh_fail(SH_EINVAL, "sh_dev_download: null pointer");
    HIPCHK(hipMemcpyAsync(host, dptr, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SH_OK;
}

}  // extern "C"

// ------------------------------------------------------------------ field batch ops
static int32_t dev_fp_op(sh_ctx* c, int op, const uint64_t* d_a, const uint64_t* d_b, uint64_t* d_out, size_t n) {
    CTXCHK(c);
    if (n == 0) return SH_OK;
    const bool needs_b = (op == FPOP_ADD || op == FPOP_SUB || op == FPOP_MUL || op == FPOP_DIV);
    if (!d_a || !d_out || (needs_b && !d_b)) return sh_fail(SH_EINVAL, "field op: null pointer");
    const bool checks = (op == FPOP_DIV || op == FPOP_INV);
    if (checks) HIPCHK(hipMemsetAsync(c->d_flag, 0, 4, c->stream));
    HIPCHK(launch_fp_binop(op, (const fp*)d_a, (const fp*)d_b, (fp*)d_out, n, c->d_flag, c->stream));
    if (checks) {
        HIPCHK(hipMemcpyAsync(c->h_pinned, c->d_flag, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (*(volatile uint32_t*)c->h_pinned) return sh_fail(SH_EDIVZERO, "Cannot divide by zero");   // field.rs:102
    }
    return SH_OK;
}
static int32_t host_fp_op(sh_ctx* c, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
    CTXCHK(c);
    if (n == 0) return SH_OK;
    const bool needs_b = (op == FPOP_ADD || op == FPOP_SUB || op == FPOP_MUL || op == FPOP_DIV);
    if (!a || !out || (needs_b && !b)) return sh_fail(SH_EINVAL, "field op: null pointer");
    void *da, *db = nullptr, *dout;
    SHCHK(ensure_scratch(c, SCR_IO_A, n * 16, &da));
    SHCHK(ensure_scratch(c, SCR_IO_C, n * 16, &dout));
    HIPCHK(hipMemcpyAsync(da, a, n * 16, hipMemcpyHostToDevice, c->stream));
    if (needs_b) {
        SHCHK(ensure_scratch(c, SCR_IO_B, n * 16, &db));
        HIPCHK(hipMemcpyAsync(db, b, n * 16, hipMemcpyHostToDevice, c->stream));
    }
    SHCHK(dev_fp_op(c, op, (const uint64_t*)da, (const uint64_t*)db, (uint64_t*)dout, n));
    HIPCHK(hipMemcpyAsync(out, dout, n * 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SH_OK;
}

// ------------------------------------------------------------------ NTT / LDE
static int32_t dev_ntt(sh_ctx* c, fp* d_data, uint32_t log_n, bool inverse, size_t batch = 1) {
    if (log_n > SH_MAX_LOG_N) return sh_failf(SH_EINVAL, "ntt: log_n %u > %d", log_n, SH_MAX_LOG_N);
    if (batch == 0) return SH_OK;
    if (batch > ((size_t)1 << (SH_MAX_LOG_N + 2 - log_n))) return sh_failf(SH_EINVAL, "ntt: batch %zu x 2^%u exceeds 2^%d elements", batch, log_n, SH_MAX_LOG_N + 2);
    const size_t n = ((size_t)1 << log_n) * batch;
    const fp* tw;
    SHCHK(get_twiddles(c, log_n, inverse, &tw));
    void* scratch;
    SHCHK(ensure_scratch(c, SCR_WORK, n * sizeof(fp), &scratch));
    NttPlan plan;
    ntt_make_plan(log_n, &plan);
    fp scale = fp_mont_one();
    if (inverse) scale = fp_to_mont(fp_inv(fp_from_u64x2((uint64_t)1 << log_n, 0)));
    HIPCHK(launch_ntt(plan, d_data, (fp*)scratch, tw, scale, inverse, batch, c->stream));
    return SH_OK;
}

// ------------------------------------------------------------------ Merkle
// Reduce `count` (power of two) digests at `cur` to the root in c->d_root (and the host mailbox).  `cur` is not
// modified.  Wide levels: one 2^K-node subtree per lane (throughput).  From 2^17 nodes down: four lanes per
// compression, levels in LDS (latency), in at most two launches.
#define QUAD_MAX_LOG_COUNT 17
static int32_t reduce_digests(sh_ctx* c, const digest64* cur, size_t count) {
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
static uint32_t leaf_subtree_k(uint32_t log_L) {
    // enough lanes to fill the chip first (>= 2^18 lanes), then deeper per-lane subtrees
    uint32_t K = log_L > 18 ? log_L - 18 : 0;
    return K > 3 ? 3 : K;
}
// root of merkle(leafhash(elems[0..L))) -> c->d_root (enqueued, not synchronised)
static int32_t merkle_root_from_elems(sh_ctx* c, const fp* d_elems, size_t L) {
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
static int32_t wait_mailbox(sh_ctx* c, bool* got) {
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
static int32_t fetch_root(sh_ctx* c, uint8_t root[64]) {
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

// ------------------------------------------------------------------ FRI
static int32_t check_params(const sh_fri_params* p, fp* offset) {
    if (!p) return sh_fail(SH_EINVAL, "fri: null params");
    if (p->expansion == 0) return sh_fail(SH_EINVAL, "fri: expansion must be >= 1");
    *offset = fp_load_host(p->offset);
    if (!fp_is_canonical(*offset) || fp_is_zero(*offset)) return sh_fail(SH_EINVAL, "fri: offset must be a non-zero canonical field element");
    return SH_OK;
}
// `keep`: when set, every round's codeword stays resident (round r >= 1 at offset N - N/2^(r-1) of one N-element arena,
// instead of the two ping-pong buffers)  and its device pointer is recorded — the query phase opens them afterwards.
static int32_t fri_commit_dev(sh_ctx* c, const fp* d_cw, size_t N, const sh_fri_params* prm, sh_transcript_cb cb, void* user,
                              uint8_t* roots_out, uint64_t* last_out, size_t* last_len, std::vector<const fp*>* keep = nullptr) {
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

struct PsCbState { sh_proofstream* ps; };
static int32_t ps_callback(void* user, uint32_t, const uint8_t root[64], int32_t want_alpha, uint64_t alpha_out[2]) {
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

int32_t sh_fp_new(sh_ctx* c, const uint64_t* x, uint64_t* out, size_t n) { return host_fp_op(c, FPOP_NEW, x, nullptr, out, n); }
int32_t sh_fp_add(sh_ctx* c, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) { return host_fp_op(c, FPOP_ADD, a, b, out, n); }
int32_t sh_fp_sub(sh_ctx* c, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) { return host_fp_op(c, FPOP_SUB, a, b, out, n); }
int32_t sh_fp_mul(sh_ctx* c, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) { return host_fp_op(c, FPOP_MUL, a, b, out, n); }
int32_t sh_fp_div(sh_ctx* c, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) { return host_fp_op(c, FPOP_DIV, a, b, out, n); }
int32_t sh_fp_inv(sh_ctx* c, const uint64_t* a, uint64_t* out, size_t n) { return host_fp_op(c, FPOP_INV, a, nullptr, out, n); }
int32_t sh_dev_fp_add(sh_ctx* c, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) { return dev_fp_op(c, FPOP_ADD, a, b, out, n); }
int32_t sh_dev_fp_sub(sh_ctx* c, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) { return dev_fp_op(c, FPOP_SUB, a, b, out, n); }
int32_t sh_dev_fp_mul(sh_ctx* c, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) { return dev_fp_op(c, FPOP_MUL, a, b, out, n); }
int32_t sh_dev_fp_div(sh_ctx* c, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) { return dev_fp_op(c, FPOP_DIV, a, b, out, n); }
int32_t sh_dev_fp_inv(sh_ctx* c, const uint64_t* a, uint64_t* out, size_t n) { return dev_fp_op(c, FPOP_INV, a, nullptr, out, n); }

int32_t sh_dev_ntt(sh_ctx* c, uint64_t* d_data, uint32_t log_n, int32_t inverse) {
    CTXCHK(c);
    if (!d_data) return sh_fail(SH_EINVAL, "sh_dev_ntt: null pointer");
    return dev_ntt(c, (fp*)d_data, log_n, inverse != 0);
}
int32_t sh_dev_ntt_batch(sh_ctx* c, uint64_t* d_data, size_t batch, uint32_t log_n, int32_t inverse) {
    CTXCHK(c);
    if (!d_data && batch) return sh_fail(SH_EINVAL, "sh_dev_ntt_batch: null pointer");
    return dev_ntt(c, (fp*)d_data, log_n, inverse != 0, batch);
}
int32_t sh_ntt_batch(sh_ctx* c, uint64_t* data, size_t batch, uint32_t log_n, int32_t inverse) {
    CTXCHK(c);
    if (!batch) return SH_OK;
    if (!data) return sh_fail(SH_EINVAL, "sh_ntt_batch: null pointer");
    if (log_n > SH_MAX_LOG_N) return sh_failf(SH_EINVAL, "sh_ntt_batch: log_n %u > %d", log_n, SH_MAX_LOG_N);
    if (batch > ((size_t)1 << (SH_MAX_LOG_N + 2 - log_n))) return sh_failf(SH_EINVAL, "sh_ntt_batch: batch %zu x 2^%u exceeds 2^%d elements", batch, log_n, SH_MAX_LOG_N + 2);
    const size_t bytes = (batch << log_n) * 16;
    void* d;
    SHCHK(ensure_scratch(c, SCR_IO_A, bytes, &d));
    HIPCHK(hipMemcpyAsync(d, data, bytes, hipMemcpyHostToDevice, c->stream));
    SHCHK(dev_ntt(c, (fp*)d, log_n, inverse != 0, batch));
    HIPCHK(hipMemcpyAsync(data, d, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SH_OK;
}
int32_t sh_dev_twiddle_scale(sh_ctx* c, uint64_t* d_data, size_t rows, uint32_t log_cols, uint64_t row0, uint64_t col0, uint32_t log_n, int32_t inverse) {
    CTXCHK(c);
    if (!rows) return SH_OK;
    if (!d_data) return sh_fail(SH_EINVAL, "sh_dev_twiddle_scale: null pointer");
    if (log_n > SH_MAX_LOG_N || log_cols > log_n) return sh_failf(SH_EINVAL, "sh_dev_twiddle_scale: log_cols %u, log_n %u out of range", log_cols, log_n);
    if ((row0 + rows) > ((uint64_t)1 << 32) || (col0 >> 32)) return sh_fail(SH_EINVAL, "sh_dev_twiddle_scale: row/column index beyond 2^32");
    const fp* tw;
    SHCHK(get_twiddles(c, log_n, inverse != 0, &tw));
    HIPCHK(launch_twiddle_scale((fp*)d_data, rows, log_cols, row0, col0, tw, log_n, c->stream));
    return SH_OK;
}
int32_t sh_twiddle_scale(sh_ctx* c, uint64_t* data, size_t rows, uint32_t log_cols, uint64_t row0, uint64_t col0, uint32_t log_n, int32_t inverse) {
    CTXCHK(c);
    if (!rows) return SH_OK;
    if (!data) return sh_fail(SH_EINVAL, "sh_twiddle_scale: null pointer");
    if (log_cols > SH_MAX_LOG_N || rows > ((size_t)1 << (SH_MAX_LOG_N + 2 - log_cols))) return sh_fail(SH_EINVAL, "sh_twiddle_scale: block too large");
    const size_t bytes = (rows << log_cols) * 16;
    void* d;
    SHCHK(ensure_scratch(c, SCR_IO_A, bytes, &d));
    HIPCHK(hipMemcpyAsync(d, data, bytes, hipMemcpyHostToDevice, c->stream));
    SHCHK(sh_dev_twiddle_scale(c, (uint64_t*)d, rows, log_cols, row0, col0, log_n, inverse));
    HIPCHK(hipMemcpyAsync(data, d, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SH_OK;
}
int32_t sh_ntt(sh_ctx* c, uint64_t* data, uint32_t log_n, int32_t inverse) {
    CTXCHK(c);
    if (!data) return sh_fail(SH_EINVAL, "sh_ntt: null pointer");
    if (log_n > SH_MAX_LOG_N) return sh_failf(SH_EINVAL, "sh_ntt: log_n %u > %d", log_n, SH_MAX_LOG_N);
    const size_t bytes = ((size_t)1 << log_n) * 16;
    void* d;
    SHCHK(ensure_scratch(c, SCR_IO_A, bytes, &d));
    HIPCHK(hipMemcpyAsync(d, data, bytes, hipMemcpyHostToDevice, c->stream));
    SHCHK(dev_ntt(c, (fp*)d, log_n, inverse != 0));
    HIPCHK(hipMemcpyAsync(data, d, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SH_OK;
}
int32_t sh_dev_coset_lde(sh_ctx* c, const uint64_t* d_coeffs, uint32_t log_n, uint32_t log_expansion, const uint64_t offset[2], uint64_t* d_out) {
    CTXCHK(c);
    if (!d_coeffs || !offset || !d_out) return sh_fail(SH_EINVAL, "sh_dev_coset_lde: null pointer");
    if (log_n + log_expansion > SH_MAX_LOG_N) return sh_failf(SH_EINVAL, "sh_dev_coset_lde: log_n + log_expansion > %d", SH_MAX_LOG_N);
    const fp off = fp_load_host(offset);
    if (!fp_is_canonical(off)) return sh_fail(SH_EINVAL, "sh_dev_coset_lde: non-canonical offset");
    const size_t n = (size_t)1 << log_n, N = n << log_expansion;
    HIPCHK(launch_coset_scale_pad((const fp*)d_coeffs, n, off, (fp*)d_out, N, c->stream));
    return dev_ntt(c, (fp*)d_out, log_n + log_expansion, false);
}
int32_t sh_coset_lde(sh_ctx* c, const uint64_t* coeffs, uint32_t log_n, uint32_t log_expansion, const uint64_t offset[2], uint64_t* out) {
    CTXCHK(c);
    if (!coeffs || !offset || !out) return sh_fail(SH_EINVAL, "sh_coset_lde: null pointer");
    if (log_n + log_expansion > SH_MAX_LOG_N) return sh_failf(SH_EINVAL, "sh_coset_lde: log_n + log_expansion > %d", SH_MAX_LOG_N);
    const size_t n = (size_t)1 << log_n, N = n << log_expansion;
    void *din, *dout;
    SHCHK(ensure_scratch(c, SCR_IO_A, n * 16, &din));
    SHCHK(ensure_scratch(c, SCR_IO_B, N * 16, &dout));
    HIPCHK(hipMemcpyAsync(din, coeffs, n * 16, hipMemcpyHostToDevice, c->stream));
    SHCHK(sh_dev_coset_lde(c, (const uint64_t*)din, log_n, log_expansion, offset, (uint64_t*)dout));
    HIPCHK(hipMemcpyAsync(out, dout, N * 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SH_OK;
}

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
// MerkleTree::open, merkle.rs:70-108.  All levels are built once on the GPU (level by level, K = 1) and the
// log2(L) siblings are gathered from them, instead of the reference's per-call subtree re-commits.
int32_t sh_merkle_open(sh_ctx* c, const uint8_t* leaves, size_t L, size_t index, uint8_t* path, size_t* path_len) {
    CTXCHK(c);
    if (!leaves || !path || !path_len) return sh_fail(SH_EINVAL, "sh_merkle_open: null pointer");
    if (!is_pow2(L)) return sh_fail(SH_EINVAL, "sh_merkle_open: leaf count must be a power of two (merkle.rs:72)");
    if (index >= L) return sh_fail(SH_EBOUNDS, "sh_merkle_open: index out of bounds (merkle.rs:75)");
    if (L < 2) return sh_fail(SH_EBOUNDS, "sh_merkle_open: a one-leaf tree has no path (merkle.rs:104)");
    const uint32_t depth = ilog2(L);
    void* d;                                   // levels 0..depth-1 back to back: L + L/2 + ... + 2 digests
    SHCHK(ensure_scratch(c, SCR_IO_A, 2 * L * 64, &d));
    digest64* lvl = (digest64*)d;
    HIPCHK(hipMemcpyAsync(lvl, leaves, L * 64, hipMemcpyHostToDevice, c->stream));
    size_t n = L, idx = index;
    for (uint32_t l = 0; l < depth; ++l) {
        HIPCHK(hipMemcpyAsync(path + 64 * (size_t)l, lvl + (idx ^ 1), 64, hipMemcpyDeviceToHost, c->stream));
        if (l + 1 < depth) HIPCHK(launch_node_subtree(lvl, lvl + n, n >> 1, 1, c->stream));
        lvl += n; n >>= 1; idx >>= 1;
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    *path_len = depth;
    return SH_OK;
}

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

// ------------------------------------------------------------------ query phase: batched openings, prove, verify
// (definitions G9-G12 in DESIGN.md §9 — the reference holds none of this.)
// All levels of the tree whose L leaf digests already sit at lvl[0..L): level l at offset 2L - (2L >> l).
static int32_t build_levels(sh_ctx* c, digest64* lvl, size_t L) {
    size_t n = L;
    while (n > 2) {
        HIPCHK(launch_node_subtree(lvl, lvl + n, n >> 1, 1, c->stream));
        lvl += n; n >>= 1;
    }
    return SH_OK;
}
// paths (n_idx x depth x 64 B, host) of `h_idx` in the tree whose levels are at `levels`
static int32_t gather_paths_to_host(sh_ctx* c, const digest64* levels, size_t L, const uint64_t* h_idx, size_t n_idx, uint8_t* h_paths) {
    if (!n_idx) return SH_OK;
    const uint32_t depth = ilog2(L);
    void *d_idx, *d_paths;
    SHCHK(ensure_scratch(c, SCR_IDX, n_idx * 8, &d_idx));
    SHCHK(ensure_scratch(c, SCR_PATHS, n_idx * depth * 64, &d_paths));
    HIPCHK(hipMemcpyAsync(d_idx, h_idx, n_idx * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(launch_gather_paths(levels, L, (const uint64_t*)d_idx, n_idx, (digest64*)d_paths, c->stream));
    HIPCHK(hipMemcpyAsync(h_paths, d_paths, n_idx * depth * 64, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SH_OK;
}
static int32_t check_open_args(const char* who, size_t L, const uint64_t* idx, size_t n_idx) {
    if (!is_pow2(L)) return sh_failf(SH_EINVAL, "%s: leaf count must be a power of two (merkle.rs:72)", who);
    if (L < 2) return sh_failf(SH_EBOUNDS, "%s: a one-leaf tree has no path (merkle.rs:104)", who);
    if (ilog2(L) > SH_MAX_LOG_N) return sh_failf(SH_EINVAL, "%s: L > 2^%d", who, SH_MAX_LOG_N);
    for (size_t i = 0; i < n_idx; ++i) if (idx[i] >= L) return sh_failf(SH_EBOUNDS, "%s: index out of bounds (merkle.rs:75)", who);
    return SH_OK;
}

// G10: index sampling.  sample_index = big-endian integer of the digest mod size (the G3 convention);
// candidates are Blake2b512(seed || counter as u64 LE), counter = 0, 1, ...; a candidate is kept when its residue
// mod reduced_size is new, until `number` are kept.
static uint64_t sample_index(const uint8_t* bytes, size_t n, uint64_t size) {
    unsigned __int128 acc = 0;
    for (size_t i = 0; i < n; ++i) acc = ((acc << 8) | bytes[i]) % size;
    return (uint64_t)acc;
}
static int32_t sample_indices(const uint8_t seed[32], uint64_t size, uint64_t reduced_size, uint32_t number, std::vector<uint64_t>* out) {
    out->clear();
    if (size == 0 || reduced_size == 0) return sh_fail(SH_EINVAL, "sample_indices: empty domain");
    if (number > reduced_size) return sh_failf(SH_EINVAL, "sample_indices: cannot sample %u indices with distinct residues mod %llu", number, (unsigned long long)reduced_size);
    std::vector<uint64_t> reduced;
    uint8_t buf[40], dig[64];
    memcpy(buf, seed, 32);
    for (uint64_t counter = 0; out->size() < number; ++counter) {
        for (int k = 0; k < 8; ++k) buf[32 + k] = (uint8_t)(counter >> (8 * k));
        host_blake2b512(buf, 40, dig);
        const uint64_t index = sample_index(dig, 64, size), red = index % reduced_size;
        bool seen = false;
        for (uint64_t r : reduced) if (r == red) { seen = true; break; }
        if (!seen) { out->push_back(index); reduced.push_back(red); }
    }
    return SH_OK;
}

static int32_t fri_prove_dev(sh_ctx* c, const fp* d_cw, size_t N, const sh_fri_params* prm, sh_proofstream* ps, uint64_t* top_out, size_t* n_top) {
    CTXCHK(c);
    if (!ps || !n_top) return sh_fail(SH_EINVAL, "fri_prove: null pointer");
    *n_top = 0;
    if (!prm) return sh_fail(SH_EINVAL, "fri: null params");
    const uint32_t t = prm->num_colinearity_tests;
    const uint32_t R = sh_fri_num_rounds(N, prm->expansion, t);
    std::vector<uint8_t> roots((size_t)(R ? R : 1) * 64);
    std::vector<uint64_t> last(2 * (N >> (R ? R - 1 : 0)));          // the last codeword only (N/2^(R-1) elements)
    size_t last_len = 0;
    std::vector<const fp*> cws;
    PsCbState st{ps};
    SHCHK(fri_commit_dev(c, d_cw, N, prm, ps_callback, &st, roots.data(), last.data(), &last_len, &cws));
    SHCHK(sh_proofstream_push_codeword(ps, last.data(), last_len));
    if (R < 2) return SH_OK;                                   // a single codeword: nothing to test against
    uint8_t seed[32];
    SHCHK(sh_proofstream_prover_fiat_shamir(ps, seed));
    std::vector<uint64_t> top;
    SHCHK(sample_indices(seed, N / 2, last_len, t, &top));
    if (top_out) memcpy(top_out, top.data(), top.size() * 8);
    *n_top = top.size();

    // Per tree r: openings wanted = a_r, b_r (tests of round r, r < R-1) and c_(r-1) (tests of round r-1, r >= 1).
    // Each tree's levels are built once (leaf hashes + K = 1 node levels into one buffer) and all its paths and leaf
    // values come back in one gather each.
    std::vector<std::vector<uint64_t>> idx(R), vals(R);
    std::vector<std::vector<uint8_t>> paths(R);
    for (uint32_t r = 0; r < R; ++r) {
        const size_t L = N >> r;
        if (r + 1 < R) {
            for (uint32_t s_ = 0; s_ < t; ++s_) idx[r].push_back(top[s_] % (L / 2));
            for (uint32_t s_ = 0; s_ < t; ++s_) idx[r].push_back(top[s_] % (L / 2) + L / 2);
        }
        if (r >= 1) for (uint32_t s_ = 0; s_ < t; ++s_) idx[r].push_back(top[s_] % L);        // c of round r-1: top % (L_(r-1)/2)
        const size_t n_idx = idx[r].size();
        const uint32_t depth = ilog2(L);
        void *lv, *d_idx, *d_vals;
        SHCHK(ensure_scratch(c, SCR_LEVELS, 2 * L * sizeof(digest64), &lv));
        HIPCHK(launch_hash_leaves(cws[r], (digest64*)lv, L, c->stream));
        SHCHK(build_levels(c, (digest64*)lv, L));
        paths[r].resize(n_idx * depth * 64);
        SHCHK(gather_paths_to_host(c, (const digest64*)lv, L, idx[r].data(), n_idx, paths[r].data()));
        SHCHK(ensure_scratch(c, SCR_IDX, n_idx * 8, &d_idx));              // still holds idx[r] from the call above
        SHCHK(ensure_scratch(c, SCR_PATHS, n_idx * sizeof(fp), &d_vals));
        HIPCHK(launch_gather_elems(cws[r], (const uint64_t*)d_idx, n_idx, (fp*)d_vals, c->stream));
        vals[r].resize(2 * n_idx);
        HIPCHK(hipMemcpyAsync(vals[r].data(), d_vals, n_idx * sizeof(fp), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    for (uint32_t r = 0; r + 1 < R; ++r) {                                  // G11: per round, t value triples, then 3t paths
        const uint32_t d0 = ilog2(N >> r), d1 = d0 - 1;
        const size_t c_off = (r + 2 < R) ? 2 * (size_t)t : 0;               // where tree r+1 keeps its c entries
        for (uint32_t s_ = 0; s_ < t; ++s_) {
            uint64_t abc[6] = {vals[r][2 * s_], vals[r][2 * s_ + 1], vals[r][2 * (t + s_)], vals[r][2 * (t + s_) + 1],
                               vals[r + 1][2 * (c_off + s_)], vals[r + 1][2 * (c_off + s_) + 1]};
            SHCHK(sh_proofstream_push_leaves(ps, abc, 3));
        }
        for (uint32_t s_ = 0; s_ < t; ++s_) {
            SHCHK(sh_proofstream_push_path(ps, paths[r].data() + (size_t)s_ * d0 * 64, d0));
            SHCHK(sh_proofstream_push_path(ps, paths[r].data() + (size_t)(t + s_) * d0 * 64, d0));
            SHCHK(sh_proofstream_push_path(ps, paths[r + 1].data() + (c_off + s_) * d1 * 64, d1));
        }
    }
    return SH_OK;
}

// host radix-2 NTT (decimation in time) for the verifier's degree test of the last codeword (a few hundred points)
static void host_ntt(std::vector<fp>& a, fp w) {
    const size_t n = a.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        fp wl = w;
        for (size_t k = len; k < n; k <<= 1) wl = fp_mul(wl, wl);
        for (size_t s0 = 0; s0 < n; s0 += len) {
            fp x = fp_one();
            for (size_t j = 0; j < len / 2; ++j) {
                const fp u = a[s0 + j], v = fp_mul(a[s0 + j + len / 2], x);
                a[s0 + j] = fp_add(u, v); a[s0 + j + len / 2] = fp_sub(u, v);
                x = fp_mul(x, wl);
            }
        }
    }
}

extern "C" {

int32_t sh_dev_merkle_open_elems_batch(sh_ctx* c, const uint64_t* d_elems, size_t L, const uint64_t* indices, size_t n_idx, uint8_t* paths) {
    CTXCHK(c);
    if (!d_elems || (!indices && n_idx) || (!paths && n_idx)) return sh_fail(SH_EINVAL, "sh_dev_merkle_open_elems_batch: null pointer");
    SHCHK(check_open_args("sh_dev_merkle_open_elems_batch", L, indices, n_idx));
    void* lv;
    SHCHK(ensure_scratch(c, SCR_LEVELS, 2 * L * sizeof(digest64), &lv));
    HIPCHK(launch_hash_leaves((const fp*)d_elems, (digest64*)lv, L, c->stream));
    SHCHK(build_levels(c, (digest64*)lv, L));
    return gather_paths_to_host(c, (const digest64*)lv, L, indices, n_idx, paths);
}
int32_t sh_merkle_open_batch(sh_ctx* c, const uint8_t* leaves, size_t L, const uint64_t* indices, size_t n_idx, uint8_t* paths) {
    CTXCHK(c);
    if (!leaves || (!indices && n_idx) || (!paths && n_idx)) return sh_fail(SH_EINVAL, "sh_merkle_open_batch: null pointer");
    SHCHK(check_open_args("sh_merkle_open_batch", L, indices, n_idx));
    void* lv;
    SHCHK(ensure_scratch(c, SCR_LEVELS, 2 * L * sizeof(digest64), &lv));
    HIPCHK(hipMemcpyAsync(lv, leaves, L * 64, hipMemcpyHostToDevice, c->stream));
    SHCHK(build_levels(c, (digest64*)lv, L));
    return gather_paths_to_host(c, (const digest64*)lv, L, indices, n_idx, paths);
}
int32_t sh_fri_sample_indices(const uint8_t seed[32], uint64_t size, uint64_t reduced_size, uint32_t number, uint64_t* out) {
    if (!seed || (!out && number)) return sh_fail(SH_EINVAL, "sh_fri_sample_indices: null pointer");
    std::vector<uint64_t> v;
    SHCHK(sample_indices(seed, size, reduced_size, number, &v));
    memcpy(out, v.data(), v.size() * 8);
    return SH_OK;
}
int32_t sh_dev_fri_prove(sh_ctx* c, const uint64_t* d_codeword, size_t N, const sh_fri_params* params, sh_proofstream* ps,
                         uint64_t* top_level_indices, size_t* n_indices) {
    return fri_prove_dev(c, (const fp*)d_codeword, N, params, ps, top_level_indices, n_indices);
}
int32_t sh_fri_prove(sh_ctx* c, const uint64_t* codeword, size_t N, const sh_fri_params* params, sh_proofstream* ps,
                     uint64_t* top_level_indices, size_t* n_indices) {
    CTXCHK(c);
    if (!codeword) return sh_fail(SH_EINVAL, "sh_fri_prove: null pointer");
    if (!is_pow2(N) || ilog2(N) > SH_MAX_LOG_N) return sh_failf(SH_EINVAL, "sh_fri_prove: N must be a power of two <= 2^%d", SH_MAX_LOG_N);
    void* d;
    SHCHK(ensure_scratch(c, SCR_IO_C, N * 16, &d));
    HIPCHK(hipMemcpyAsync(d, codeword, N * 16, hipMemcpyHostToDevice, c->stream));
    return fri_prove_dev(c, (const fp*)d, N, params, ps, top_level_indices, n_indices);
}

// G12: the verifier.  Host only ("CPU-side is enough"): ~R·3t path checks of log2 L compressions.
// Reads the stream from its read index.  *accepted = 1/0; on 0, sh_last_error() says which check failed.
int32_t sh_fri_verify(sh_proofstream* ps, size_t N, const sh_fri_params* prm, int32_t* accepted, uint64_t* top_level_indices, size_t* n_indices) {
    if (!ps || !accepted) return sh_fail(SH_EINVAL, "sh_fri_verify: null pointer");
    *accepted = 0;
    if (n_indices) *n_indices = 0;
    if (!is_pow2(N) || ilog2(N) > 119) return sh_fail(SH_EINVAL, "sh_fri_verify: N must be a power of two <= 2^119");
    fp off;
    SHCHK(check_params(prm, &off));
    const uint32_t t = prm->num_colinearity_tests, log_N = ilog2(N);
    const uint32_t R = sh_fri_num_rounds(N, prm->expansion, t);
    auto reject = [&](const char* why) { sh_fail(SH_OK, why); return SH_OK; };
    const uint8_t* js; size_t jl;
    std::vector<uint8_t> roots((size_t)R * 64);
    std::vector<fp> alphas;
    for (uint32_t r = 0; r < R; ++r) {
        if (sh_proofstream_pull(ps, &js, &jl) != SH_OK || !host_parse_merkle_root(js, jl, roots.data() + 64 * (size_t)r)) return reject("verify: expected a MerkleRoot item");
        if (r + 1 < R) {
            uint8_t dig[32]; uint64_t a[2];
            SHCHK(sh_proofstream_verifier_fiat_shamir(ps, dig));
            SHCHK(sh_sample_field(dig, a));
            alphas.push_back(fp_load_host(a));
        }
    }
    std::vector<uint64_t> lastw;
    if (sh_proofstream_pull(ps, &js, &jl) != SH_OK || !host_parse_elems(js, jl, "Codeword", &lastw)) return reject("verify: expected the last Codeword item");
    const size_t L_last = N >> (R ? R - 1 : 0);
    if (lastw.size() != 2 * L_last) return reject("verify: last codeword has the wrong length");
    std::vector<fp> last(L_last);
    for (size_t i = 0; i < L_last; ++i) { last[i] = fp_from_u64x2(lastw[2 * i], lastw[2 * i + 1]); if (!fp_is_canonical(last[i])) return reject("verify: non-canonical element in the last codeword"); }
    if (R == 0) { *accepted = 1; return SH_OK; }
    {   // last root == Merkle root of the last codeword's leaf hashes
        std::vector<uint8_t> lv(L_last * 64);
        for (size_t i = 0; i < L_last; ++i) { uint8_t le[16]; memcpy(le, &lastw[2 * i], 8); memcpy(le + 8, &lastw[2 * i + 1], 8); host_blake2b512(le, 16, lv.data() + 64 * i); }
        for (size_t n = L_last; n > 1; n >>= 1)
            for (size_t i = 0; i < n / 2; ++i) { uint8_t d[64]; host_blake2b512(lv.data() + 128 * i, 128, d); memcpy(lv.data() + 64 * i, d, 64); }
        if (memcmp(lv.data(), roots.data() + 64 * (size_t)(R - 1), 64) != 0) return reject("verify: last codeword does not match the last Merkle root");
    }
    {   // low degree: the last codeword, on the coset off_last·<w_last>, must interpolate to degree < L_last/expansion.
        // INTT gives c_j·off_last^j up to the factor 1/L_last; zero-ness of the high coefficients is unaffected by either.
        const uint32_t log_last = ilog2(L_last);
        std::vector<fp> co = last;
        if (L_last > 1) host_ntt(co, fp_inv(host_root_of_unity(log_last)));
        const size_t max_coeffs = L_last / prm->expansion ? L_last / prm->expansion : 1;
        for (size_t j = max_coeffs; j < L_last; ++j) if (!fp_is_zero(co[j])) return reject("verify: last codeword is not of low degree");
    }
    if (R < 2) { *accepted = 1; return SH_OK; }
    uint8_t seed[32];
    SHCHK(sh_proofstream_verifier_fiat_shamir(ps, seed));
    std::vector<uint64_t> top;
    if (sample_indices(seed, N / 2, L_last, t, &top) != SH_OK) return reject("verify: cannot sample the test indices for these parameters");
    if (top_level_indices) memcpy(top_level_indices, top.data(), top.size() * 8);
    if (n_indices) *n_indices = top.size();
    const fp inv2 = fp_inv(fp_make(2, 0, 0, 0));
    fp off_r = off, w_r = host_root_of_unity(log_N);
    for (uint32_t r = 0; r + 1 < R; ++r) {
        const size_t L = N >> r;
        const uint32_t d0 = ilog2(L), d1 = d0 - 1;
        std::vector<std::vector<uint64_t>> abc(t);
        for (uint32_t s_ = 0; s_ < t; ++s_) {
            if (sh_proofstream_pull(ps, &js, &jl) != SH_OK || !host_parse_elems(js, jl, "Leaves", &abc[s_]) || abc[s_].size() != 6) return reject("verify: expected a Leaves item with three values");
            const fp ay = fp_from_u64x2(abc[s_][0], abc[s_][1]), by = fp_from_u64x2(abc[s_][2], abc[s_][3]), cy = fp_from_u64x2(abc[s_][4], abc[s_][5]);
            if (!fp_is_canonical(ay) || !fp_is_canonical(by) || !fp_is_canonical(cy)) return reject("verify: non-canonical value in a Leaves item");
            // colinearity of (ax, ay), (-ax, by), (alpha, cy), ax = off_r·w_r^a  <=>  cy = fold(ay, by) (G4)
            const uint64_t a = top[s_] % (L / 2);
            const fp ax = fp_mul(off_r, fp_pow(w_r, a));
            const fp q = fp_mul(alphas[r], fp_inv(ax));
            const fp want = fp_mul(inv2, fp_add(fp_mul(fp_add(fp_one(), q), ay), fp_mul(fp_sub(fp_one(), q), by)));
            if (!fp_eq(want, cy)) return reject("verify: colinearity check failed");
        }
        for (uint32_t s_ = 0; s_ < t; ++s_) {
            const uint64_t a = top[s_] % (L / 2);
            const uint64_t where[3] = {a, a + L / 2, a};
            for (int k = 0; k < 3; ++k) {
                std::vector<uint8_t> path;
                const uint32_t depth = k < 2 ? d0 : d1;
                if (sh_proofstream_pull(ps, &js, &jl) != SH_OK || !host_parse_path(js, jl, &path) || path.size() != (size_t)depth * 64) return reject("verify: expected a Path item of the tree's depth");
                uint8_t le[16], leaf[64];
                memcpy(le, &abc[s_][2 * k], 8); memcpy(le + 8, &abc[s_][2 * k + 1], 8);
                host_blake2b512(le, 16, leaf);
                int32_t ok = 0;
                SHCHK(sh_merkle_verify(roots.data() + 64 * (size_t)(r + (k == 2 ? 1 : 0)), leaf, path.data(), depth, where[k], &ok));
                if (!ok) return reject("verify: Merkle authentication path check failed");
            }
        }
        off_r = fp_mul(off_r, off_r); w_r = fp_mul(w_r, w_r);
    }
    *accepted = 1;
    return SH_OK;
}

}  // extern "C"

// ------------------------------------------------------------------ host-only scalar helpers
int32_t sh_sample_field(const uint8_t digest[32], uint64_t out[2]) {          // G3: big-endian integer mod p
    if (!digest || !out) return sh_fail(SH_EINVAL, "sh_sample_field: null pointer");
    uint64_t w[4] = {0, 0, 0, 0};                                            // w[0] most significant
    for (int i = 0; i < 32; ++i) w[i / 8] = (w[i / 8] << 8) | digest[i];
    const fp hi = fp_reduce_once(fp_from_u64x2(w[1], w[0])), lo = fp_reduce_once(fp_from_u64x2(w[3], w[2]));
    fp_store_host(out, fp_add(mont_mul(hi, fp_mont_r2()), lo));             // hi·2^128 + lo (mod p)
    return SH_OK;
}
int32_t sh_root_of_unity(uint32_t log_n, uint64_t out[2]) {                  // G1
    if (!out) return sh_fail(SH_EINVAL, "sh_root_of_unity: null pointer");
    if (log_n > 119) return sh_fail(SH_EINVAL, "sh_root_of_unity: the field has no root of unity of order > 2^119");
    fp_store_host(out, host_root_of_unity(log_n));
    return SH_OK;
}
int32_t sh_generator(uint64_t out[2]) {
    if (!out) return sh_fail(SH_EINVAL, "sh_generator: null pointer");
    fp_store_host(out, host_generator());
    return SH_OK;
}
int32_t sh_fp_scalar(int32_t op, const uint64_t a[2], const uint64_t b[2], uint64_t out[2]) {
    if (!a || !b || !out) return sh_fail(SH_EINVAL, "sh_fp_scalar: null pointer");
    fp x = fp_load_host(a), y = fp_load_host(b), r;
    if (!fp_is_canonical(x) || (op != 4 && !fp_is_canonical(y))) return sh_fail(SH_EINVAL, "sh_fp_scalar: non-canonical operand");
    switch (op) {
    case 0: r = fp_add(x, y); break;                                         // field.rs:57-68
    case 1: r = fp_sub(x, y); break;                                         // field.rs:70-82
    case 2: r = fp_mul(x, y); break;                                         // field.rs:84-95
    case 3:                                                                  // field.rs:97-114
        if (fp_is_zero(y)) return sh_fail(SH_EDIVZERO, "Cannot divide by zero");
        r = fp_mul(x, fp_inv(y));
        break;
    case 4: r = fp_pow(x, fp_to_u128(y)); break;                             // true modpow
    default: return sh_fail(SH_EINVAL, "sh_fp_scalar: unknown op");
    }
    fp_store_host(out, r);
    return SH_OK;
}

}  // extern "C"
