// api.hip — the C ABI of include/stark_hip.h: context, device buffers, and the host drivers that
// sequence the gfx950 kernels (field batch ops, NTT/LDE, Merkle commit, FRI fold and FRI commit).
// No entry point here computes a result on the CPU: without a usable device they return SH_EHIP.
#include "api_internal.h"

// ------------------------------------------------------------------ errors
static thread_local std::string g_last_error;
int32_t sh_fail(int32_t code, const char* msg) { g_last_error = msg ? msg : ""; return code; }
int32_t sh_failf(int32_t code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_last_error = buf;
    return code;
}
int32_t ensure_scratch(sh_ctx* c, int slot, size_t bytes, void** out) {
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

// [lo | hi] Montgomery-form power table of the (inverse) n-th root of unity, cached per context.
int32_t get_twiddles(sh_ctx* c, uint32_t log_n, bool inverse, const fp** out) {
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

// Direct tables of the inter-pass twiddles (kernels.h: NttPassTables) for the passes whose table is small enough to stay in L2
// across the tiles that share it (m <= 2^STARK_HIP_NTT_PASS_TW_LOG elements, default 2^20 = 16 MiB): every pass but the first
// of a large transform.  Built once per (size, direction, pass) and cached with the context.
int32_t get_pass_tables(sh_ctx* c, const NttPlan& plan, bool inverse, const fp* tw, NttPassTables* out) {
    static const uint32_t max_log = [] { const char* e = getenv("STARK_HIP_NTT_PASS_TW_LOG"); const int v = e ? atoi(e) : 20; return (uint32_t)(v < 0 ? 0 : v > 26 ? 26 : v); }();
    uint32_t consumed = 0;
    for (uint32_t pi = 0; pi < NTT_MAX_PASSES; ++pi) out->t[pi] = nullptr;
    out->scale_folded = false;
    for (uint32_t pi = 0; pi + 1 < plan.num_passes; ++pi) {
        const uint32_t log_m = plan.log_n - consumed;
        consumed += plan.bits[pi];
        if (log_m > max_log) continue;
        const uint32_t key = (plan.log_n * 2 + (inverse ? 1 : 0)) * NTT_MAX_PASSES + pi;
        auto it = c->ptw_cache.find(key);
        if (it == c->ptw_cache.end()) {
            fp* d = nullptr;
            HIPCHK(hipMalloc(&d, ((size_t)1 << log_m) * sizeof(fp)));
            const hipError_t e = launch_pass_twiddles(d, tw, plan.log_n, log_m, plan.bits[pi], c->stream);
            if (e != hipSuccess) { (void)hipFree(d); return sh_failf(SH_EHIP, "pass twiddle table: %s", hipGetErrorString(e)); }
            if (inverse && pi + 2 == plan.num_passes) {            // the inverse transform's 1/n: every element passes this table exactly once
                const fp n_inv = fp_to_mont(fp_inv(fp_from_u64x2((uint64_t)1 << plan.log_n, 0)));
                const hipError_t e2 = launch_scale_powers(d, (size_t)1 << log_m, n_inv, fp_one(), c->stream);
                if (e2 != hipSuccess) { (void)hipFree(d); return sh_failf(SH_EHIP, "pass twiddle table: %s", hipGetErrorString(e2)); }
            }
            it = c->ptw_cache.emplace(key, d).first;
        }
        if (inverse && pi + 2 == plan.num_passes) out->scale_folded = true;
        out->t[pi] = it->second;
    }
    return SH_OK;
}

// STARK_HIP_DEBUG_BT=1: SIGUSR1 makes the receiving thread print its native stack (tools/dump_threads.py sends it to every
// thread of a stuck process and resolves the offsets) — the box has no debugger.
#include <execinfo.h>
#include <signal.h>
#include <sys/syscall.h>
#include <unistd.h>
static void bt_handler(int) {
    void* frames[48];
    const int n = backtrace(frames, 48);
    char head[64];
    const int len = snprintf(head, sizeof head, "---- native stack of tid %ld\n", (long)syscall(SYS_gettid));
    if (len > 0) (void)!write(2, head, (size_t)len);
    backtrace_symbols_fd(frames, n, 2);
}
static void maybe_install_bt_handler() {
    static std::once_flag once;
    std::call_once(once, [] { if (getenv("STARK_HIP_DEBUG_BT")) { struct sigaction sa{}; sa.sa_handler = bt_handler; sa.sa_flags = SA_RESTART; sigaction(SIGUSR1, &sa, nullptr); } });
}

extern "C" {

const char* sh_last_error(void) { return g_last_error.c_str(); }
const char* sh_version(void) { return "stark_hip 0.1 (gfx950)"; }

int32_t sh_ctx_create(int32_t device, void* hip_stream, sh_ctx** out) {
    if (!out) return sh_fail(SH_EINVAL, "sh_ctx_create: null out");
    *out = nullptr;
    maybe_install_bt_handler();
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
    SHCHK(noalloc_init_device(device));            // resolves the fixed __device__ scratch of the no-alloc entry point on this GPU
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
uint64_t sh_ctx_chain_retries(const sh_ctx* c) { return c ? c->chain_retries + (c->sub ? c->sub->chain_retries : 0) : 0; }
int32_t sh_ctx_destroy(sh_ctx* c) {
    if (!c) return SH_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->sub) { (void)sh_ctx_destroy(c->sub); c->sub = nullptr; (void)hipSetDevice(c->device); }
    delete c->pool;
    delete c->enqueuer;
    if (getenv("STARK_HIP_DEBUG")) fprintf(stderr, "[stark_hip] ctx %p: mailbox roots %u, fallbacks %llu\n", (void*)c, c->mail_seq, (unsigned long long)c->mail_fallbacks);
    for (auto& kv : c->tw_cache) (void)hipFree(kv.second);
    for (auto& kv : c->ptw_cache) (void)hipFree(kv.second);
    for (int i = 0; i < NUM_SCRATCH; ++i) if (c->scratch[i]) (void)hipFree(c->scratch[i]);
    if (c->d_flag) (void)hipFree(c->d_flag);
    if (c->d_root) (void)hipFree(c->d_root);
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    if (c->h_mail) (void)hipHostFree(c->h_mail);
    if (c->h_batch) (void)hipHostFree(c->h_batch);
    if (c->h_chain) (void)hipHostFree(c->h_chain);
    if (c->c0_buf) (void)hipFree(c->c0_buf);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    for (auto& e : c->stage_ev) if (e) (void)hipEventDestroy(e);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->h_query) (void)hipHostFree(c->h_query);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return SH_OK;
}
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
    // A large batched inversion in three launches (field_kernels.hip): chunk products, one small inversion of those, finish —
    // instead of every workgroup waiting ~21 us per chunk for its own single-lane inversion.
    static const bool flat = getenv("STARK_HIP_BINV_FLAT") != nullptr;
    static const uint32_t split_log = [] { const char* e = getenv("STARK_HIP_BINV_SPLIT_LOG"); const int v = e ? atoi(e) : 0; return (uint32_t)(v >= 12 && v <= 40 ? v : 22); }();
    const bool inv_op = (op == FPOP_DIV || op == FPOP_INV || op == FPOP_INV_MONT);
    if (inv_op && !flat && n >= ((size_t)1 << split_log)) {
        const size_t chunks = fp_batch_inv_chunks(n);
        void* ext;
        SHCHK(ensure_scratch(c, SCR_BINV, chunks * sizeof(fp), &ext));
        HIPCHK(launch_fp_batch_inv_products(op, (const fp*)d_a, (const fp*)d_b, n, (fp*)ext, c->d_flag, c->stream));
        HIPCHK(launch_fp_binop(op == FPOP_INV ? FPOP_INV : FPOP_INV_MONT, (const fp*)ext, nullptr, (fp*)ext, chunks, c->d_flag, c->stream));
        HIPCHK(launch_fp_batch_inv_finish(op, (const fp*)d_a, (const fp*)d_b, (fp*)d_out, n, (fp*)ext, c->d_flag, c->stream));
    } else
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
    const int32_t rc = dev_fp_op(c, op, (const uint64_t*)da, (const uint64_t*)db, (uint64_t*)dout, n);
    if (rc != SH_OK && rc != SH_EDIVZERO) return rc;
    HIPCHK(hipMemcpyAsync(out, dout, n * 16, hipMemcpyDeviceToHost, c->stream));   // a zero divisor: 0 there, every other result is valid
    HIPCHK(hipStreamSynchronize(c->stream));
    return rc;
}

// ------------------------------------------------------------------ NTT / LDE
int32_t dev_ntt(sh_ctx* c, fp* d_data, uint32_t log_n, bool inverse, size_t batch) {
    if (log_n > SH_MAX_LOG_N) return sh_failf(SH_EINVAL, "ntt: log_n %u > %d", log_n, SH_MAX_LOG_N);
    if (batch == 0) return SH_OK;
    if (batch > ((size_t)1 << (SH_MAX_LOG_N + 2 - log_n))) return sh_failf(SH_EINVAL, "ntt: batch %zu x 2^%u exceeds 2^%d elements", batch, log_n, SH_MAX_LOG_N + 2);
    const size_t n = ((size_t)1 << log_n) * batch;
    const fp* tw;
    SHCHK(get_twiddles(c, log_n, inverse, &tw));
    void* scratch;
    SHCHK(ensure_scratch(c, SCR_WORK, n * sizeof(fp), &scratch));
    NttPlan plan;
    ntt_make_plan(log_n, &plan, batch);
    fp scale = fp_mont_one();
    if (inverse) scale = fp_to_mont(fp_inv(fp_from_u64x2((uint64_t)1 << log_n, 0)));
    NttPassTables ptab;
    SHCHK(get_pass_tables(c, plan, inverse, tw, &ptab));
    HIPCHK(launch_ntt(plan, d_data, (fp*)scratch, tw, scale, inverse && !ptab.scale_folded, batch, c->stream, nullptr, &ptab));
    return SH_OK;
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
int32_t sh_dev_coset_lde_batch(sh_ctx* c, const uint64_t* d_coeffs, size_t batch, uint32_t log_n, uint32_t log_expansion, const uint64_t offset[2], uint64_t* d_out) {
    CTXCHK(c);
    if (!batch) return SH_OK;
    if (!d_coeffs || !offset || !d_out) return sh_fail(SH_EINVAL, "sh_dev_coset_lde: null pointer");
    if (log_n + log_expansion > SH_MAX_LOG_N) return sh_failf(SH_EINVAL, "sh_dev_coset_lde: log_n + log_expansion > %d", SH_MAX_LOG_N);
    const uint32_t log_N = log_n + log_expansion;
    if (batch > 65535 || batch > ((size_t)1 << (SH_MAX_LOG_N + 2 - log_N))) return sh_failf(SH_EINVAL, "sh_dev_coset_lde_batch: batch %zu x 2^%u exceeds 65535 polynomials or 2^%d elements", batch, log_N, SH_MAX_LOG_N + 2);
    const fp off = fp_load_host(offset);
    if (!fp_is_canonical(off)) return sh_fail(SH_EINVAL, "sh_dev_coset_lde: non-canonical offset");
    const size_t n = (size_t)1 << log_n, N = n << log_expansion;
    // Aliasing: in place (d_coeffs == d_out) is defined for ONE polynomial only; with batch > 1 polynomial 0's padded output
    // would overwrite coefficients not yet read.  Any other overlap of the two ranges is an error too.
    {
        const uintptr_t ci = (uintptr_t)d_coeffs, ce = ci + batch * n * sizeof(fp), oi = (uintptr_t)d_out, oe = oi + batch * N * sizeof(fp);
        const bool overlap = ci < oe && oi < ce;
        if (overlap && !(batch == 1 && ci == oi)) return sh_fail(SH_EINVAL, "sh_dev_coset_lde_batch: input and output overlap (in place is defined for batch == 1 with d_coeffs == d_out only)");
    }
    NttPlan plan;
    ntt_make_plan(log_N, &plan, batch);
    NttLdeShape sh;
    static const bool no_fuse = getenv("STARK_NO_LDE_FUSE") != nullptr;
    if (no_fuse || d_coeffs == d_out || !ntt_lde_fusable(plan, log_expansion, &sh)) {
        HIPCHK(launch_coset_scale_pad((const fp*)d_coeffs, n, off, (fp*)d_out, N, c->stream, batch));
        return dev_ntt(c, (fp*)d_out, log_N, false, batch);
    }
    // Fused form: no zero-padded, pre-scaled copy of the coefficients is ever written — the first NTT pass reads the n
    // coefficients, scales them by offset^j on the way into LDS and never touches the padding (one HBM pass less).
    const fp* tw;
    SHCHK(get_twiddles(c, log_N, false, &tw));
    void *scratch, *tab;
    SHCHK(ensure_scratch(c, SCR_WORK, batch * N * sizeof(fp), &scratch));
    SHCHK(ensure_scratch(c, SCR_LDE_TAB, (sh.rows + sh.chunks + sh.cols) * sizeof(fp), &tab));
    fp* pow_row = (fp*)tab, *pow_chunk = pow_row + sh.rows, *pow_col = pow_chunk + sh.chunks;
    auto pow2k = [&](uint32_t k) { fp b = fp_to_mont(off); for (uint32_t i = 0; i < k; ++i) b = mont_mul(b, b); return b; };   // offset^(2^k), Montgomery form
    const sh_ctx::LdeTabKey key{{offset[0], offset[1]}, sh.rows, sh.chunks, sh.cols, sh.log_row_stride, sh.log_cols, tab};
    const sh_ctx::LdeTabKey& have = c->lde_tab_key;
    const bool cached = have.ptr == key.ptr && have.off[0] == key.off[0] && have.off[1] == key.off[1] && have.rows == key.rows && have.chunks == key.chunks &&
                        have.cols == key.cols && have.log_row_stride == key.log_row_stride && have.log_cols == key.log_cols;
    if (!cached) {                                     // same coset, same shape as the last call on this context (same stream): the tables are still there
        c->lde_tab_key = sh_ctx::LdeTabKey{};
        HIPCHK(launch_pow_table(pow_row, sh.rows, pow2k(sh.log_row_stride), c->stream));
        HIPCHK(launch_pow_table(pow_chunk, sh.chunks, pow2k(sh.log_cols), c->stream));
        HIPCHK(launch_pow_table(pow_col, sh.cols, pow2k(0), c->stream));
        c->lde_tab_key = key;
    }
    const NttLdeFuse fuse{(const fp*)d_coeffs, log_expansion, pow_row, pow_chunk, pow_col};
    NttPassTables ptab;
    SHCHK(get_pass_tables(c, plan, false, tw, &ptab));
    HIPCHK(launch_ntt(plan, (fp*)d_out, (fp*)scratch, tw, fp_mont_one(), false, batch, c->stream, &fuse, &ptab));
    return SH_OK;
}
int32_t sh_dev_coset_lde(sh_ctx* c, const uint64_t* d_coeffs, uint32_t log_n, uint32_t log_expansion, const uint64_t offset[2], uint64_t* d_out) {
    return sh_dev_coset_lde_batch(c, d_coeffs, 1, log_n, log_expansion, offset, d_out);
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

}  // extern "C"

extern "C" {
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
