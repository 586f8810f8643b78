// api_dist.hip — ONE transform / ONE commit spread over the P GPUs of a node, behind the C ABI (sh_dist_*; SURVEY.md §8e, BASELINE.json
// config 5).  One process (or host thread) per GPU; every rank makes the same call with its own slice.  No counterpart in the
// reference (a single-threaded CPU crate): the data model is the library's own — a vector of n field elements lies across the
// ranks in natural order, rank g holding elements [g·n/P, (g+1)·n/P).
//
// Transport.  The ranks exchange DEVICE buffers through three operations (all-to-all, all-gather, grouped point-to-point).
// sh_dist_create binds them to RCCL (librccl is dlopen()ed, so the library loads and the single-GPU entry points work where RCCL
// is absent): grouped ncclSend / ncclRecv and ncclAllGather over xGMI, enqueued on the context's stream.
// sh_dist_create_with_transport takes the three operations from the host instead (MPI, or the gloo-backed callbacks of the test
// suite, which stage through host memory) — the orchestration below is the same code either way.
//
// Four-step NTT (n = n1·n2, j = j1·n2 + j2, k = k1 + n1·k2), natural order in and out:
//   A  exchange-transpose rows j1 -> columns j2        (pack = k_transpose16, all-to-all, unpack = k_interleave_blocks)
//   B  n2/P batched n1-point NTTs over j1, then ·w_n^(j2·k1)
//   C  exchange-transpose columns j2 -> rows k1
//   D  n1/P batched n2-point NTTs over j2
//   E  exchange-transpose rows k1 -> natural order
// A is skipped for SH_DIST_IN_COLUMNS, E for SH_DIST_OUT_ROWS (then exactly one all-to-all).  One rank: the direct transform.
//
// Sharded FRI commit.  Round r works on codeword r of length L.  Rank g holds half-slices g and P+g of it back to back
// ([lo | hi], h = L/2P elements each), so cw[i] and cw[L/2+i] are on one card and the split-and-fold is local — and FUSED with the
// leaf hashing of its result, exactly as on one GPU (k_fold_leaf_subtree with a slice offset): the h outputs are slice g of
// codeword r+1, a contiguous block of leaves, i.e. one aligned subtree of that round's Merkle tree.  Its root stays on the
// device; a 64-byte all-gather and one four-lanes-per-compression launch over the P sub-roots publish the round's root into the
// same pinned chain record the single-GPU commit uses.  Every rank serves an identical transcript replica from its own record
// (no HIP call on that thread, api_fri.hip), and all launches and collectives of all sharded rounds are enqueued up front by the
// context's enqueue thread.  After the fold the two halves of the h outputs move point to point to the ranks that own them as
// half-slices of the next round.  Once half-slices would drop under `min_half` elements the codeword is all-gathered and every
// rank finishes the remaining rounds with the single-GPU driver (the round count depends on the current length only, G5).
#include <dlfcn.h>
#include "api_internal.h"

// ------------------------------------------------------------------ RCCL, bound at run time
namespace {
struct nccl_uid { char internal[128]; };
typedef void* nccl_comm_t;
enum { NCCL_UINT8 = 1 };
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(nccl_uid*) = nullptr;
    int (*CommInitRank)(nccl_comm_t*, int, nccl_uid, int) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string error;
};
RcclApi g_rccl;
std::once_flag g_rccl_once;

// A process that has torch loaded already holds torch's librccl (same soname): dlopen returns that copy, which shares the HIP
// runtime the library itself runs on.  Otherwise the ROCm installation's.
void rccl_load() {
    const char* env = getenv("STARK_HIP_RCCL_PATH");
    const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
        if (!n || !*n) continue;
        g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.lib) break;
        g_rccl.error = dlerror();
    }
    if (!g_rccl.lib) { g_rccl.error = "cannot load librccl (" + g_rccl.error + "); set STARK_HIP_RCCL_PATH"; return; }
    bool ok = true;
    auto sym = [&](const char* name) { void* p = dlsym(g_rccl.lib, name); if (!p) { ok = false; g_rccl.error = std::string("librccl lacks ") + name; } return p; };
    g_rccl.GetUniqueId = (int (*)(nccl_uid*))sym("ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(nccl_comm_t*, int, nccl_uid, int))sym("ncclCommInitRank");
    g_rccl.CommDestroy = (int (*)(nccl_comm_t))sym("ncclCommDestroy");
    g_rccl.GroupStart = (int (*)())sym("ncclGroupStart");
    g_rccl.GroupEnd = (int (*)())sym("ncclGroupEnd");
    g_rccl.Send = (int (*)(const void*, size_t, int, int, nccl_comm_t, hipStream_t))sym("ncclSend");
    g_rccl.Recv = (int (*)(void*, size_t, int, int, nccl_comm_t, hipStream_t))sym("ncclRecv");
    g_rccl.AllGather = (int (*)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t))sym("ncclAllGather");
    g_rccl.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
    if (!ok) { dlclose(g_rccl.lib); g_rccl.lib = nullptr; }
}
int32_t rccl_get() {
    std::call_once(g_rccl_once, rccl_load);
    if (!g_rccl.lib) return sh_failf(SH_EHIP, "RCCL unavailable: %s", g_rccl.error.c_str());
    return SH_OK;
}
}  // namespace

struct sh_dist {
    sh_ctx* c = nullptr;
    int32_t rank = 0, world = 1;
    sh_dist_transport t{};
    nccl_comm_t comm = nullptr;                  // RCCL transport only
};

#define NCCLCHK(expr)                                                                                                          \
    do {                                                                                                                       \
        const int _r = (expr);                                                                                                 \
        if (_r != 0) return sh_failf(SH_EHIP, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "?", __FILE__, __LINE__); \
    } while (0)

static int32_t rccl_all_to_all(void* user, const void* d_send, void* d_recv, size_t bytes, void* stream) {
    sh_dist* d = (sh_dist*)user;
    NCCLCHK(g_rccl.GroupStart());
    for (int s = 0; s < d->world; ++s) {
        NCCLCHK(g_rccl.Send((const char*)d_send + (size_t)s * bytes, bytes, NCCL_UINT8, s, d->comm, (hipStream_t)stream));
        NCCLCHK(g_rccl.Recv((char*)d_recv + (size_t)s * bytes, bytes, NCCL_UINT8, s, d->comm, (hipStream_t)stream));
    }
    NCCLCHK(g_rccl.GroupEnd());
    return SH_OK;
}
static int32_t rccl_all_gather(void* user, const void* d_send, void* d_recv, size_t bytes, void* stream) {
    sh_dist* d = (sh_dist*)user;
    NCCLCHK(g_rccl.AllGather(d_send, d_recv, bytes, NCCL_UINT8, d->comm, (hipStream_t)stream));
    return SH_OK;
}
static int32_t rccl_sendrecv(void* user, const sh_p2p_op* ops, size_t n, void* stream) {
    sh_dist* d = (sh_dist*)user;
    NCCLCHK(g_rccl.GroupStart());
    for (size_t i = 0; i < n; ++i) {
        if (ops[i].is_send) NCCLCHK(g_rccl.Send(ops[i].d_buf, ops[i].bytes, NCCL_UINT8, ops[i].peer, d->comm, (hipStream_t)stream));
        else NCCLCHK(g_rccl.Recv(ops[i].d_buf, ops[i].bytes, NCCL_UINT8, ops[i].peer, d->comm, (hipStream_t)stream));
    }
    NCCLCHK(g_rccl.GroupEnd());
    return SH_OK;
}
// a transport operation failed: RCCL's own message stands; a host-supplied transport gets a generic one
#define TCHK(expr, what) do { if ((expr) != 0) return d->comm ? SH_EHIP : sh_fail(SH_EHIP, what ": the host's transport callback reported a failure"); } while (0)

static int32_t dist_check(sh_dist* d) {
    if (!d || !d->c) return sh_fail(SH_EINVAL, "null sh_dist");
    HIPCHK(hipSetDevice(d->c->device));
    return SH_OK;
}

// ------------------------------------------------------------------ exchanges
// The global matrix is (P·a) x (P·b) elements, row-major, this rank holding rows [g·a, (g+1)·a) as x (a x P·b).  Leaves this
// rank's columns [g·b, (g+1)·b) as rows in `out` (b x P·a).  t1, t2: scratch of a·P·b elements each; `out` may be t1's peer only
// (x, t1, t2, out: x may equal t2; out must differ from t1 and t2 when P > 1).
// tw != null: the pack also multiplies element (r, c) by w_n^((row0 + r)·c), the twiddle between the two transform steps.
static int32_t exchange_transpose(sh_dist* d, const fp* x, size_t a, size_t cols, fp* t1, fp* t2, fp* out, const fp* tw = nullptr, uint32_t log_n = 0, uint64_t row0 = 0) {
    sh_ctx* c = d->c;
    const size_t P = (size_t)d->world, b = cols / P;
    fp* packed = P == 1 ? out : t1;                                                          // [dest s][col c][row r]
    if (tw) HIPCHK(launch_transpose16_tw(x, packed, a, cols, tw, log_n, row0, c->stream));
    else HIPCHK(launch_transpose16(x, packed, a, cols, c->stream));
    if (P == 1) return SH_OK;
    TCHK(d->t.all_to_all(d->t.user, t1, t2, b * a * sizeof(fp), c->stream), "all-to-all");   // [src s][col c][row of s]
    HIPCHK(launch_interleave_blocks(t2, out, P, b, a, c->stream));                           // [c][s][r]
    return SH_OK;
}

static int32_t dist_ntt_impl(sh_dist* d, const fp* d_local, uint32_t log_n, bool inverse, int32_t in_layout, int32_t out_layout, fp* d_out) {
    sh_ctx* c = d->c;
    const size_t P = (size_t)d->world, g = (size_t)d->rank;
    if (log_n > SH_MAX_LOG_N + 2) return sh_failf(SH_EINVAL, "sh_dist_ntt: log_n %u too large", log_n);
    const size_t n = (size_t)1 << log_n, m = n / P;
    const bool force4 = getenv("STARK_HIP_DIST_FORCE_FOURSTEP") != nullptr;          // tests and traces: the four-step code on one rank
    if (P == 1 && !force4 && in_layout == SH_DIST_NATURAL && out_layout == SH_DIST_NATURAL) {                   // one rank: the direct transform
        if (d_out != d_local) HIPCHK(hipMemcpyAsync(d_out, d_local, n * sizeof(fp), hipMemcpyDeviceToDevice, c->stream));
        return dev_ntt(c, d_out, log_n, inverse);
    }
    const uint32_t l1 = log_n / 2, l2 = log_n - l1;
    const size_t n1 = (size_t)1 << l1, n2 = (size_t)1 << l2;
    if (n1 % P || n2 % P) return sh_failf(SH_EINVAL, "sh_dist_ntt: a 2^%u x 2^%u matrix does not split over %zu ranks", l1, l2, P);
    void *va, *vb, *vc;
    SHCHK(ensure_scratch(c, SCR_DIST_A, m * sizeof(fp), &va));
    SHCHK(ensure_scratch(c, SCR_DIST_B, m * sizeof(fp), &vb));
    SHCHK(ensure_scratch(c, SCR_DIST_C, m * sizeof(fp), &vc));
    fp *A = (fp*)va, *B = (fp*)vb, *C = (fp*)vc;
    const fp* tw;
    SHCHK(get_twiddles(c, log_n, inverse, &tw));
    if (in_layout == SH_DIST_NATURAL) SHCHK(exchange_transpose(d, d_local, n1 / P, n2, B, C, A));               // A = (n2/P, n1): [j2l][j1]
    else if (in_layout == SH_DIST_IN_COLUMNS) HIPCHK(hipMemcpyAsync(A, d_local, m * sizeof(fp), hipMemcpyDeviceToDevice, c->stream));
    else return sh_fail(SH_EINVAL, "sh_dist_ntt: unknown input layout");
    SHCHK(dev_ntt(c, A, l1, inverse, n2 / P));                                                                  // [j2l][k1]
    // the twiddle w_n^(j2·k1) rides in the pack of the next exchange (row = j2, column = k1)
    if (out_layout == SH_DIST_OUT_ROWS) {
        SHCHK(exchange_transpose(d, A, n2 / P, n1, B, C, d_out, tw, log_n, g * (n2 / P)));                      // (n1/P, n2): [k1l][j2]
        return dev_ntt(c, d_out, l2, inverse, n1 / P);
    }
    if (out_layout != SH_DIST_NATURAL) return sh_fail(SH_EINVAL, "sh_dist_ntt: unknown output layout");
    fp* z = P == 1 ? B : A;                      // P > 1: the pack has consumed A before the unpack writes it
    SHCHK(exchange_transpose(d, A, n2 / P, n1, B, C, z, tw, log_n, g * (n2 / P)));                              // (n1/P, n2): [k1l][j2]
    SHCHK(dev_ntt(c, z, l2, inverse, n1 / P));                                                                  // [k1l][k2] = X[k1 + n1·k2]
    return exchange_transpose(d, z, n1 / P, n2, B, C, d_out);                                                   // [k2l][k1] = natural slice
}

// ------------------------------------------------------------------ sharded commit
// digests at `cur` (2^lg of them, one tree) -> its root: into the round's host record (publish) or to device memory at d_out.
static int32_t reduce_tree(sh_ctx* c, const digest64* cur, digest64* ping, digest64* pong, uint32_t lg, digest64* d_out, uint8_t* d_rec, uint32_t gen, const uint32_t* d_abort) {
    uint64_t* rec_roots = reinterpret_cast<uint64_t*>(d_rec);
    if (lg == 0) {
        if (d_rec) HIPCHK(launch_publish_digests(cur, 1, rec_roots, reinterpret_cast<uint32_t*>(d_rec + 64), gen, c->stream, d_abort));
        else HIPCHK(hipMemcpyAsync(d_out, cur, sizeof(digest64), hipMemcpyDeviceToDevice, c->stream));
        return SH_OK;
    }
    digest64* other = ping;
    while (lg > 0) {
        bool quad;
        const uint32_t step = forest_step(lg, 1, &quad);
        const size_t total = (size_t)1 << lg;
        if (!quad) { HIPCHK(launch_node_subtree(cur, other, total >> step, step, c->stream)); }
        else if (step == lg) {
            HIPCHK(launch_quad_reduce(cur, d_rec ? c->d_root : d_out, total, step, d_rec ? rec_roots : nullptr, d_rec ? gen : 0, c->stream, nullptr, 0, d_abort));
            return SH_OK;
        } else HIPCHK(launch_quad_reduce(cur, other, total, step, nullptr, 0, c->stream, nullptr, 0, d_abort));
        cur = other; other = (other == ping) ? pong : ping; lg -= step;
    }
    return SH_OK;
}

extern "C" {

int32_t sh_dist_unique_id(uint8_t id[SH_DIST_ID_BYTES]) {
    if (!id) return sh_fail(SH_EINVAL, "sh_dist_unique_id: null pointer");
    SHCHK(rccl_get());
    nccl_uid u;
    NCCLCHK(g_rccl.GetUniqueId(&u));
    memcpy(id, u.internal, SH_DIST_ID_BYTES);
    return SH_OK;
}
static int32_t dist_shape_ok(int32_t rank, int32_t world) {
    if (world < 1 || (world & (world - 1)) || world > 512) return sh_fail(SH_EINVAL, "sh_dist_create: the number of ranks must be a power of two <= 512");
    if (rank < 0 || rank >= world) return sh_fail(SH_EINVAL, "sh_dist_create: rank out of range");
    return SH_OK;
}
int32_t sh_dist_create(sh_ctx* ctx, int32_t rank, int32_t world, const uint8_t id[SH_DIST_ID_BYTES], sh_dist** out) {
    if (!out) return sh_fail(SH_EINVAL, "sh_dist_create: null out");
    *out = nullptr;
    CTXCHK(ctx);
    if (!id) return sh_fail(SH_EINVAL, "sh_dist_create: null unique id");
    SHCHK(dist_shape_ok(rank, world));
    SHCHK(rccl_get());
    sh_dist* d = new (std::nothrow) sh_dist();
    if (!d) return sh_fail(SH_ENOMEM, "sh_dist_create: out of memory");
    d->c = ctx; d->rank = rank; d->world = world;
    nccl_uid u;
    memcpy(u.internal, id, SH_DIST_ID_BYTES);
    const int r = g_rccl.CommInitRank(&d->comm, world, u, rank);
    if (r != 0) { delete d; return sh_failf(SH_EHIP, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r)); }
    d->t.user = d;
    d->t.all_to_all = rccl_all_to_all;
    d->t.all_gather = rccl_all_gather;
    d->t.sendrecv = rccl_sendrecv;
    *out = d;
    return SH_OK;
}
int32_t sh_dist_create_with_transport(sh_ctx* ctx, int32_t rank, int32_t world, const sh_dist_transport* t, sh_dist** out) {
    if (!out) return sh_fail(SH_EINVAL, "sh_dist_create_with_transport: null out");
    *out = nullptr;
    CTXCHK(ctx);
    SHCHK(dist_shape_ok(rank, world));
    if (world > 1 && (!t || !t->all_to_all || !t->all_gather || !t->sendrecv)) return sh_fail(SH_EINVAL, "sh_dist_create_with_transport: incomplete transport");
    sh_dist* d = new (std::nothrow) sh_dist();
    if (!d) return sh_fail(SH_ENOMEM, "sh_dist_create_with_transport: out of memory");
    d->c = ctx; d->rank = rank; d->world = world;
    if (t) d->t = *t;
    *out = d;
    return SH_OK;
}
int32_t sh_dist_destroy(sh_dist* d) {
    if (!d) return SH_OK;
    if (d->c) { (void)hipSetDevice(d->c->device); (void)hipStreamSynchronize(d->c->stream); }
    if (d->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(d->comm);
    delete d;
    return SH_OK;
}
int32_t sh_dist_rank(const sh_dist* d, int32_t* rank, int32_t* world) {
    if (!d || !rank || !world) return sh_fail(SH_EINVAL, "sh_dist_rank: null pointer");
    *rank = d->rank; *world = d->world;
    return SH_OK;
}

// Every operation of the transport on small buffers with known contents, checked on the host: all-gather, all-to-all, and a ring
// of point-to-point pairs (to itself on one rank).  What a host runs once after sh_dist_create.
int32_t sh_dist_selftest(sh_dist* d) {
    SHCHK(dist_check(d));
    sh_ctx* c = d->c;
    const size_t P = (size_t)d->world, g = (size_t)d->rank, W = 256;               // W u64 words per block
    if (!d->t.all_to_all || !d->t.all_gather || !d->t.sendrecv) return sh_fail(SH_EINVAL, "sh_dist_selftest: this handle has no transport (one rank, none given)");
    void *vs, *vr;
    SHCHK(ensure_scratch(c, SCR_DIST_A, P * W * 8, &vs));
    SHCHK(ensure_scratch(c, SCR_DIST_B, P * W * 8, &vr));
    std::vector<uint64_t> h(P * W), r(P * W);
    auto val = [&](size_t from, size_t to, size_t i) { return (uint64_t)(from * 1000003u + to * 10007u + i) * 0x9e3779b97f4a7c15ull; };
    for (size_t s = 0; s < P; ++s) for (size_t i = 0; i < W; ++i) h[s * W + i] = val(g, s, i);
    HIPCHK(hipMemcpyAsync(vs, h.data(), P * W * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemsetAsync(vr, 0, P * W * 8, c->stream));
    TCHK(d->t.all_to_all(d->t.user, vs, vr, W * 8, c->stream), "all-to-all");
    HIPCHK(hipMemcpyAsync(r.data(), vr, P * W * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (size_t s = 0; s < P; ++s) for (size_t i = 0; i < W; ++i) if (r[s * W + i] != val(s, g, i)) return sh_failf(SH_EHIP, "sh_dist_selftest: all-to-all block from rank %zu is wrong", s);
    HIPCHK(hipMemsetAsync(vr, 0, P * W * 8, c->stream));
    TCHK(d->t.all_gather(d->t.user, vs, vr, W * 8, c->stream), "all-gather");         // block 0 of every rank's send buffer
    HIPCHK(hipMemcpyAsync(r.data(), vr, P * W * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (size_t s = 0; s < P; ++s) for (size_t i = 0; i < W; ++i) if (r[s * W + i] != val(s, 0, i)) return sh_failf(SH_EHIP, "sh_dist_selftest: all-gather block of rank %zu is wrong", s);
    HIPCHK(hipMemsetAsync(vr, 0, W * 8, c->stream));
    const size_t next = (g + 1) % P, prev = (g + P - 1) % P;
    sh_p2p_op ops[2] = {{(int32_t)next, 1, (char*)vs + W * 8 * (P > 1 ? 1 : 0), W * 8}, {(int32_t)prev, 0, vr, W * 8}};   // send block (P > 1 ? 1 : 0)
    TCHK(d->t.sendrecv(d->t.user, ops, 2, c->stream), "send/recv");
    HIPCHK(hipMemcpyAsync(r.data(), vr, W * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < W; ++i) if (r[i] != val(prev, P > 1 ? 1 : 0, i)) return sh_fail(SH_EHIP, "sh_dist_selftest: point-to-point block is wrong");
    return SH_OK;
}

int32_t sh_dist_ntt(sh_dist* d, const uint64_t* d_local, uint32_t log_n, int32_t inverse, int32_t input_layout, int32_t output_layout, uint64_t* d_out) {
    SHCHK(dist_check(d));
    if (!d_local || !d_out) return sh_fail(SH_EINVAL, "sh_dist_ntt: null pointer");
    return dist_ntt_impl(d, (const fp*)d_local, log_n, inverse != 0, input_layout, output_layout, (fp*)d_out);
}

int32_t sh_dist_coset_lde(sh_dist* d, const uint64_t* d_local_coeffs, uint32_t log_n, uint32_t log_expansion, const uint64_t offset[2], uint64_t* d_out) {
    SHCHK(dist_check(d));
    sh_ctx* c = d->c;
    if (!d_local_coeffs || !offset || !d_out) return sh_fail(SH_EINVAL, "sh_dist_coset_lde: null pointer");
    const size_t P = (size_t)d->world, g = (size_t)d->rank;
    if (P == 1) return sh_dev_coset_lde(c, d_local_coeffs, log_n, log_expansion, offset, d_out);
    if (log_n + log_expansion > SH_MAX_LOG_N + 2) return sh_fail(SH_EINVAL, "sh_dist_coset_lde: domain too large");
    const fp off = fp_load_host(offset);
    if (!fp_is_canonical(off)) return sh_fail(SH_EINVAL, "sh_dist_coset_lde: non-canonical offset");
    const size_t n = (size_t)1 << log_n, N = n << log_expansion, m_in = n / P, m_out = N / P, E = (size_t)1 << log_expansion;
    if (n % P || !m_in) return sh_fail(SH_EINVAL, "sh_dist_coset_lde: the coefficients do not split over the ranks");
    if ((const fp*)d_local_coeffs == (fp*)d_out) return sh_fail(SH_EINVAL, "sh_dist_coset_lde: input and output must not alias");
    void* vs;
    SHCHK(ensure_scratch(c, SCR_DIST_C, m_out * sizeof(fp), &vs));     // dist_ntt's third buffer: free until its first all-to-all
    fp* scaled = (fp*)vs;
    HIPCHK(hipMemcpyAsync(scaled, d_local_coeffs, m_in * sizeof(fp), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(launch_scale_powers(scaled, m_in, fp_to_mont(fp_pow(off, g * m_in)), off, c->stream));
    // padded vector = [coeffs | zeros]; rank g owns padded[g·m_out, (g+1)·m_out), m_out = E·m_in: the coefficient slice of rank s
    // lands on rank s / E at offset (s mod E)·m_in.  Only those slices move, point to point; ranks >= P/E own zeros only.
    fp* mine = (fp*)d_out;
    HIPCHK(hipMemsetAsync(mine, 0, m_out * sizeof(fp), c->stream));
    std::vector<sh_p2p_op> ops;
    const size_t dst = g / E;
    if (dst == g) HIPCHK(hipMemcpyAsync(mine + (g % E) * m_in, scaled, m_in * sizeof(fp), hipMemcpyDeviceToDevice, c->stream));
    else ops.push_back({(int32_t)dst, 1, scaled, m_in * sizeof(fp)});
    for (size_t s = g * E; s < (g + 1) * E && s < P; ++s)
        if (s != g) ops.push_back({(int32_t)s, 0, mine + (s % E) * m_in, m_in * sizeof(fp)});
    if (!ops.empty()) TCHK(d->t.sendrecv(d->t.user, ops.data(), ops.size(), c->stream), "send/recv");
    return dist_ntt_impl(d, mine, log_n + log_expansion, false, SH_DIST_NATURAL, SH_DIST_NATURAL, mine);
}

int32_t sh_dist_all_gather(sh_dist* d, const uint64_t* d_local, size_t count, uint64_t* d_out) {
    SHCHK(dist_check(d));
    if (!count) return SH_OK;
    if (!d_local || !d_out) return sh_fail(SH_EINVAL, "sh_dist_all_gather: null pointer");
    if (d->world == 1) { if (d_out != d_local) HIPCHK(hipMemcpyAsync(d_out, d_local, count * 16, hipMemcpyDeviceToDevice, d->c->stream)); return SH_OK; }
    TCHK(d->t.all_gather(d->t.user, d_local, d_out, count * 16, d->c->stream), "all-gather");
    return SH_OK;
}

}  // extern "C"

static int32_t dist_fri_commit_impl(sh_dist* d, const fp* d_local, size_t N, const sh_fri_params* prm, sh_proofstream* ps, size_t min_half,
                                    uint8_t* roots_out, uint64_t* last_out, size_t* last_len) {
    sh_ctx* c = d->c;
    const size_t P = (size_t)d->world, g = (size_t)d->rank;
    fp off;
    SHCHK(check_params(prm, &off));
    if (!is_pow2(N) || N < 2 * P) return sh_fail(SH_EINVAL, "sh_dist_fri_commit: N must be a power of two >= 2·ranks");
    const uint32_t log_N = ilog2(N);
    if (log_N > SH_MAX_LOG_N + 2) return sh_fail(SH_EINVAL, "sh_dist_fri_commit: N too large");
    if (min_half < 1) min_half = 1;
    const size_t m = N / P;
    // S rounds are hashed sharded (codewords 0 .. S-1); codeword S is folded sharded, gathered, and committed locally.
    uint32_t S = 0;
    for (size_t L = N; L / (2 * P) >= min_half && sh_fri_num_rounds(L, prm->expansion, prm->num_colinearity_tests) > 1; L /= 2) ++S;
    const size_t L_tail = N >> S, h_tail = m >> S;               // codeword S: its length, this rank's slice of it
    if (log_N - S > SH_MAX_LOG_N) return sh_fail(SH_EINVAL, "sh_dist_fri_commit: the locally committed tail exceeds 2^28 elements; lower min_half");

    // ---- everything that allocates happens before the first launch
    void *chA = nullptr, *chB = nullptr, *bufA = nullptr, *bufB = nullptr, *digA, *digB, *d_c0, *vsub, *vfull;
    if (S >= 1) SHCHK(ensure_scratch(c, SCR_IO_A, (m / 2) * sizeof(fp), &chA));
    if (S >= 2) SHCHK(ensure_scratch(c, SCR_IO_B, (m / 4 ? m / 4 : 1) * sizeof(fp), &chB));
    if (P > 1 && S >= 1) {
        SHCHK(ensure_scratch(c, SCR_DIST_A, m * sizeof(fp), &bufA));
        SHCHK(ensure_scratch(c, SCR_DIST_B, (m / 2) * sizeof(fp), &bufB));
    }
    auto leaf_k = [&](size_t len) { uint32_t K = leaf_subtree_k(ilog2(len)); const uint32_t lg = ilog2(len); return K < lg ? K : lg; };
    const size_t dig0 = m >> leaf_k(m);
    SHCHK(ensure_scratch(c, SCR_DIG_A, dig0 * sizeof(digest64), &digA));
    SHCHK(ensure_scratch(c, SCR_DIG_B, (dig0 / 2 ? dig0 / 2 : 1) * sizeof(digest64), &digB));
    SHCHK(ensure_scratch(c, SCR_C0, sizeof(fp), &d_c0));
    SHCHK(ensure_scratch(c, SCR_DIST_SUB, (P + 1) * sizeof(digest64), &vsub));
    SHCHK(ensure_scratch(c, SCR_IO_C, L_tail * sizeof(fp), &vfull));
    digest64 *d_sub = (digest64*)vsub, *d_all = d_sub + 1;
    fp* full = (fp*)vfull;
    const fp* tw_inv = nullptr;
    if (S) SHCHK(get_twiddles(c, log_N, true, &tw_inv));
    auto chunk_of = [&](uint32_t r) -> fp* { return r == 0 ? const_cast<fp*>(d_local) : (fp*)((r & 1) ? chA : chB); };     // this rank's slice of codeword r (m >> r elements)
    auto buf_of = [&](uint32_t r) -> fp* { return P == 1 ? chunk_of(r) : (fp*)((r & 1) ? bufB : bufA); };                 // its [lo | hi] half-slices of codeword r

    int32_t rc_tail = SH_OK;
    if (S) {
        const ChainLayout lay(1, S, 0);
        SHCHK(ensure_chain(c, lay.total, lay.off_last));
        uint32_t gen = (c->chain_gen + 1) & 0x7fffffffu;
        if (gen == 0) gen = 1;
        c->chain_gen = gen;
        // No re-launch here: a rank that enqueued a round twice would run its collectives twice and the others would not.  The GPU-side
        // wait is bounded by the same wall clock as the host's wait for a root; running out of it fails the commit.
        const double timeout_s = [] { const char* e = getenv("STARK_HIP_CHAIN_TIMEOUT_S"); double v = e ? atof(e) : 0; return v > 0 ? v : 20.0; }();
        const uint64_t timeout_ticks = (uint64_t)(timeout_s * 1e8) + 1;
        uint32_t* d_abort = c->d_flag + 32;
        uint32_t* h_abort = reinterpret_cast<uint32_t*>(c->h_chain);
        uint32_t* hd_abort = reinterpret_cast<uint32_t*>(c->d_chain);
        for (int i = 0; i < 16; ++i) h_abort[i] = 0;
        for (uint32_t r = 0; r < S; ++r) {
            *reinterpret_cast<volatile uint32_t*>(c->h_chain + lay.root_seq(r)) = 0;
            *reinterpret_cast<volatile uint32_t*>(c->c0_host + lay.c0_seq(r)) = 0;
        }
        __atomic_thread_fence(__ATOMIC_SEQ_CST);
        HIPCHK(hipMemsetAsync(d_abort, 0, 4, c->stream));

        auto wait_c0 = [&](uint32_t r) -> int32_t {              // the kernels behind it read c0(r) from d_c0
            HIPCHK(launch_wait_c0(reinterpret_cast<const uint32_t*>(c->c0_dev + lay.c0_seq(r)), gen, reinterpret_cast<const fp*>(c->c0_dev + lay.c0(r)), (fp*)d_c0, 1,
                                  d_abort, hd_abort, timeout_ticks, c->stream));
            return SH_OK;
        };
        // chunk r (two halves of hl elements) -> the ranks that hold them as half-slices of codeword r: half 2g+q goes to rank
        // (2g+q) mod P, as its lo when 2g+q < P, else its hi; this rank's lo comes from rank g/2, its hi from rank (P+g)/2.
        auto regroup = [&](uint32_t r) -> int32_t {
            if (P == 1) return SH_OK;
            fp* chunk = chunk_of(r);
            fp* buf = buf_of(r);
            const size_t hl = (m >> r) / 2;
            std::vector<sh_p2p_op> ops;
            for (size_t q = 0; q < 2; ++q) {
                const size_t idx = 2 * g + q, dst = idx % P;
                if (dst == g) HIPCHK(hipMemcpyAsync(buf + (idx < P ? 0 : hl), chunk + q * hl, hl * sizeof(fp), hipMemcpyDeviceToDevice, c->stream));
                else ops.push_back({(int32_t)dst, 1, chunk + q * hl, hl * sizeof(fp)});
            }
            const size_t src_lo = g / 2, src_hi = (P + g) / 2;
            if (src_lo != g) ops.push_back({(int32_t)src_lo, 0, buf, hl * sizeof(fp)});
            if (src_hi != g) ops.push_back({(int32_t)src_hi, 0, buf + hl, hl * sizeof(fp)});
            if (!ops.empty()) TCHK(d->t.sendrecv(d->t.user, ops.data(), ops.size(), c->stream), "send/recv");
            return SH_OK;
        };
        auto enqueue_all = [&]() -> int32_t {
            for (uint32_t r = 0; r < S; ++r) {
                const size_t outs = m >> r;                      // this rank's elements of codeword r
                const uint32_t K = leaf_k(outs);
                digest64* dst = (digest64*)digA;
                if (r == 0) HIPCHK(launch_leaf_subtree(d_local, dst, outs >> K, K, c->stream));
                else {
                    if (c->c0_buf && c->c0_dev == c->c0_buf) {   // records on the card: the fold kernel waits itself (api_fri.hip, chain_wait.cuh)
                        const ChainWait cwt{reinterpret_cast<const uint32_t*>(c->c0_dev + lay.c0_seq(r - 1)), gen, d_abort, hd_abort, timeout_ticks};
                        HIPCHK(launch_fold_leaf_subtree(buf_of(r - 1), 2 * outs, 1, reinterpret_cast<const fp*>(c->c0_dev + lay.c0(r - 1)), fp_zero(), tw_inv, log_N, r - 1, chunk_of(r), dst,
                                                        K, c->stream, d_abort, g * outs, &cwt));
                    } else {
                        SHCHK(wait_c0(r - 1));
                        HIPCHK(launch_fold_leaf_subtree(buf_of(r - 1), 2 * outs, 1, (const fp*)d_c0, fp_zero(), tw_inv, log_N, r - 1, chunk_of(r), dst, K, c->stream, d_abort, g * outs));
                    }
                }
                uint8_t* rec = c->d_chain + lay.roots(r);
                if (P == 1) SHCHK(reduce_tree(c, dst, (digest64*)digB, dst, ilog2(outs) - K, nullptr, rec, gen, d_abort));
                else {
                    SHCHK(reduce_tree(c, dst, (digest64*)digB, dst, ilog2(outs) - K, d_sub, nullptr, 0, d_abort));
                    TCHK(d->t.all_gather(d->t.user, d_sub, d_all, sizeof(digest64), c->stream), "all-gather");
                    HIPCHK(launch_quad_reduce(d_all, c->d_root, P, ilog2(P), reinterpret_cast<uint64_t*>(rec), gen, c->stream, nullptr, 0, d_abort));
                }
                SHCHK(regroup(r));
            }
            // codeword S: folded sharded (no hashing: the local driver below hashes the gathered codeword), then gathered
            SHCHK(wait_c0(S - 1));
            const size_t outs = h_tail;
            fp winv = fp_inv(host_root_of_unity(log_N));
            for (uint32_t i = 0; i + 1 < S; ++i) winv = fp_mul(winv, winv);          // w_L^-1 for L = N >> (S-1)
            const fp* in = buf_of(S - 1);
            fp* outp = P == 1 ? full : chunk_of(S);
            HIPCHK(launch_fri_fold(in, in + outs, outs, fp_zero(), winv, outp, c->stream, (const fp*)d_c0, g * outs, d_abort));
            if (P > 1) TCHK(d->t.all_gather(d->t.user, outp, full, outs * sizeof(fp), c->stream), "all-gather");
            return SH_OK;
        };
        ChainEnqueuer* helper = nullptr;
        if (!c->enqueuer) { try { c->enqueuer = new ChainEnqueuer(c->device); } catch (const std::system_error&) { c->enqueuer = nullptr; } }
        helper = c->enqueuer;
        if (!helper) return sh_fail(SH_ENOMEM, "sh_dist_fri_commit: cannot start the enqueue thread");
        std::string helper_msg;
        ChainGuard guard{c, &lay, gen};
        guard.waits = S;
        guard.helper = helper;
        guard.armed = true;
        helper->post(enqueue_all);

        const fp inv2 = fp_inv(fp_make(2, 0, 0, 0));
        fp off_inv = fp_inv(off);
        PsCbState st{ps};
        for (uint32_t r = 0; r < S; ++r) {
            const int32_t w = wait_seq(c, reinterpret_cast<const uint32_t*>(c->h_chain + lay.root_seq(r)), gen, r, timeout_s, helper);
            if (w == WAIT_ENQUEUE_FAILED) { const int32_t rc = helper->wait(&helper_msg); return sh_fail(rc, helper_msg.c_str()); }
            if (w == WAIT_RETRY) return sh_failf(SH_EHIP, "sh_dist_fri_commit: round %u: the GPU gave up waiting for its challenge (another rank is more than %.0f s behind?)", r, timeout_s);
            if (w != SH_OK) return w;
            memcpy(roots_out + 64 * (size_t)r, c->h_chain + lay.roots(r), 64);
            uint64_t aw[2];
            if (ps_callback(&st, r, roots_out + 64 * (size_t)r, 1, aw) != 0) return sh_failf(SH_ECALLBACK, "sh_dist_fri_commit: transcript update failed in round %u", r);
            const fp alpha = fp_load_host(aw);
            fp* h_c0 = reinterpret_cast<fp*>(c->c0_host + lay.c0(r));
            h_c0[0] = mont_mul(alpha, fp_to_mont(fp_to_mont(fp_mul(inv2, off_inv))));     // c0 = alpha · 2^-1 · offset_r^-1, Montgomery form
            c0_flush();
            __atomic_store_n(reinterpret_cast<uint32_t*>(c->c0_host + lay.c0_seq(r)), gen, __ATOMIC_RELEASE);
            c0_flush();
            guard.next_c0 = r + 1;
            off_inv = fp_mul(off_inv, off_inv);
        }
        { const int32_t rc = helper->wait(&helper_msg); if (rc != SH_OK) return sh_fail(rc, helper_msg.c_str()); }
        guard.armed = false;
        HIPCHK(hipStreamSynchronize(c->stream));
        if (*h_abort) return sh_fail(SH_EHIP, "sh_dist_fri_commit: a round timed out on the GPU waiting for its challenge");
    } else {
        if (P == 1) HIPCHK(hipMemcpyAsync(full, d_local, m * sizeof(fp), hipMemcpyDeviceToDevice, c->stream));
        else TCHK(d->t.all_gather(d->t.user, d_local, full, m * sizeof(fp), c->stream), "all-gather");
    }
    // ---- the remaining rounds, locally, on the same transcript: codeword S on the coset offset^(2^S)
    sh_fri_params tail = *prm;
    fp off_s = off;
    for (uint32_t i = 0; i < S; ++i) off_s = fp_mul(off_s, off_s);
    fp_store_host(tail.offset, off_s);
    PsCbState st{ps};
    rc_tail = fri_commit_dev(c, full, L_tail, &tail, ps_callback, &st, roots_out + 64 * (size_t)S, last_out, last_len);
    if (rc_tail != SH_OK) return rc_tail;
    return sh_proofstream_push_codeword(ps, last_out, *last_len);
}

extern "C" int32_t sh_dist_fri_commit(sh_dist* d, const uint64_t* d_local, size_t N, const sh_fri_params* params, sh_proofstream* ps, size_t min_half,
                                      uint8_t* roots_out, uint64_t* last_codeword_out, size_t* last_len) {
    SHCHK(dist_check(d));
    if (!d_local || !ps || !roots_out || !last_codeword_out || !last_len) return sh_fail(SH_EINVAL, "sh_dist_fri_commit: null pointer");
    if (!params) return sh_fail(SH_EINVAL, "fri: null params");
    try {
        return dist_fri_commit_impl(d, (const fp*)d_local, N, params, ps, min_half ? min_half : ((size_t)1 << 16), roots_out, last_codeword_out, last_len);
    } catch (const std::bad_alloc&) { return sh_fail(SH_ENOMEM, "sh_dist_fri_commit: out of host memory");
    } catch (...) { return sh_fail(SH_ENOMEM, "sh_dist_fri_commit: unexpected exception on the host side"); }
}
