// api_query.hip — query phase behind include/stark_hip.h: batched Merkle openings, index sampling, sh_fri_prove and the
// host verifier sh_fri_verify.
#include "api_internal.h"

// ------------------------------------------------------------------ query phase: batched openings, prove, verify
// (definitions G9-G12 in DESIGN.md §9 — the reference holds none of this.)
// Paths of `h_idx` (n_idx x depth x 64 B, to the host) in ONE tree, merkle.rs:70-108.  The tree is given by its leaves'
// preimages (d_elems: field elements, leaf = Blake2b512(le16)) or by its leaf digests (d_leaves).  Built like a commit
// (lane-per-subtree, then four-lanes-per-compression launches), keeping the digest array every launch writes — about
// L/2^K·64·1.15 bytes instead of all 2·L·64 — and opened by k_open_paths, which rebuilds the levels in between per path.
static int32_t open_paths_one_tree(sh_ctx* c, const fp* d_elems, const digest64* d_leaves, size_t L, const uint64_t* h_idx, size_t n_idx, uint8_t* h_paths) {
    if (!n_idx) return SH_OK;
    const uint32_t depth = ilog2(L);
    OpenTree T;
    memset(&T, 0, sizeof T);
    T.cw = d_elems;
    T.log_L = depth;
    uint32_t K = 0;
    if (d_elems) { K = leaf_subtree_k(depth); if (K >= depth) K = depth - 1; }
    // sizes first: one allocation
    size_t need = d_elems ? (L >> K) : 0;
    for (uint32_t lg = depth - K;;) { bool q; const uint32_t st = forest_step(lg, 1, &q); if (st >= lg) break; lg -= st; need += (size_t)1 << lg; }
    void* lv = nullptr;
    SHCHK(ensure_scratch(c, SCR_LEVELS, (need ? need : 1) * sizeof(digest64), &lv));
    digest64* next = (digest64*)lv;
    const digest64* cur;
    if (d_elems) {
        HIPCHK(launch_leaf_subtree(d_elems, next, L >> K, K, c->stream));
        cur = next; next += L >> K;
    } else cur = d_leaves;
    T.level[0] = K; T.stored[0] = cur; T.n_stored = 1;
    uint32_t level = K;
    for (uint32_t lg = depth - K;;) {
        bool quad;
        const uint32_t st = forest_step(lg, 1, &quad);
        if (st >= lg) break;                                             // the levels up to the root are the path kernel's last segment
        const size_t n_out = (size_t)1 << (lg - st);
        if (quad) HIPCHK(launch_quad_reduce(cur, next, (size_t)1 << lg, st, nullptr, 0, c->stream));
        else HIPCHK(launch_node_subtree(cur, next, n_out, st, c->stream));
        lg -= st; level += st;
        if (T.n_stored >= OPEN_MAX_STORED) return sh_fail(SH_EHIP, "merkle_open: internal error: too many digest levels");
        T.level[T.n_stored] = level; T.stored[T.n_stored++] = next;
        cur = next; next += n_out;
    }
    T.idx_off = 0; T.n_idx = (uint32_t)n_idx; T.wg_off = 0; T.path_off = 0;
    const size_t n_seg = T.n_stored + ((T.level[0] > 0) ? 1 : 0), n_wg = n_idx * n_seg;
    if (n_wg > 0x7fffffffu) return sh_fail(SH_EINVAL, "merkle_open: too many indices for one launch");
    const size_t off_idx = n_idx * depth * sizeof(digest64), off_tree = (off_idx + n_idx * 8 + 15) & ~(size_t)15, total = off_tree + sizeof(OpenTree);
    void* dq;
    SHCHK(ensure_scratch(c, SCR_PATHS, total, &dq));
    uint8_t* dqb = (uint8_t*)dq;
    HIPCHK(hipMemcpyAsync(dqb + off_idx, h_idx, n_idx * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(dqb + off_tree, &T, sizeof T, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));                             // T lives on this stack frame
    HIPCHK(launch_open_paths(reinterpret_cast<const OpenTree*>(dqb + off_tree), &T, 1, (uint32_t)n_wg, reinterpret_cast<const uint64_t*>(dqb + off_idx),
                             reinterpret_cast<digest64*>(dqb), nullptr, c->stream));
    HIPCHK(hipMemcpyAsync(h_paths, dqb, off_idx, hipMemcpyDeviceToHost, c->stream));
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
    if (number > reduced_size || number > size)          // only min(size, reduced_size) distinct residues exist: the loop below would never end
        return sh_failf(SH_EINVAL, "sample_indices: cannot sample %u indices below %llu with distinct residues mod %llu", number, (unsigned long long)size, (unsigned long long)reduced_size);
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
    if (!d_cw) return sh_fail(SH_EINVAL, "fri_prove: null codeword");
    if (!is_pow2(N) || ilog2(N) > SH_MAX_LOG_N) return sh_failf(SH_EINVAL, "fri_prove: N must be a power of two <= 2^%d", SH_MAX_LOG_N);   // before anything is sized from N
    const uint32_t t = prm->num_colinearity_tests;
    const uint32_t R = sh_fri_num_rounds(N, prm->expansion, t);
    std::vector<uint8_t> roots((size_t)(R ? R : 1) * 64);
    std::vector<uint64_t> last(2 * (N >> (R ? R - 1 : 0)));          // the last codeword only (N/2^(R-1) elements)
    size_t last_len = 0;
    CommitKeep kept;
    PsCbState st{ps};
    SHCHK(fri_commit_dev(c, d_cw, N, prm, ps_callback, &st, roots.data(), last.data(), &last_len, &kept));
    SHCHK(sh_proofstream_push_codeword(ps, last.data(), last_len));
    if (R < 2) return SH_OK;                                   // a single codeword: nothing to test against
    uint8_t seed[32];
    SHCHK(sh_proofstream_prover_fiat_shamir(ps, seed));
    std::vector<uint64_t> top;
    SHCHK(sample_indices(seed, N / 2, last_len, t, &top));
    if (top_out) memcpy(top_out, top.data(), top.size() * 8);
    *n_top = top.size();

    // Per tree r: openings wanted = a_r, b_r (tests of round r, r < R-1) and c_(r-1) (tests of round r-1, r >= 1).
    // All trees go through ONE launch (k_open_paths: paths from the digest arrays the commit left behind, opened values
    // alongside), one copy in and one copy out.
    std::vector<OpenTree> trees(R);
    size_t n_idx_total = 0, n_path = 0, n_wg = 0;
    for (uint32_t r = 0; r < R; ++r) {
        const uint32_t depth = ilog2(N >> r);
        const size_t n_idx = (size_t)t * ((r + 1 < R ? 2 : 0) + (r >= 1 ? 1 : 0));
        OpenTree& T = trees[r];
        memset(&T, 0, sizeof T);
        T.cw = kept.cws[r];
        T.log_L = depth;
        const auto& stored = kept.trees[r].stored;
        if (stored.size() > OPEN_MAX_STORED) return sh_fail(SH_EHIP, "fri_prove: internal error: more digest levels kept than the path kernel takes");
        for (const auto& lv : stored) if (lv.first < depth) { T.level[T.n_stored] = lv.first; T.stored[T.n_stored++] = lv.second; }
        T.idx_off = (uint32_t)n_idx_total; T.n_idx = (uint32_t)n_idx;
        T.wg_off = (uint32_t)n_wg; T.path_off = n_path;
        n_idx_total += n_idx; n_path += n_idx * depth; n_wg += n_idx * (T.n_stored + ((T.n_stored == 0 || T.level[0] > 0) ? 1 : 0));
    }
    if (n_wg > 0x7fffffffu) return sh_fail(SH_EINVAL, "fri_prove: too many openings for one launch");
    const size_t off_vals = n_path * sizeof(digest64), off_idx = off_vals + n_idx_total * sizeof(fp);
    const size_t off_trees = (off_idx + n_idx_total * 8 + 15) & ~(size_t)15, total = off_trees + R * sizeof(OpenTree);
    void* dq;
    SHCHK(ensure_scratch(c, SCR_PATHS, total, &dq));
    if (c->h_query_bytes < total) {
        if (c->h_query) { HIPCHK(hipHostFree(c->h_query)); c->h_query = nullptr; c->h_query_bytes = 0; }
        HIPCHK(hipHostMalloc((void**)&c->h_query, total, hipHostMallocDefault));
        c->h_query_bytes = total;
    }
    uint8_t* hq = c->h_query;
    uint64_t* h_idx = reinterpret_cast<uint64_t*>(hq + off_idx);
    for (uint32_t r = 0; r < R; ++r) {
        const size_t L = N >> r;
        uint64_t* ix = h_idx + trees[r].idx_off;
        if (r + 1 < R) {
            for (uint32_t s_ = 0; s_ < t; ++s_) *ix++ = top[s_] % (L / 2);
            for (uint32_t s_ = 0; s_ < t; ++s_) *ix++ = top[s_] % (L / 2) + L / 2;
        }
        if (r >= 1) for (uint32_t s_ = 0; s_ < t; ++s_) *ix++ = top[s_] % L;                  // c of round r-1: top % (L_(r-1)/2)
    }
    memcpy(hq + off_trees, trees.data(), R * sizeof(OpenTree));
    uint8_t* dqb = (uint8_t*)dq;
    HIPCHK(hipMemcpyAsync(dqb + off_idx, hq + off_idx, total - off_idx, hipMemcpyHostToDevice, c->stream));
    HIPCHK(launch_open_paths(reinterpret_cast<const OpenTree*>(dqb + off_trees), trees.data(), R, (uint32_t)n_wg, reinterpret_cast<const uint64_t*>(dqb + off_idx),
                             reinterpret_cast<digest64*>(dqb), reinterpret_cast<fp*>(dqb + off_vals), c->stream));
    if (off_idx) HIPCHK(hipMemcpyAsync(hq, dqb, off_idx, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    const uint64_t* h_vals = reinterpret_cast<const uint64_t*>(hq + off_vals);
    for (uint32_t r = 0; r + 1 < R; ++r) {                                  // G11: per round, t value triples, then 3t paths
        const uint32_t d0 = ilog2(N >> r), d1 = d0 - 1;
        const size_t c_off = (r + 2 < R) ? 2 * (size_t)t : 0;               // where tree r+1 keeps its c entries
        const uint64_t *v0 = h_vals + 2 * (size_t)trees[r].idx_off, *v1 = h_vals + 2 * (size_t)trees[r + 1].idx_off;
        const uint8_t *p0 = hq + trees[r].path_off * 64, *p1 = hq + trees[r + 1].path_off * 64;
        for (uint32_t s_ = 0; s_ < t; ++s_) {
            uint64_t abc[6] = {v0[2 * s_], v0[2 * s_ + 1], v0[2 * (t + s_)], v0[2 * (t + s_) + 1], v1[2 * (c_off + s_)], v1[2 * (c_off + s_) + 1]};
            SHCHK(sh_proofstream_push_leaves(ps, abc, 3));
        }
        for (uint32_t s_ = 0; s_ < t; ++s_) {
            SHCHK(sh_proofstream_push_path(ps, p0 + (size_t)s_ * d0 * 64, d0));
            SHCHK(sh_proofstream_push_path(ps, p0 + (size_t)(t + s_) * d0 * 64, d0));
            SHCHK(sh_proofstream_push_path(ps, p1 + (c_off + s_) * d1 * 64, d1));
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
    return open_paths_one_tree(c, (const fp*)d_elems, nullptr, L, indices, n_idx, paths);
}
int32_t sh_merkle_open_batch(sh_ctx* c, const uint8_t* leaves, size_t L, const uint64_t* indices, size_t n_idx, uint8_t* paths) {
    CTXCHK(c);
    if (!leaves || (!indices && n_idx) || (!paths && n_idx)) return sh_fail(SH_EINVAL, "sh_merkle_open_batch: null pointer");
    SHCHK(check_open_args("sh_merkle_open_batch", L, indices, n_idx));
    void* dl;
    SHCHK(ensure_scratch(c, SCR_IO_C, L * sizeof(digest64), &dl));
    HIPCHK(hipMemcpyAsync(dl, leaves, L * 64, hipMemcpyHostToDevice, c->stream));
    return open_paths_one_tree(c, nullptr, (const digest64*)dl, L, indices, n_idx, paths);
}
int32_t sh_fri_sample_indices(const uint8_t seed[32], uint64_t size, uint64_t reduced_size, uint32_t number, uint64_t* out) {
    try {
    if (!seed || (!out && number)) return sh_fail(SH_EINVAL, "sh_fri_sample_indices: null pointer");
    std::vector<uint64_t> v;
    SHCHK(sample_indices(seed, size, reduced_size, number, &v));
    memcpy(out, v.data(), v.size() * 8);
    return SH_OK;
    } catch (const std::bad_alloc&) { return sh_fail(SH_ENOMEM, "sh_fri_sample_indices: out of host memory"); } catch (...) { return sh_fail(SH_ENOMEM, "sh_fri_sample_indices: unexpected exception on the host side"); }
}
// nothing unwinds through the C ABI: host-side allocation failures come back as SH_ENOMEM
static int32_t fri_prove_guarded(sh_ctx* c, const fp* d_cw, size_t N, const sh_fri_params* params, sh_proofstream* ps, uint64_t* top, size_t* n_top) {
    try { return fri_prove_dev(c, d_cw, N, params, ps, top, n_top); }
    catch (const std::bad_alloc&) { return sh_fail(SH_ENOMEM, "fri_prove: out of host memory"); }
    catch (...) { return sh_fail(SH_ENOMEM, "fri_prove: unexpected exception on the host side"); }
}
int32_t sh_dev_fri_prove(sh_ctx* c, const uint64_t* d_codeword, size_t N, const sh_fri_params* params, sh_proofstream* ps,
                         uint64_t* top_level_indices, size_t* n_indices) {
    return fri_prove_guarded(c, (const fp*)d_codeword, N, params, ps, top_level_indices, n_indices);
}
const char* sh_fri_proof_format(void) { return "stark-anat-fri/1"; }
int32_t sh_fri_prove(sh_ctx* c, const uint64_t* codeword, size_t N, const sh_fri_params* params, sh_proofstream* ps,
                     uint64_t* top_level_indices, size_t* n_indices) {
    CTXCHK(c);
    if (!codeword) return sh_fail(SH_EINVAL, "sh_fri_prove: null pointer");
    if (!is_pow2(N) || ilog2(N) > SH_MAX_LOG_N) return sh_failf(SH_EINVAL, "sh_fri_prove: N must be a power of two <= 2^%d", SH_MAX_LOG_N);
    void* d;
    SHCHK(ensure_scratch(c, SCR_IO_C, N * 16, &d));
    SHCHK(upload_pipelined(c, codeword, N, (fp*)d, nullptr));
    return fri_prove_guarded(c, (const fp*)d, N, params, ps, top_level_indices, n_indices);
}

// G12: the verifier.  Host only ("CPU-side is enough"): ~R·3t path checks of log2 L compressions.
// Reads the stream from its read index.  *accepted = 1/0; on 0, sh_last_error() says which check failed.
int32_t sh_fri_verify(sh_proofstream* ps, size_t N, const sh_fri_params* prm, int32_t* accepted, uint64_t* top_level_indices, size_t* n_indices) {
    try {
    if (!ps || !accepted) return sh_fail(SH_EINVAL, "sh_fri_verify: null pointer");
    *accepted = 0;
    if (n_indices) *n_indices = 0;
    if (!is_pow2(N) || ilog2(N) > 119) return sh_fail(SH_EINVAL, "sh_fri_verify: N must be a power of two <= 2^119");
    fp off;
    SHCHK(check_params(prm, &off));
    const uint32_t t = prm->num_colinearity_tests, log_N = ilog2(N);
    const uint32_t R = sh_fri_num_rounds(N, prm->expansion, t);
    auto reject = [&](const char* why) { sh_fail(SH_OK, why); return SH_OK; };
    const bool dbg = getenv("STARK_HIP_DEBUG") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_field = 0, t_parse = 0, t_hash = 0, t_head = 0;
    const double t_begin = now();
    struct Report { bool on; double *f, *p, *h, *hd; double t0; std::function<double()> clk;
                    ~Report() { if (on) fprintf(stderr, "[stark_hip] verify: %.0f us total: roots/last codeword/low degree %.0f, colinearity (field) %.0f, path parsing %.0f, path hashing %.0f\n", clk() - t0, *hd, *f, *p, *h); } }
        report{dbg, &t_field, &t_parse, &t_hash, &t_head, t_begin, now};
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
    if (!host_ps_pull_elems(ps, "Codeword", &lastw)) return reject("verify: expected the last Codeword item");
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
    t_head = now() - t_begin;
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
        const double tf0 = now();
        // 1/ax for the t test points of the round with ONE field inversion (Montgomery's trick; ax = off_r·w_r^a is never zero)
        std::vector<fp> axs(t), pre(t);
        fp run = fp_one();
        for (uint32_t s_ = 0; s_ < t; ++s_) {
            axs[s_] = fp_mul(off_r, fp_pow(w_r, top[s_] % (L / 2)));
            pre[s_] = run;
            run = fp_mul(run, axs[s_]);
        }
        fp inv_run = fp_inv(run);
        for (uint32_t s_ = t; s_-- > 0;) { const fp ai = fp_mul(inv_run, pre[s_]); inv_run = fp_mul(inv_run, axs[s_]); axs[s_] = ai; }   // axs now holds the inverses
        for (uint32_t s_ = 0; s_ < t; ++s_) {
            if (!host_ps_pull_elems(ps, "Leaves", &abc[s_]) || abc[s_].size() != 6) return reject("verify: expected a Leaves item with three values");
            const fp ay = fp_from_u64x2(abc[s_][0], abc[s_][1]), by = fp_from_u64x2(abc[s_][2], abc[s_][3]), cy = fp_from_u64x2(abc[s_][4], abc[s_][5]);
            if (!fp_is_canonical(ay) || !fp_is_canonical(by) || !fp_is_canonical(cy)) return reject("verify: non-canonical value in a Leaves item");
            // colinearity of (ax, ay), (-ax, by), (alpha, cy), ax = off_r·w_r^a  <=>  cy = fold(ay, by) (G4)
            const fp q = fp_mul(alphas[r], axs[s_]);
            const fp want = fp_mul(inv2, fp_add(fp_mul(fp_add(fp_one(), q), ay), fp_mul(fp_sub(fp_one(), q), by)));
            if (!fp_eq(want, cy)) return reject("verify: colinearity check failed");
        }
        t_field += now() - tf0;
        for (uint32_t s_ = 0; s_ < t; ++s_) {
            const uint64_t a = top[s_] % (L / 2);
            const uint64_t where[3] = {a, a + L / 2, a};
            for (int k = 0; k < 3; ++k) {
                std::vector<uint8_t> path;
                const uint32_t depth = k < 2 ? d0 : d1;
                const double tp0 = now();
                if (!host_ps_pull_path(ps, &path) || path.size() != (size_t)depth * 64) return reject("verify: expected a Path item of the tree's depth");
                const double tp1 = now();
                t_parse += tp1 - tp0;
                uint8_t le[16], leaf[64];
                memcpy(le, &abc[s_][2 * k], 8); memcpy(le + 8, &abc[s_][2 * k + 1], 8);
                host_blake2b512(le, 16, leaf);
                int32_t ok = 0;
                SHCHK(sh_merkle_verify(roots.data() + 64 * (size_t)(r + (k == 2 ? 1 : 0)), leaf, path.data(), depth, where[k], &ok));
                if (!ok) return reject("verify: Merkle authentication path check failed");
                t_hash += now() - tp1;
            }
        }
        off_r = fp_mul(off_r, off_r); w_r = fp_mul(w_r, w_r);
    }
    *accepted = 1;
    return SH_OK;
    } catch (const std::bad_alloc&) { return sh_fail(SH_ENOMEM, "sh_fri_verify: out of host memory"); } catch (...) { return sh_fail(SH_ENOMEM, "sh_fri_verify: unexpected exception on the host side"); }
}

}  // extern "C"
