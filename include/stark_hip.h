/*
 * stark_hip.h — C ABI of libstark_hip.so: the MI355X (gfx950) implementation of the FRI-commit hot
 * path of the `stark-anat` crate (greged93/stark-anatomy).  This is the drop-in boundary: every
 * entry point below is what a Rust `extern "C"` block in the reference crate would bind (see
 * INTEGRATION.md for the binding), and cites the reference item (file:line under
 * /root/reference/src) it replaces.  The reference has no FFI of its own: the
 * boundary is its public Rust API, restated here over plain pointers and sizes.
 *
 * Conventions
 *  - Field element = canonical value in [0,p), p = 1 + 407*2^119 (field.rs:15), as little-endian
 *    u128 = uint64_t[2] {lo, hi} (== Rust `u128::to_le_bytes`, what `FieldElement::value()`
 *    returns, field.rs:52-54).  Arrays are contiguous, 16 B per element.
 *  - Hash = 64 bytes (merkle.rs:5 `Hash = [u8; 64]`, `MerkleHash::as_bytes`, merkle.rs:22-24).
 *  - Every function returns int32_t status: SH_OK or a negative SH_E* code; sh_last_error()
 *    gives a thread-local message.  Nothing throws or unwinds across this boundary.  Contract
 *    violations that `panic!`/`assert!` in the reference (non-power-of-two leaf count
 *    merkle.rs:36, index out of range merkle.rs:75, division by zero field.rs:101-103) come back
 *    as SH_EINVAL / SH_EBOUNDS / SH_EDIVZERO so the Rust shim can re-raise the same panic.
 *  - `sh_*`      : caller-owned HOST buffers; data is copied to the GPU, computed there, copied back.
 *    `sh_dev_*`  : caller-owned DEVICE buffers (hipMalloc / sh_dev_alloc / torch tensor data_ptr) on
 *                  the context's device; work is enqueued on the context's stream; functions that
 *                  return host results (roots, status flags) synchronise that stream.
 *  - There is NO CPU fallback: every compute entry point fails with SH_EHIP when no gfx950 device
 *    is usable.  Host-only helpers are listed separately at the end.
 *  - Threading: calls on one sh_ctx must be serialised by the caller; distinct contexts (one per
 *    GPU / thread) are independent (the reference is single-threaded plain data).
 */
#ifndef STARK_HIP_H
#define STARK_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SH_OK        0
#define SH_EINVAL   (-1)  /* null pointer, zero / non-power-of-two length, unsupported size   */
#define SH_EDIVZERO (-2)  /* field.rs:101-103 "Cannot divide by zero"                        */
#define SH_EHIP     (-3)  /* HIP runtime error or no usable gfx950 device                     */
#define SH_ENOMEM   (-4)
#define SH_ECALLBACK (-5) /* transcript callback returned non-zero                            */
#define SH_EBOUNDS  (-6)  /* index out of range (merkle.rs:75; proof_stream.rs:30-32)         */
#define SH_EPARSE   (-7)  /* proof_stream.rs:43 serde_json error on deserialize               */

typedef struct sh_ctx sh_ctx;

const char* sh_last_error(void);
const char* sh_version(void);

/* ---------------- context / device memory ---------------- */
/* `hip_stream` may be NULL (the context creates its own stream) or a hipStream_t the caller owns
 * (e.g. torch.cuda.current_stream().cuda_stream) on `device`. */
int32_t sh_ctx_create(int32_t device, void* hip_stream, sh_ctx** out);
int32_t sh_ctx_destroy(sh_ctx* ctx);
/* Diagnostics: rounds that the FRI commit driver had to enqueue a second time on this context so far (see the commit driver
 * below); 0 in normal operation. */
uint64_t sh_ctx_chain_retries(const sh_ctx* ctx);
int32_t sh_ctx_sync(sh_ctx* ctx);
int32_t sh_ctx_device(sh_ctx* ctx, int32_t* device);
int32_t sh_dev_alloc(sh_ctx* ctx, size_t bytes, void** dptr);
int32_t sh_dev_free(sh_ctx* ctx, void* dptr);
int32_t sh_dev_upload(sh_ctx* ctx, void* dptr, const void* host, size_t bytes);
int32_t sh_dev_download(sh_ctx* ctx, void* host, const void* dptr, size_t bytes);

/* ---------------- field, batched (field.rs) ----------------
 * out[i] = a[i] (+,-,*) b[i]  — `impl Add/Sub/Mul for FieldElement`, field.rs:57-95.
 * sh_fp_div: out[i] = a[i] / b[i] — field.rs:97-114; SH_EDIVZERO if any b[i] == 0 (field.rs:101-103: out[i] = 0 at those
 * positions, every other quotient is still written).  sh_fp_inv: out[i] = 1 / a[i], same rule.  Division and inversion
 * are batched by Montgomery's trick: one field inversion per 2048 elements, 3-4 multiplies per element.
 * Inputs must be canonical (as `FieldElement::new` guarantees, field.rs:30-35); sh_fp_new reduces arbitrary u128 values first. */
int32_t sh_fp_new(sh_ctx* ctx, const uint64_t* x, uint64_t* out, size_t n);
int32_t sh_fp_add(sh_ctx* ctx, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int32_t sh_fp_sub(sh_ctx* ctx, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int32_t sh_fp_mul(sh_ctx* ctx, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int32_t sh_fp_div(sh_ctx* ctx, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int32_t sh_fp_inv(sh_ctx* ctx, const uint64_t* a, uint64_t* out, size_t n);
int32_t sh_dev_fp_add(sh_ctx* ctx, const uint64_t* d_a, const uint64_t* d_b, uint64_t* d_out, size_t n);
int32_t sh_dev_fp_sub(sh_ctx* ctx, const uint64_t* d_a, const uint64_t* d_b, uint64_t* d_out, size_t n);
int32_t sh_dev_fp_mul(sh_ctx* ctx, const uint64_t* d_a, const uint64_t* d_b, uint64_t* d_out, size_t n);
int32_t sh_dev_fp_div(sh_ctx* ctx, const uint64_t* d_a, const uint64_t* d_b, uint64_t* d_out, size_t n);
int32_t sh_dev_fp_inv(sh_ctx* ctx, const uint64_t* d_a, uint64_t* d_out, size_t n);

/* ---------------- NTT / coset LDE (not in the reference) ----------------
 * In place, natural order in and out: X[k] = sum_j x[j]·w^(jk), w = g^(2^119 / n),
 * g = 85408008396924667383611388730472331217; inverse uses w^-1 and scales by n^-1.
 * Built only from field.rs Add/Sub/Mul semantics.  1 <= 2^log_n, log_n <= SH_MAX_LOG_N. */
#define SH_MAX_LOG_N 28
int32_t sh_ntt(sh_ctx* ctx, uint64_t* data, uint32_t log_n, int32_t inverse);
int32_t sh_dev_ntt(sh_ctx* ctx, uint64_t* d_data, uint32_t log_n, int32_t inverse);
/* `batch` independent transforms of 2^log_n elements each, stored back to back (batch·2^log_n <= 2^(SH_MAX_LOG_N+2)).
 * Together with sh_*_twiddle_scale these are the local steps of the multi-GPU four-step NTT
 * (BASELINE.json C5; host side in stark_anatomy_amd/dist_ntt.py); no reference counterpart. */
int32_t sh_ntt_batch(sh_ctx* ctx, uint64_t* data, size_t batch, uint32_t log_n, int32_t inverse);
int32_t sh_dev_ntt_batch(sh_ctx* ctx, uint64_t* d_data, size_t batch, uint32_t log_n, int32_t inverse);
/* data[r][c] *= w^((row0+r)·(col0+c)) over a rows x 2^log_cols row-major block, w = g^(2^119 / 2^log_n)
 * (w^-1 when inverse): the twiddle between the column and the row transforms of a four-step NTT of 2^log_n points. */
int32_t sh_twiddle_scale(sh_ctx* ctx, uint64_t* data, size_t rows, uint32_t log_cols, uint64_t row0, uint64_t col0, uint32_t log_n, int32_t inverse);
int32_t sh_dev_twiddle_scale(sh_ctx* ctx, uint64_t* d_data, size_t rows, uint32_t log_cols, uint64_t row0, uint64_t col0, uint32_t log_n, int32_t inverse);
/* out[i] = f(offset·w_N^i), i in [0, n << log_expansion), f = sum_j coeffs[j] X^j. */
int32_t sh_coset_lde(sh_ctx* ctx, const uint64_t* coeffs, uint32_t log_n, uint32_t log_expansion,
                     const uint64_t offset[2], uint64_t* out);
int32_t sh_dev_coset_lde(sh_ctx* ctx, const uint64_t* d_coeffs, uint32_t log_n, uint32_t log_expansion,
                         const uint64_t offset[2], uint64_t* d_out);
/* `batch` (<= 65535) polynomials of 2^log_n coefficients stored back to back -> their extensions back to back, one launch per NTT
 * pass for all of them: the front end of a batch of commits (BASELINE.json C4: 1024 codewords of 2^14).  No reference counterpart.
 * Aliasing: d_coeffs == d_out (in place) is allowed for batch == 1 only; any other overlap of the two ranges is SH_EINVAL. */
int32_t sh_dev_coset_lde_batch(sh_ctx* ctx, const uint64_t* d_coeffs, size_t batch, uint32_t log_n, uint32_t log_expansion,
                               const uint64_t offset[2], uint64_t* d_out);

/* ---------------- polynomial layer ----------------
 * NO COUNTERPART IN THE REFERENCE CRATE (it has no polynomial type): these are the operations a STARK prover
 * needs next to the FRI commit, defined as in DESIGN.md §10 and pinned by the oracle's schoolbook restatement.
 * A polynomial is its coefficient array, lowest degree first, at least one coefficient, no trimming.
 * sh_poly_mul: out[0 .. na+nb-1) = a · b, through the NTT (na + nb - 1 <= 2^SH_MAX_LOG_N). */
int32_t sh_poly_mul(sh_ctx* ctx, const uint64_t* a, size_t na, const uint64_t* b, size_t nb, uint64_t* out);
int32_t sh_dev_poly_mul(sh_ctx* ctx, const uint64_t* d_a, size_t na, const uint64_t* d_b, size_t nb, uint64_t* d_out);
/* Quotient by the zerofier Z(X) = X^n - c, n = 2^log_n, of evaluations on the coset {offset·w_N^i}, N = 2^log_N >= n:
 * out[i] = cw[i] / Z(offset·w_N^i).  Z takes only N/n distinct values there: they are inverted in one batch.
 * SH_EDIVZERO if the coset meets a root of Z (field.rs:101-103). */
int32_t sh_coset_zerofier_div(sh_ctx* ctx, const uint64_t* cw, uint32_t log_N, uint32_t log_n, const uint64_t offset[2],
                              const uint64_t c[2], uint64_t* out);
int32_t sh_dev_coset_zerofier_div(sh_ctx* ctx, const uint64_t* d_cw, uint32_t log_N, uint32_t log_n, const uint64_t offset[2],
                                  const uint64_t c[2], uint64_t* d_out);
/* Coefficient form: f = q · (X^n - c) + r with deg r < n, any n >= 1.  q gets nf - n coefficients (none, and q may be NULL,
 * if nf <= n), r gets n (zero-padded).  Parallel over the n residue classes of the exponent, nf/n steps deep. */
int32_t sh_poly_divmod_zerofier(sh_ctx* ctx, const uint64_t* f, size_t nf, size_t n, const uint64_t c[2], uint64_t* q, uint64_t* r);
int32_t sh_dev_poly_divmod_zerofier(sh_ctx* ctx, const uint64_t* d_f, size_t nf, size_t n, const uint64_t c[2], uint64_t* d_q, uint64_t* d_r);

/* ---------------- Merkle (merkle.rs) ----------------
 * sh_merkle_commit == `MerkleTree::commit(&[MerkleHash]).root`, merkle.rs:35-68: Blake2b-512 over
 * adjacent pairs H(left||right), bottom-up; L must be a power of two (else SH_EINVAL, merkle.rs:36);
 * L == 1 -> root == leaf (merkle.rs:48,62). `leaves` = L×64 bytes of already-hashed leaves. */
int32_t sh_merkle_commit(sh_ctx* ctx, const uint8_t* leaves, size_t L, uint8_t root[64]);
int32_t sh_dev_merkle_commit(sh_ctx* ctx, const uint8_t* d_leaves, size_t L, uint8_t root[64]);
/* leaf hash of a field element: out[i] = Blake2b512(value_i.to_le_bytes()). */
int32_t sh_hash_leaves(sh_ctx* ctx, const uint64_t* elems, size_t L, uint8_t* out);
int32_t sh_dev_hash_leaves(sh_ctx* ctx, const uint64_t* d_elems, size_t L, uint8_t* d_out);
/* fused: root of the tree whose leaves are sh_hash_leaves(elems) (what one FRI round commits to). */
int32_t sh_merkle_commit_elems(sh_ctx* ctx, const uint64_t* elems, size_t L, uint8_t root[64]);
int32_t sh_dev_merkle_commit_elems(sh_ctx* ctx, const uint64_t* d_elems, size_t L, uint8_t root[64]);
/* d_out[i] = Merkle root of the 2^K-leaf subtree over leafhash(d_elems[i·2^K .. (i+1)·2^K)), 0 <= K <= 4: the
 * bottom K+1 levels of merkle.rs:48-60 for n_out independent subtrees in one launch (one subtree per lane).
 * This is the dominant kernel of the commit; also the building block for batches of small commits. */
int32_t sh_dev_merkle_subtrees(sh_ctx* ctx, const uint64_t* d_elems, size_t n_out, uint32_t K, uint8_t* d_out);
/* Allocation-free variant of sh_dev_merkle_commit_elems.  NO COUNTERPART IN THE REFERENCE: it exists for callers that
 * cannot allocate device memory — tight serving loops, harnesses that forbid hipMalloc.  It allocates nothing, but it still waits for the root on
 * the host (mailbox poll, or a copy and a stream synchronise), so it cannot run under stream capture.
 * Same kernels and the same root as sh_dev_merkle_commit_elems, but the digest buffers live in a fixed 8 MiB + 33 KiB
 * __device__ array of the library (one copy per GPU, held until the process ends) instead of the context's growable
 * scratch; it calls no hipMalloc / hipFree.  L must be a power of two <= 2^SH_NOALLOC_MAX_LOG_L (else SH_EINVAL).
 * Calls on the same GPU share that array: calls that overlap on one GPU (two contexts / streams) race. */
#define SH_NOALLOC_MAX_LOG_L 20
int32_t sh_dev_merkle_commit_elems_noalloc(sh_ctx* ctx, const uint64_t* d_elems, size_t L, uint8_t root[64]);
/* `MerkleTree::open(index)`, merkle.rs:70-108: path[0] = sibling leaf … path[log2 L − 1] = sibling
 * of the root's child; *path_len = log2 L.  SH_EBOUNDS if index >= L or L < 2 (merkle.rs:75,104). */
int32_t sh_merkle_open(sh_ctx* ctx, const uint8_t* leaves, size_t L, size_t index, uint8_t* path, size_t* path_len);

/* ---------------- FRI split-and-fold ----------------
 * out[i] = 2^-1·((1 + alpha/(offset·omega^i))·cw[i] + (1 − alpha/(offset·omega^i))·cw[L/2+i]),
 * i in [0, L/2).  The formula is evaluated as written for ANY non-zero canonical omega (not checked against w_L: a FRI round
 * passes the primitive L-th root w_L, the multi-GPU path passes the same root with a slice offset); SH_EDIVZERO if offset or
 * omega is zero (field.rs:101-103), SH_EINVAL if a scalar is not canonical. */
int32_t sh_fri_fold(sh_ctx* ctx, const uint64_t* cw, size_t L, const uint64_t alpha[2],
                    const uint64_t offset[2], const uint64_t omega[2], uint64_t* out);
int32_t sh_dev_fri_fold(sh_ctx* ctx, const uint64_t* d_cw, size_t L, const uint64_t alpha[2],
                        const uint64_t offset[2], const uint64_t omega[2], uint64_t* d_out);
/* The same fold for the slice i in [i0, i0+count) of the folded codeword: d_lo[k] = cw[i0+k], d_hi[k] = cw[L/2+i0+k].
 * One rank's share of a round when a single codeword is spread over GPUs (dist_ntt.py). */
int32_t sh_dev_fri_fold_range(sh_ctx* ctx, const uint64_t* d_lo, const uint64_t* d_hi, size_t count, uint64_t i0,
                              const uint64_t alpha[2], const uint64_t offset[2], const uint64_t omega[2], uint64_t* d_out);
/* d_data[i] *= base^(j0+i), i < count: the coset pre-scale (G7) of the slice [j0, j0+count) of a coefficient vector. */
int32_t sh_dev_scale_powers(sh_ctx* ctx, uint64_t* d_data, size_t count, uint64_t j0, const uint64_t base[2]);

/* ---------------- device-side pieces of the multi-GPU path (no counterpart in the reference) ----------------
 * None of these synchronises or reads host memory except sh_dev_merkle_commit_pairs, which returns the round's root.
 * sh_dev_transpose: d_out[c][r] = d_in[r][c], rows x cols elements — the pack of an all-to-all exchange (cols = P·b:
 * d_out is then [dest s][col][row], contiguous per destination).  sh_dev_interleave_blocks: d_out[c][s][r] =
 * d_in[s][c][r], (P, b, a) -> (b, P, a) — the unpack after it.  In and out must not alias. */
int32_t sh_dev_zero(sh_ctx* ctx, void* dptr, size_t bytes);                          /* hipMemsetAsync on the context's stream */
int32_t sh_dev_transpose(sh_ctx* ctx, const uint64_t* d_in, size_t rows, size_t cols, uint64_t* d_out);
int32_t sh_dev_interleave_blocks(sh_ctx* ctx, const uint64_t* d_in, size_t P, size_t b, size_t a, uint64_t* d_out);
/* Merkle root over leafhash(d_elems[0..L)) written to 64 bytes of DEVICE memory at d_root (a rank's sub-root of a sharded round). */
int32_t sh_dev_merkle_root_elems(sh_ctx* ctx, const uint64_t* d_elems, size_t L, uint8_t* d_root);
/* Root of the 2P-leaf tree over the all-gathered sub-roots of a sharded round: d_pairs = P x 2 digests, rank s's roots over
 * half-slices s and P + s of the codeword; leaves in half-slice order.  P a power of two. */
int32_t sh_dev_merkle_commit_pairs(sh_ctx* ctx, const uint8_t* d_pairs, size_t P, uint8_t root[64]);

/* ---------------- FRI commit driver ----------------
 * for r in 0..R: root_r = merkle(leafhash(cw_r)); cb(root_r) -> alpha_r; cw_{r+1} = fold(cw_r).
 * R = sh_fri_num_rounds(N, expansion, num_colinearity_tests); the last round's codeword (length
 * N >> (R-1)) is returned in the clear.  The transcript stays on the host: the callback receives
 * every root in order and (when want_alpha != 0) must return the challenge, so the caller's
 * `ProofStream` (proof_stream.rs:25-27, 51-56) remains authoritative and byte-identical.
 * While a callback runs, the next round's kernel is already on the GPU waiting for the challenge it returns.  That wait is a
 * bounded spin (STARK_HIP_CHAIN_SPIN_US, default 500): a kernel that runs out abandons its round, the kernels behind it return
 * at once, and the driver enqueues the round again when the challenge is there — so a slow callback, or one that waits for
 * this GPU, costs at most one spin budget plus a re-launch per round, not the commit.  sh_ctx_chain_retries() counts these.
 * A root that does not arrive within STARK_HIP_CHAIN_TIMEOUT_S (default 20 s) fails the commit with SH_EHIP. */
typedef struct {
    uint32_t expansion;                 /* 4 in every reference configuration            */
    uint32_t num_colinearity_tests;     /* 17                                            */
    uint64_t offset[2];                 /* coset offset, canonical; g by convention      */
} sh_fri_params;
typedef int32_t (*sh_transcript_cb)(void* user, uint32_t round, const uint8_t root[64],
                                    int32_t want_alpha, uint64_t alpha_out[2]);
uint32_t sh_fri_num_rounds(size_t N, uint32_t expansion, uint32_t num_colinearity_tests);
int32_t sh_fri_commit(sh_ctx* ctx, const uint64_t* codeword, size_t N, const sh_fri_params* params,
                      sh_transcript_cb cb, void* user, uint8_t* roots_out /* R×64 */,
                      uint64_t* last_codeword_out /* (N >> (R-1)) elems */, size_t* last_len);
/* device-resident codeword; `d_codeword` is not modified. */
int32_t sh_dev_fri_commit(sh_ctx* ctx, const uint64_t* d_codeword, size_t N, const sh_fri_params* params,
                          sh_transcript_cb cb, void* user, uint8_t* roots_out,
                          uint64_t* last_codeword_out, size_t* last_len);

/* ---------------- ProofStream (proof_stream.rs) — host side, tiny, byte-exact ----------------
 * Items are held as their compact serde_json fragments (what `serialize` observes,
 * proof_stream.rs:38-40, 67-69).  The FRI item type is
 *   enum ProofItem { MerkleRoot(MerkleHash), Codeword(Vec<u128>) }   (serde externally tagged). */
typedef struct sh_proofstream sh_proofstream;
int32_t sh_proofstream_new(sh_proofstream** out);                                      /* Default, proof_stream.rs:17 */
int32_t sh_proofstream_free(sh_proofstream* ps);
int32_t sh_proofstream_push_json(sh_proofstream* ps, const uint8_t* json, size_t len); /* push, :25-27 (any S) */
int32_t sh_proofstream_push_merkle_root(sh_proofstream* ps, const uint8_t root[64]);
int32_t sh_proofstream_push_codeword(sh_proofstream* ps, const uint64_t* elems, size_t n);
/* query-phase items (G9): {"Leaves":[dec,...]} and {"Path":[{"inner":[..]},...]} (Vec<MerkleHash>, merkle.rs:9-13) */
int32_t sh_proofstream_push_leaves(sh_proofstream* ps, const uint64_t* elems, size_t n);
int32_t sh_proofstream_push_path(sh_proofstream* ps, const uint8_t* digests, size_t n);
int32_t sh_proofstream_len(sh_proofstream* ps, size_t* n_items, size_t* read_index);
int32_t sh_proofstream_pull(sh_proofstream* ps, const uint8_t** json, size_t* len);    /* pull, :29-36; SH_EBOUNDS */
int32_t sh_proofstream_serialize(sh_proofstream* ps, uint8_t* out, size_t cap, size_t* needed); /* :38-40 */
int32_t sh_proofstream_deserialize(const uint8_t* bytes, size_t len, sh_proofstream** out);     /* :42-49; SH_EPARSE */
int32_t sh_proofstream_prover_fiat_shamir(sh_proofstream* ps, uint8_t out[32]);        /* :51-56 */
int32_t sh_proofstream_verifier_fiat_shamir(sh_proofstream* ps, uint8_t out[32]);      /* :58-64 */
/* FRI commit against a library ProofStream: pushes each root, samples alpha = sample(prover_fiat_shamir())
 * (G3) and finally pushes the last codeword. */
int32_t sh_fri_commit_proofstream(sh_ctx* ctx, const uint64_t* codeword, size_t N, const sh_fri_params* params,
                                  sh_proofstream* ps, uint8_t* roots_out, uint64_t* last_codeword_out, size_t* last_len);
int32_t sh_dev_fri_commit_proofstream(sh_ctx* ctx, const uint64_t* d_codeword, size_t N, const sh_fri_params* params,
                                      sh_proofstream* ps, uint8_t* roots_out, uint64_t* last_codeword_out, size_t* last_len);

/* ---------------- FRI query phase, prover and verifier ----------------
 * Not in the reference (it stops at MerkleTree::open/verify and ProofStream); defined after the upstream tutorial the
 * reference follows, with the reference's own primitives: SHA-256 Fiat-Shamir (proof_stream.rs:51-64), Blake2b Merkle
 * paths in MerkleTree::open order (merkle.rs:104-107), checked like is_leaf_in_tree (merkle.rs:114-154).
 *
 * sh_merkle_open_batch: `n_idx` openings of one tree in one pass — all levels are built once on the GPU and the
 * siblings gathered; paths[q] = n-th path, log2(L)·64 bytes each, leaf sibling first.  SH_EBOUNDS / SH_EINVAL as
 * sh_merkle_open.  The _elems form hashes field elements to leaves first (G2). */
int32_t sh_merkle_open_batch(sh_ctx* ctx, const uint8_t* leaves, size_t L, const uint64_t* indices, size_t n_idx, uint8_t* paths);
int32_t sh_dev_merkle_open_elems_batch(sh_ctx* ctx, const uint64_t* d_elems, size_t L, const uint64_t* indices, size_t n_idx, uint8_t* paths);
/* G10: `number` indices in [0,size) with pairwise distinct residues mod reduced_size, from Blake2b512(seed || counter_le64). Host only.
 * SH_EINVAL if number > min(size, reduced_size): that many distinct residues do not exist. */
int32_t sh_fri_sample_indices(const uint8_t seed[32], uint64_t size, uint64_t reduced_size, uint32_t number, uint64_t* out);
/* G11: commit (as sh_fri_commit_proofstream) followed by the query phase.  Items pushed after the last codeword, per
 * round r < R-1: t x {"Leaves":[cw_r[a],cw_r[b],cw_(r+1)[a]]}, then per test the paths of a, b = a + L_r/2 (tree r) and
 * a (tree r+1); a = top_index mod L_r/2.  top_level_indices (t entries, may be NULL) / *n_indices (0 when R < 2). */
int32_t sh_fri_prove(sh_ctx* ctx, const uint64_t* codeword, size_t N, const sh_fri_params* params, sh_proofstream* stream,
                     uint64_t* top_level_indices, size_t* n_indices);
int32_t sh_dev_fri_prove(sh_ctx* ctx, const uint64_t* d_codeword, size_t N, const sh_fri_params* params, sh_proofstream* stream,
                         uint64_t* top_level_indices, size_t* n_indices);
/* G12: verifier, host only; consumes the stream from its read index.  *accepted = 1 iff the roots, the last codeword
 * (root match, degree < L_last/expansion), every colinearity test and every authentication path check out; on 0,
 * sh_last_error() names the failed check.  Returns SH_OK unless the arguments themselves are invalid. */
int32_t sh_fri_verify(sh_proofstream* stream, size_t N, const sh_fri_params* params, int32_t* accepted,
                      uint64_t* top_level_indices, size_t* n_indices);
/* Name and version of the item / sampling formats above (G2, G3, G6, G9-G12), e.g. "stark-anat-fri/1".  The reference crate
 * pins none of them (it has no FRI layer) and they are NOT interoperable with the upstream Python tutorial (which hashes
 * decimal strings, appends bytes(counter) and pickles its stream); a later pin by the reference bumps the version. */
const char* sh_fri_proof_format(void);

/* Batch of B independent commits of equal length N (BASELINE config 4: 1024 x 2^14), codewords back to back
 * (B*N elements).  Rounds run in lock step: every kernel covers all B codewords, each codeword has its own
 * transcript `streams[b]` and challenge.  Results are exactly those of B separate sh_fri_commit_proofstream calls:
 * roots_out[(b*R + r)*64 ..], last_codewords_out[b*last_len ..], and each stream gets its R roots + last codeword. */
int32_t sh_fri_commit_batch(sh_ctx* ctx, const uint64_t* codewords, size_t B, size_t N, const sh_fri_params* params,
                            sh_proofstream** streams, uint8_t* roots_out, uint64_t* last_codewords_out, size_t* last_len);
int32_t sh_dev_fri_commit_batch(sh_ctx* ctx, const uint64_t* d_codewords, size_t B, size_t N, const sh_fri_params* params,
                                sh_proofstream** streams, uint8_t* roots_out, uint64_t* last_codewords_out, size_t* last_len);

/* ---------------- ONE transform / ONE commit over the P GPUs of a node (BASELINE.json config 5) ----------------
 * No counterpart in the reference (a single-threaded CPU crate).  One process (or host thread) per GPU, each with its own sh_ctx;
 * every rank makes the same sh_dist_* call with its own slice.  A vector of n elements lies across the ranks in natural order:
 * rank g holds elements [g·n/P, (g+1)·n/P).  P must be a power of two.
 *
 * Transport = how ranks exchange DEVICE buffers.  sh_dist_create binds it to RCCL (grouped ncclSend/ncclRecv and ncclAllGather
 * over xGMI on the context's stream; librccl is dlopen()ed on first use — STARK_HIP_RCCL_PATH overrides the search): rank 0 calls
 * sh_dist_unique_id, hands the 128 bytes to the other ranks by its own means (the Rust host: MPI, a file, torch.distributed …),
 * every rank calls sh_dist_create with them (a collective call, like ncclCommInitRank).  sh_dist_create_with_transport takes the
 * three operations from the host instead; each must order itself after the work already enqueued on `hip_stream` and leave the
 * stream ordered after its own completion (it may simply synchronise), returns 0 on success, and is called from a library thread. */
#define SH_DIST_ID_BYTES 128
typedef struct sh_dist sh_dist;
typedef struct { int32_t peer; int32_t is_send; void* d_buf; size_t bytes; } sh_p2p_op;
typedef struct {
    void* user;
    /* block s (bytes_per_rank bytes at d_send + s·bytes_per_rank) goes to rank s; rank s's block for this rank lands at d_recv + s·bytes_per_rank */
    int32_t (*all_to_all)(void* user, const void* d_send, void* d_recv, size_t bytes_per_rank, void* hip_stream);
    /* every rank's `bytes` bytes at d_send, concatenated in rank order at d_recv */
    int32_t (*all_gather)(void* user, const void* d_send, void* d_recv, size_t bytes, void* hip_stream);
    /* n_ops point-to-point transfers as one group (never to or from this rank itself) */
    int32_t (*sendrecv)(void* user, const sh_p2p_op* ops, size_t n_ops, void* hip_stream);
} sh_dist_transport;
int32_t sh_dist_unique_id(uint8_t id[SH_DIST_ID_BYTES]);
int32_t sh_dist_create(sh_ctx* ctx, int32_t rank, int32_t world, const uint8_t id[SH_DIST_ID_BYTES], sh_dist** out);
int32_t sh_dist_create_with_transport(sh_ctx* ctx, int32_t rank, int32_t world, const sh_dist_transport* transport, sh_dist** out);
int32_t sh_dist_destroy(sh_dist* d);                                         /* before the context it was made on */
int32_t sh_dist_rank(const sh_dist* d, int32_t* rank, int32_t* world);
/* all three operations of the transport on small known buffers, checked on the host (run once after creation) */
int32_t sh_dist_selftest(sh_dist* d);
/* Four-step NTT of the 2^log_n-point vector whose slice this rank holds: n = n1·n2 (n1 = 2^(log_n/2)), three all-to-all
 * transposes, batched local transforms and the w_n^(j2·k1) twiddle between them.  One rank: the direct transform.
 * input_layout  SH_DIST_NATURAL: d_local = x[g·n/P ..);  SH_DIST_IN_COLUMNS: d_local[c][j1] = x[j1·n2 + g·n2/P + c] (skips the first exchange).
 * output_layout SH_DIST_NATURAL: d_out = X[g·n/P ..);    SH_DIST_OUT_ROWS: d_out[k1l][k2] = X[g·n1/P + k1l + n1·k2] (skips the last one).
 * d_out may be d_local.  n1 and n2 must be multiples of P. */
#define SH_DIST_NATURAL 0
#define SH_DIST_IN_COLUMNS 1
#define SH_DIST_OUT_ROWS 1
int32_t sh_dist_ntt(sh_dist* d, const uint64_t* d_local, uint32_t log_n, int32_t inverse, int32_t input_layout, int32_t output_layout, uint64_t* d_out);
/* coset LDE (G7) of a polynomial whose 2^log_n coefficients lie across the ranks: this rank's natural-order slice of the
 * 2^(log_n+log_expansion) evaluations f(offset·w^i) into d_out (N/P elements; must not alias the input).  Only the coefficient
 * slices move (point to point, rank s -> rank s >> log_expansion); the zero padding is laid down locally. */
int32_t sh_dist_coset_lde(sh_dist* d, const uint64_t* d_local_coeffs, uint32_t log_n, uint32_t log_expansion, const uint64_t offset[2], uint64_t* d_out);
/* the ranks' slices of `count` elements each, concatenated in rank order, on every rank */
int32_t sh_dist_all_gather(sh_dist* d, const uint64_t* d_local, size_t count, uint64_t* d_out);
/* FRI commit of ONE codeword of N elements lying across the ranks (d_local: N/P elements).  Every rank passes its own, identical
 * ProofStream replica and gets identical roots (R x 64 bytes), last codeword and stream (as sh_dev_fri_commit_proofstream on the
 * gathered codeword, byte for byte).  Rounds run sharded — fold fused with the next round's leaf hashing on each rank's
 * half-slices, one 64-byte all-gather of sub-roots and one point-to-point regroup per round — while a rank's half-slice holds at
 * least min_half elements (0 = default 2^16); then the codeword is all-gathered and every rank finishes the remaining rounds
 * with the single-GPU driver.  A rank that falls more than STARK_HIP_CHAIN_TIMEOUT_S (default 20 s) behind fails the commit. */
int32_t sh_dist_fri_commit(sh_dist* d, const uint64_t* d_local, size_t N, const sh_fri_params* params, sh_proofstream* ps, size_t min_half,
                           uint8_t* roots_out, uint64_t* last_codeword_out, size_t* last_len);

/* ---------------- host-only scalar helpers (no GPU involved; used by the driver and the shim) ---------------- */
int32_t sh_sha256(const uint8_t* in, size_t len, uint8_t out[32]);                      /* crate sha2, proof_stream.rs:52-55 */
int32_t sh_blake2b512(const uint8_t* in, size_t len, uint8_t out[64]);                  /* crate blake2, merkle.rs:134-136   */
int32_t sh_merkle_verify(const uint8_t root[64], const uint8_t leaf[64], const uint8_t* path, size_t path_len,
                         size_t index, int32_t* ok);                                   /* merkle.rs:110-154 */
int32_t sh_sample_field(const uint8_t digest[32], uint64_t out[2]);                     /* G3 */
int32_t sh_root_of_unity(uint32_t log_n, uint64_t out[2]);                              /* G1 */
int32_t sh_generator(uint64_t out[2]);                                                  /* G1 */
/* scalar field ops: op 0 add, 1 sub, 2 mul, 3 div, 4 pow — FieldElement operators for single values (field.rs:57-114).
 * Op 4 is TRUE modular exponentiation a^b mod p for any exponent.  It is deliberately not field.rs:37-46, which raises the
 * value as a plain 320-bit integer (I320::pow, base.rs:30-42, wrapping mod 2^320) and reduces once at the end: the two agree
 * while a^b < 2^319 (the crate's only test, 2^160, field.rs:140-151, passes here) and differ beyond. */
int32_t sh_fp_scalar(int32_t op, const uint64_t a[2], const uint64_t b[2], uint64_t out[2]);

#ifdef __cplusplus
}
#endif
#endif
