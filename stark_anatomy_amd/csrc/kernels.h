// kernels.h — host-callable launchers of the HIP kernels (internal; the public ABI is include/stark_hip.h).
// All launchers enqueue on `s` and return the hipError_t of the launch; they never synchronise.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "blake2b.cuh"
#include "fp128.cuh"

// ---- field_kernels.hip ----
enum { FPOP_ADD = 0, FPOP_SUB = 1, FPOP_MUL = 2, FPOP_DIV = 3, FPOP_INV = 4, FPOP_NEW = 5, FPOP_INV_MONT = 6 };
// out[i] = a[i] op b[i]; *d_flag is set to 1 if a zero divisor is met (FPOP_DIV / FPOP_INV / FPOP_INV_MONT: batched
// inversion, one field inversion per 2048 elements).  FPOP_INV_MONT: out = 2^128 / a, the factor that makes
// mont_mul(x, out) = x / a.
hipError_t launch_fp_binop(int op, const fp* a, const fp* b, fp* out, size_t n, uint32_t* d_flag, hipStream_t s);
// Split form of the batched inversion for large n (field_kernels.hip): products of the chunks of 2048 divisors into
// ext[fp_batch_inv_chunks(n)]; the caller inverts ext in place (launch_fp_binop, FPOP_INV for op == FPOP_INV, else FPOP_INV_MONT); finish.
size_t fp_batch_inv_chunks(size_t n);
hipError_t launch_fp_batch_inv_products(int op, const fp* a, const fp* b, size_t n, fp* ext, uint32_t* d_flag, hipStream_t s);
hipError_t launch_fp_batch_inv_finish(int op, const fp* a, const fp* b, fp* out, size_t n, fp* ext, uint32_t* d_flag, hipStream_t s);
// ---- dist_kernels.hip (multi-GPU data movement) ----
// out[c][r] = in[r][c], rows x cols elements (the pack of an all-to-all exchange when cols = P·b)
hipError_t launch_transpose16(const fp* in, fp* out, size_t rows, size_t cols, hipStream_t s);
// out[c][r] = in[r][c]·w_n^((row0 + r)·c): the pack with the four-step twiddle folded in (tw: the two-level table of get_twiddles)
hipError_t launch_transpose16_tw(const fp* in, fp* out, size_t rows, size_t cols, const fp* tw, uint32_t log_n, uint64_t row0, hipStream_t s);
// out[c][s][r] = in[s][c][r], (P, b, a) -> (b, P, a): the unpack after the exchange
hipError_t launch_interleave_blocks(const fp* in, fp* out, size_t P, size_t b, size_t a, hipStream_t s);
// leaves[i] = pairs[i][0], leaves[P + i] = pairs[i][1]
hipError_t launch_pair_leaves(const digest64* pairs, digest64* leaves, size_t P, hipStream_t s);
// ---- query_kernels.hip ----
// One tree of a proof's query phase: the codeword under it, the digest arrays its commit left in memory (level[k] = log2
// of the leaves under one digest of stored[k], ascending, all < log_L), and where its indices / workgroups / paths sit.
#define OPEN_MAX_STORED 8
struct OpenTree {
    const fp* cw;
    const digest64* stored[OPEN_MAX_STORED];
    uint32_t level[OPEN_MAX_STORED];
    uint32_t log_L, n_stored;
    uint32_t idx_off, n_idx;             // its indices: idx[idx_off .. idx_off + n_idx); vals likewise
    uint32_t wg_off;                     // its workgroups: wg_off + j·n_seg + segment, n_seg = n_stored + (n_stored == 0 || level[0] > 0)
    uint64_t path_off;                   // its paths: digest path_off + j·log_L + l = sibling of index j at level l (merkle.rs:70-108 order)
};
// paths and the opened values themselves (vals[idx_off + j] = cw[idx[...]]) for all trees in one launch; h_trees = host copy of d_trees (checked)
hipError_t launch_open_paths(const OpenTree* d_trees, const OpenTree* h_trees, uint32_t n_trees, uint32_t total_wgs, const uint64_t* d_idx, digest64* d_paths, fp* d_vals, hipStream_t s);
// ---- poly_kernels.hip ----
// z[k] = start · w^k − c, k < count   (values of the zerofier X^n − c over one period of a coset; w_mont = w·2^128)
hipError_t launch_zerofier_values(fp* z, size_t count, fp start, fp w_mont, fp c, hipStream_t s);
// out[i] = mont_mul(in[i], factor[i & mask])
hipError_t launch_mul_periodic(const fp* in, const fp* factor, size_t mask, fp* out, size_t n, hipStream_t s);
// f = q · (X^n − c) + r: q (nf − n coefficients, none if nf <= n) and r (n coefficients); c_mont = c·2^128
hipError_t launch_divmod_zerofier(const fp* f, size_t nf, size_t n, fp c_mont, fp* q, fp* r, hipStream_t s);

// ---- ntt_kernels.hip ----
#ifndef NTT_MAX_PASSES
#define NTT_MAX_PASSES 4
#endif
#define NTT_TW_LO_BITS 12                    // lo table: w^i, i < 4096 ; hi table: w^(4096 i)
struct NttPlan {
    uint32_t log_n;
    uint32_t num_passes;
    uint32_t bits[NTT_MAX_PASSES];           // radix bits per pass, sum == log_n; bits[0] is the most significant input digit
    uint32_t tile_log[NTT_MAX_PASSES];       // per pass: elements per LDS tile (log2), 10..12 (ntt_make_plan)
};
// batch: transforms per launch (the tile size is chosen for batch · 2^log_n elements)
void ntt_make_plan(uint32_t log_n, NttPlan* plan, size_t batch = 1);
// tw = [lo (min(n,4096)) | hi (max(n >> 12, 1))] Montgomery-form powers of the (inverse) n-th root.
// scale_mont: Montgomery-form factor applied at the final store (R mod p == 1, or n^-1·R for the inverse).
// `data` is transformed in place; `scratch` holds n elements.
// `batch` contiguous transforms of 2^plan.log_n elements each; `scratch` holds as many elements as `data`.
// Coset LDE fused into the first pass (plans of >= 2 passes, batch 1): that pass reads the n = N >> log_expansion coefficients
// from `coeffs` instead of a zero-padded, pre-scaled copy in `data`, and multiplies each by offset^j from three small tables of
// Montgomery-form powers (launch_pow_table): pow_row[r] = offset^(r << log_row_stride), r < rows; pow_chunk[k] = offset^(k << log_cols),
// k < chunks; pow_col[c] = offset^c, c < cols — shapes from ntt_lde_fusable.
struct NttLdeShape { size_t rows, chunks, cols; uint32_t log_row_stride, log_cols; };
struct NttLdeFuse { const fp* coeffs; uint32_t log_expansion; const fp *pow_row, *pow_chunk, *pow_col; };
bool ntt_lde_fusable(const NttPlan& plan, uint32_t log_expansion, NttLdeShape* shape);
hipError_t launch_pow_table(fp* out, size_t count, fp base_mont, hipStream_t s);
// Direct tables of a transform's inter-pass twiddles: t[pi][(k << log_cols_pi) + col] = w_m^(col·k) for the non-final pass pi
// (m = the pass's sub-block length, col < m/R, k < R), or null: the pass then builds each twiddle from the two-level table
// (one more multiply per element).
struct NttPassTables { const fp* t[NTT_MAX_PASSES]; bool scale_folded; };   // scale_folded: an inverse transform's 1/n rides in the last non-final pass's table
hipError_t launch_pass_twiddles(fp* out, const fp* tw, uint32_t log_n, uint32_t log_m, uint32_t log_r, hipStream_t s);
hipError_t launch_ntt(const NttPlan& plan, fp* data, fp* scratch, const fp* tw, fp scale_mont, bool apply_scale, size_t batch, hipStream_t s, const NttLdeFuse* fuse = nullptr,
                      const NttPassTables* ptab = nullptr);
hipError_t launch_scale_powers(fp* data, size_t count, fp start_mont, fp base, hipStream_t s);
hipError_t launch_twiddle_scale(fp* data, size_t rows, uint32_t log_cols, uint64_t row0, uint64_t col0, const fp* tw, uint32_t log_n, hipStream_t s);
// out[j] = j < n ? coeffs[j]·offset^j : 0   for j < N  (coset pre-scale + zero pad, G7)
hipError_t launch_coset_scale_pad(const fp* coeffs, size_t n, fp offset, fp* out, size_t N, hipStream_t s, size_t batch = 1);   // batch <= 65535 polynomials back to back

// ---- hash_kernels.hip ----
// out[i] = Blake2b512(le16(elems[i]))
hipError_t launch_hash_leaves(const fp* elems, digest64* out, size_t L, hipStream_t s);
// out[i] = root of the 2^K-leaf subtree over leafhash(elems[i·2^K .. (i+1)·2^K)), 0 <= K <= 4
hipError_t launch_leaf_subtree(const fp* elems, digest64* out, size_t n_out, uint32_t K, hipStream_t s);
// out[i] = root of the 2^K-node subtree over in[i·2^K .. (i+1)·2^K), 1 <= K <= 4
hipError_t launch_node_subtree(const digest64* in, digest64* out, size_t n_out, uint32_t K, hipStream_t s);
// single workgroup: n_in (power of two, <= 512) digests -> 1 root, levels built in LDS
#define MERKLE_TAIL_MAX 512
hipError_t launch_merkle_tail(const digest64* in, digest64* root, size_t n_in, hipStream_t s);

// ---- quad_kernels.hip ----
// out[i] = Merkle root of in[i << log_chunk .. +2^log_chunk), 1 <= log_chunk <= 9; four lanes per compression,
// levels in LDS.  With `mailbox` (single-workgroup launch only) the root + `seq` also go to pinned host memory.
#define QUAD_MAX_LOG_CHUNK 9
hipError_t launch_quad_reduce(const digest64* in, digest64* out, size_t n_in, uint32_t log_chunk, uint64_t* mailbox, uint32_t seq, hipStream_t s,
                              uint32_t* counter = nullptr, uint32_t target = 0, const uint32_t* d_abort = nullptr);
// d_abort (here and below): device word that a wait kernel raises when it abandons its round; a kernel given it returns at once
// when it is set, so an abandoned round neither publishes a root nor overwrites a codeword that the retry still needs.

// ---- hash_kernels.hip (FRI round) ----
// generic split-and-fold (any omega): out[i] = (a+b)/2 + c0·omega^-i·(a-b),  c0 = alpha/(2·offset)
// c0_dev != null: the constant is read from device memory instead (a wait kernel put it there); i_base: global index of lo[0] in the
// round's codeword (a rank's slice of a sharded commit)
hipError_t launch_fri_fold(const fp* lo, const fp* hi, size_t half, fp c0_mont, fp omega_inv, fp* out, hipStream_t s, const fp* c0_dev = nullptr, size_t i_base = 0,
                           const uint32_t* d_abort = nullptr);
// A launch that waits for its round's challenge itself (chain_wait.cuh): lane 0 of every workgroup polls the sequence word of the
// round's c0 record until it carries `gen` (bit 31: abort), for at most timeout_ticks of the 100 MHz clock; `d_abort` / `h_abort` as
// launch_wait_c0.  seq == null: no wait.  Meant for records on the card (a poll of pinned host memory is a PCIe read).
struct ChainWait { const uint32_t* seq; uint32_t gen; uint32_t* d_abort; uint32_t* h_abort; uint64_t timeout_ticks; };
// fused round for B codewords of length L stored back to back: fold each into length L/2 (also back to back in
// `out`) with twiddles w_N^-(i << tw_shift) from the inverse table of the top-level domain, and emit the 2^K-leaf
// subtree digests of the folded codewords.  c0 = alpha/(2*offset) in Montgomery form: per codeword from the device
// array `c0_arr`, or `c0_single` when c0_arr is null.
hipError_t launch_fold_leaf_subtree(const fp* cw, size_t L, size_t B, const fp* c0_arr, fp c0_single, const fp* tw_inv, uint32_t log_n_table,
                                    uint32_t tw_shift, fp* out, digest64* digests, uint32_t K, hipStream_t s, const uint32_t* d_abort = nullptr, size_t w_base = 0,
                                    const ChainWait* wait = nullptr);
// wait: the kernel waits for the challenge itself and then takes c0_arr[b] out of the round's c0 record (c0_arr points into it)
// w_base: global index of output 0 in the round's folded codeword (B == 1; a rank's slice of a sharded commit, whose lo | hi halves sit
// back to back in `cw`): the twiddle of output w is w_N^-((w_base + w) << tw_shift)

// ---- round_kernels.hip (the commit's serial chain) ----
// one workgroup: wait until the pinned host word *h_seq carries `gen` (bit 31 = abort), then copy the B fold constants
// from pinned host memory to d_c0.  Gives up after timeout_ticks (100 MHz) and raises *d_abort / *h_abort.
hipError_t launch_wait_c0(const uint32_t* h_seq, uint32_t gen, const fp* h_c0, fp* d_c0, size_t B, uint32_t* d_abort, uint32_t* h_abort, uint64_t timeout_ticks,
                          hipStream_t s);
// the device's 100 MHz wall clock (s_memrealtime) into a pinned host word
hipError_t launch_clock_probe(uint64_t* h_out, hipStream_t s);
// B digests that are already roots -> pinned host record, then the sequence word
hipError_t launch_publish_digests(const digest64* in, size_t B, uint64_t* h_roots, uint32_t* h_seq, uint32_t seq, hipStream_t s, const uint32_t* d_abort = nullptr);
// n elements -> device-mapped pinned host memory (the last codeword of a commit)
hipError_t launch_copy_to_host(const fp* src, fp* h_dst, size_t n, hipStream_t s);
// One small FRI round (B·2^log_half <= 2^15 outputs) in one launch: [wait for the challenge,] fold, leaf-hash and the bottom
// min(7, log_half) tree levels.  Trees that end inside a workgroup publish their roots to the host record (rec_*; `counter` /
// `target` as launch_quad_reduce when there is more than one workgroup), the others leave 128-leaf sub-roots in sub_out.
// fold == false: round 0, `cw` holds the leaves (B trees of 2^log_half) and nothing is waited for.
// Waiting: h_seq != null (only B == 1 and a single workgroup) -> the kernel polls the c0 record itself; else c0 = d_c0[b], filled by a
// launch_wait_c0 in front of this launch.
hipError_t launch_round_wg(bool fold, const fp* cw, fp* out, uint32_t log_half, size_t B, const uint32_t* h_seq, uint32_t gen, const fp* h_c0, const fp* d_c0, uint32_t* d_abort,
                           uint32_t* h_abort, uint64_t timeout_ticks, const fp* tw_inv, uint32_t log_n_table, uint32_t tw_shift, digest64* sub_out,
                           uint64_t* rec_roots, uint32_t* rec_seq, uint32_t* counter, uint32_t target, hipStream_t s);
#define ROUND_WG_MAX_LOG_TOTAL 15
#define ROUND_WG_MAX_LOG_LIMIT 18              // what k_round_wg's indexing allows; STARK_HIP_ROUND_WG_MAX_LOG picks the threshold in use (A/B)
uint32_t round_wg_max_log();                   // round_kernels.hip
