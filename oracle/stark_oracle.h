/*
 * stark_oracle.h — CPU ORACLE for the FRI-commit hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference crate's algorithms (greged93/stark-anatomy,
 * `stark-anat` v0.1.0) plus the spec-gap definitions G1..G8 of SURVEY.md §8(c).  It is the
 * checker for the HIP path; nothing in the product (`stark_anatomy_amd/`, `libstark_hip.so`)
 * may include, link or call it.  Only `tests/`, `__graft_entry__.smoke()` and bench.py's
 * `cpu_baseline` leg use it.
 *
 * PINNING (see DESIGN.md "Oracle"):
 *   - field arithmetic, I320, ext-Euclid: pinned by the reference's own KATs
 *     (field.rs:140-220, base.rs:225-641, utils.rs:24-67) -> tests/test_oracle_reference_kats.py
 *   - SHA-256 + compact JSON transcript: pinned by proof_stream.rs:104-108,188-191,220-223
 *   - Blake2b-512 / Merkle root: the reference holds NO golden digest (merkle.rs tests use OsRng);
 *     pinned against RFC 7693 via Python's hashlib.blake2b (independent implementation) and the
 *     reference's behavioural properties (merkle.rs:179-282).  "Merkle parity unpinned by the
 *     reference itself."
 *   - NTT / LDE / fold / sampling / fri_commit (G1..G8): not in the reference at all -> the
 *     definitions in SURVEY.md §8(c) are the spec; cross-checked by a second, independent
 *     pure-Python restatement (oracle/pyoracle.py).  "parity unpinned" by the reference.
 *   - The reference itself (Rust) cannot be built here (no cargo/rustc) -> no oracle/_ref.
 *
 * All field elements cross this ABI as canonical little-endian u128 = uint64_t[2] {lo, hi}.
 */
#ifndef STARK_ORACLE_H
#define STARK_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- field (field.rs) ---- */
void or_fp_prime(uint64_t out[2]);                                            /* field.rs:15 */
void or_fp_new(const uint64_t x[2], uint64_t out[2]);                         /* field.rs:30-35 */
void or_fp_add(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]);    /* field.rs:57-68 */
void or_fp_sub(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]);    /* field.rs:70-82 */
void or_fp_mul(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]);    /* field.rs:84-95 */
int  or_fp_div(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]);    /* field.rs:97-114; -2 on b==0 */
int  or_fp_inv(const uint64_t a[2], uint64_t out[2]);
void or_fp_pow(const uint64_t a[2], const uint64_t e[2], uint64_t out[2]);    /* true modpow */
/* op: 0 add, 1 sub, 2 mul, 3 div(b may be NULL => inverse of a). returns 0 / -2 */
int  or_fp_batch(int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);

/* ---- reference-faithful I320 path (base.rs, utils.rs): cost model + KAT pinning ---- */
typedef struct { uint64_t v[5]; } or_i320;
or_i320 or_i320_add(or_i320 a, or_i320 b);                                    /* base.rs:57-71 */
or_i320 or_i320_neg(or_i320 a);                                               /* base.rs:81-92 */
or_i320 or_i320_sub(or_i320 a, or_i320 b);                                    /* base.rs:73-79 */
or_i320 or_i320_mul(or_i320 a, or_i320 b);                                    /* base.rs:94-123 */
int     or_i320_div(or_i320 a, or_i320 b, or_i320* q);                        /* base.rs:125-137 */
int     or_i320_rem(or_i320 a, or_i320 b, or_i320* r);                        /* base.rs:139-151 */
or_i320 or_i320_pow(or_i320 a, or_i320 e);                                    /* base.rs:30-42 */
int     or_i320_cmp(or_i320 a, or_i320 b);                                    /* base.rs:159-176 */
int     or_ext_euclid(or_i320 a, or_i320 b, or_i320* g, or_i320* s, or_i320* t); /* utils.rs:3-18 */
/* field ops routed through I320 exactly as field.rs does (B1 cost model) */
void or_ref_fp_add(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]);
void or_ref_fp_sub(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]);
void or_ref_fp_mul(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]);
int  or_ref_fp_div(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]);
void or_ref_fp_pow(const uint64_t a[2], const uint64_t e[2], uint64_t out[2]); /* field.rs:37-46 incl. quirk Q1 */

/* ---- hashes ---- */
void or_blake2b512(const uint8_t* in, size_t len, uint8_t out[64]);           /* RFC 7693; crate blake2 0.10.6 */
void or_sha256(const uint8_t* in, size_t len, uint8_t out[32]);               /* FIPS 180-4; crate sha2 0.10.8 */

/* ---- Merkle (merkle.rs) ---- */
int or_merkle_commit(const uint8_t* leaves, size_t L, uint8_t root[64]);      /* merkle.rs:35-68 */
int or_merkle_open(const uint8_t* leaves, size_t L, size_t index, uint8_t* path, size_t* path_len); /* merkle.rs:70-108 */
int or_merkle_verify(const uint8_t root[64], const uint8_t leaf[64], const uint8_t* path, size_t path_len, size_t index); /* merkle.rs:110-154 */
void or_hash_leaves(const uint64_t* elems, size_t L, uint8_t* out);           /* G2 */
int or_merkle_commit_elems(const uint64_t* elems, size_t L, uint8_t root[64]);

/* ---- NTT / LDE / fold (G1, G4, G7) ---- */
int  or_root_of_unity(uint32_t log_n, uint64_t out[2]);                       /* G1 */
void or_generator(uint64_t out[2]);                                           /* G1: g, order 2^119 */
int  or_ntt(uint64_t* data, uint32_t log_n, int inverse);                     /* G7: natural in, natural out */
int  or_ntt_naive(const uint64_t* in, uint32_t log_n, int inverse, uint64_t* out); /* O(n^2) definition */
int  or_coset_lde(const uint64_t* coeffs, uint32_t log_n, uint32_t log_expansion,
                  const uint64_t offset[2], uint64_t* out);                   /* G7 */
int  or_fri_fold(const uint64_t* cw, size_t L, const uint64_t alpha[2], const uint64_t offset[2],
                 const uint64_t omega[2], uint64_t* out);                     /* G4 (progression of inverses) */
int  or_fri_fold_faithful(const uint64_t* cw, size_t L, const uint64_t alpha[2], const uint64_t offset[2],
                 const uint64_t omega[2], uint64_t* out, int use_i320);       /* G4, literally: one division per element */
void or_sample(const uint8_t digest[32], uint64_t out[2]);                    /* G3 */
uint32_t or_fri_num_rounds(size_t N, uint32_t expansion, uint32_t num_colinearity_tests); /* G5 */

/* ---- ProofStream (proof_stream.rs) ---- */
typedef struct or_ps or_ps;
or_ps* or_ps_new(void);
void   or_ps_free(or_ps*);
int    or_ps_push_raw(or_ps*, const uint8_t* json, size_t len);               /* proof_stream.rs:25-27, any S */
int    or_ps_push_root(or_ps*, const uint8_t root[64]);                       /* G6 {"MerkleRoot":{"inner":[..]}} */
int    or_ps_push_codeword(or_ps*, const uint64_t* elems, size_t n);          /* G6 {"Codeword":[..]} */
size_t or_ps_len(or_ps*);
int    or_ps_pull(or_ps*, const uint8_t** json, size_t* len);                 /* proof_stream.rs:29-36; -6 when exhausted */
size_t or_ps_serialize(or_ps*, uint8_t* out, size_t cap);                     /* proof_stream.rs:38-40, 67-69; returns needed size */
size_t or_ps_serialize_read(or_ps*, uint8_t* out, size_t cap);                /* first read_index items (verifier view) */
int    or_ps_prover_fiat_shamir(or_ps*, uint8_t out[32]);                     /* proof_stream.rs:51-56 */
int    or_ps_verifier_fiat_shamir(or_ps*, uint8_t out[32]);                   /* proof_stream.rs:58-64 */

/* ---- FRI commit (G5, G8) ---- */
/* mode 0: fast field path; mode 1: reference-faithful cost model (I320 + ext-Euclid per fold element). */
int or_fri_commit(const uint64_t* codeword, size_t N, uint32_t expansion, uint32_t num_colinearity_tests,
                  const uint64_t offset[2], or_ps* ps, int mode,
                  uint8_t* roots_out /* R*64 */, uint32_t* num_rounds,
                  uint64_t* last_codeword /* cap N */, size_t* last_len,
                  uint64_t* alphas_out /* (R-1)*2, may be NULL */);

/* deterministic inputs: SplitMix64(seed) -> two u64 -> u128 -> mod p */
void or_splitmix_elems(uint64_t seed, uint64_t* out, size_t n);
/* host threads for leaf hashing / Merkle levels / fold (default 1; same results for any count): the all-cores CPU baseline leg */
void or_set_threads(int t);
int or_get_threads(void);

#ifdef __cplusplus
}
#endif
#endif
