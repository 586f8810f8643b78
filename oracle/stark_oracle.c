/*
 * stark_oracle.c — CPU ORACLE for the FRI-commit hot path.  TEST INFRASTRUCTURE ONLY.
 * See stark_oracle.h for scope, pinning and the list of who may call this.
 * Citations are file:line under /root/reference/src (greged93/stark-anatomy).
 *
 * Build: gcc -O2 -fPIC -shared -o liboracle.so stark_oracle.c   (oracle/Makefile)
 */
#include "stark_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

/* ------------------------------------------------------------------------------------------
 * Optional host threading for the "optimized CPU" baseline leg.  Default 1 thread: every
 * parity test runs the sequential code.  Results are identical for any thread count (disjoint output ranges).
 * ------------------------------------------------------------------------------------------ */
#include <pthread.h>
static int g_threads = 1;
void or_set_threads(int t) { g_threads = t < 1 ? 1 : (t > 256 ? 256 : t); }
int or_get_threads(void) { return g_threads; }
typedef void (*par_fn)(size_t lo, size_t hi, void* arg);
typedef struct { par_fn fn; void* arg; size_t lo, hi; } par_job;
static void* par_tramp(void* p) { par_job* j = (par_job*)p; j->fn(j->lo, j->hi, j->arg); return NULL; }
static void par_for(size_t n, size_t min_chunk, par_fn fn, void* arg) {
    size_t T = (size_t)g_threads;
    if (T > n / (min_chunk ? min_chunk : 1)) T = n / (min_chunk ? min_chunk : 1);
    if (T <= 1) { fn(0, n, arg); return; }
    pthread_t th[256]; par_job jobs[256];
    size_t started = 0;
    for (size_t t = 0; t < T; ++t) {
        jobs[t].fn = fn; jobs[t].arg = arg; jobs[t].lo = n * t / T; jobs[t].hi = n * (t + 1) / T;
        if (t + 1 == T) { fn(jobs[t].lo, jobs[t].hi, arg); break; }
        if (pthread_create(&th[t], NULL, par_tramp, &jobs[t]) != 0) { fn(jobs[t].lo, n, arg); break; }   /* no thread: finish here */
        ++started;
    }
    for (size_t t = 0; t < started; ++t) pthread_join(th[t], NULL);
}

typedef unsigned __int128 u128;

/* field.rs:15  PRIME = 1 + 407 * 2^119 = 0xcb800000000000000000000000000001 */
#define P_HI 0xcb80000000000000ULL
#define P_LO 0x0000000000000001ULL
static inline u128 PRIME(void) { return ((u128)P_HI << 64) | P_LO; }

static inline u128 ld(const uint64_t a[2]) { return ((u128)a[1] << 64) | a[0]; }
static inline void st(uint64_t o[2], u128 v) { o[0] = (uint64_t)v; o[1] = (uint64_t)(v >> 64); }

/* ------------------------------------------------------------------------------------------
 * Fast field path.  Semantics = canonical residues exactly as field.rs:57-114 returns them
 * (SURVEY Q2).  Reduction uses the special form 407*2^119 == -1 (mod p); it deliberately
 * shares nothing with the GPU's Montgomery code.
 * ------------------------------------------------------------------------------------------ */
static u128 fp_new(u128 x) { return x % PRIME(); }                      /* field.rs:30-35 */

static u128 fp_add(u128 a, u128 b) {                                    /* field.rs:57-68 */
    u128 s = a + b;
    if (s < a || s >= PRIME()) s -= PRIME();                            /* wrap past 2^128 or >= p */
    return s;
}
static u128 fp_sub(u128 a, u128 b) {                                    /* field.rs:70-82 */
    return a >= b ? a - b : a + (PRIME() - b);
}
static void mul_128x128(u128 a, u128 b, uint64_t r[4]) {
    uint64_t a0 = (uint64_t)a, a1 = (uint64_t)(a >> 64), b0 = (uint64_t)b, b1 = (uint64_t)(b >> 64);
    u128 p00 = (u128)a0 * b0, p01 = (u128)a0 * b1, p10 = (u128)a1 * b0, p11 = (u128)a1 * b1;
    u128 mid = (p00 >> 64) + (uint64_t)p01 + (uint64_t)p10;
    r[0] = (uint64_t)p00;
    r[1] = (uint64_t)mid;
    u128 hi = (mid >> 64) + (p01 >> 64) + (p10 >> 64) + (uint64_t)p11;
    r[2] = (uint64_t)hi;
    r[3] = (uint64_t)((hi >> 64) + (p11 >> 64));
}
static u128 fp_mul(u128 a, u128 b) {                                    /* field.rs:84-95: (l*r) % p */
    uint64_t x[4];
    mul_128x128(a, b, x);
    /* x = q*2^119 + r ; q = 407*s + u  =>  x == u*2^119 + r - s (mod p) */
    u128 r = ((((u128)x[1]) << 64) | x[0]) & ((((u128)1) << 119) - 1);
    uint64_t q0 = (x[1] >> 55) | (x[2] << 9);
    uint64_t q1 = (x[2] >> 55) | (x[3] << 9);
    uint64_t q2 = (x[3] >> 55);
    u128 cur = q2;
    uint64_t s2 = (uint64_t)(cur / 407); cur = ((cur % 407) << 64) | q1;
    uint64_t s1 = (uint64_t)(cur / 407); cur = ((cur % 407) << 64) | q0;
    uint64_t s0 = (uint64_t)(cur / 407);
    uint64_t u  = (uint64_t)(cur % 407);
    u128 s = ((u128)s1 << 64) | s0;
    if (s2) s += (((u128)0) - PRIME());            /* s = 2^128 + s_low  ->  s - p = s_low + (2^128 - p) */
    else if (s >= PRIME()) s -= PRIME();
    u128 A = ((u128)u << 119) + r;                 /* < 407*2^119 = p-1 */
    return fp_sub(A, s);
}
static u128 fp_pow(u128 a, u128 e) {               /* true modular exponentiation */
    u128 r = 1 % PRIME();
    while (e) { if (e & 1) r = fp_mul(r, a); a = fp_mul(a, a); e >>= 1; }
    return r;
}
static u128 fp_inv(u128 a) { return fp_pow(a, PRIME() - 2); }   /* unique inverse in [1,p): == field.rs:109-110 */

void or_fp_prime(uint64_t out[2]) { st(out, PRIME()); }
void or_fp_new(const uint64_t x[2], uint64_t out[2]) { st(out, fp_new(ld(x))); }
void or_fp_add(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]) { st(out, fp_add(ld(a), ld(b))); }
void or_fp_sub(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]) { st(out, fp_sub(ld(a), ld(b))); }
void or_fp_mul(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]) { st(out, fp_mul(ld(a), ld(b))); }
int or_fp_inv(const uint64_t a[2], uint64_t out[2]) {
    if (ld(a) == 0) return -2;
    st(out, fp_inv(ld(a)));
    return 0;
}
int or_fp_div(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]) {   /* field.rs:97-114 */
    if (ld(b) == 0) return -2;                                              /* field.rs:101-103 panic */
    st(out, fp_mul(ld(a), fp_inv(ld(b))));
    return 0;
}
void or_fp_pow(const uint64_t a[2], const uint64_t e[2], uint64_t out[2]) { st(out, fp_pow(ld(a), ld(e))); }
int or_fp_batch(int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        u128 x = ld(a + 2 * i), y = b ? ld(b + 2 * i) : 0, r;
        switch (op) {
        case 0: r = fp_add(x, y); break;
        case 1: r = fp_sub(x, y); break;
        case 2: r = fp_mul(x, y); break;
        case 3:
            if (b) { if (y == 0) return -2; r = fp_mul(x, fp_inv(y)); }
            else   { if (x == 0) return -2; r = fp_inv(x); }
            break;
        default: return -1;
        }
        st(out + 2 * i, r);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Reference-faithful I320 path (base.rs).  5 x u64 little-endian limbs, two's complement.
 * ------------------------------------------------------------------------------------------ */
static const or_i320 I_ZERO = {{0, 0, 0, 0, 0}}, I_ONE = {{1, 0, 0, 0, 0}};
static int i_sign(or_i320 a) { return (a.v[4] >> 63) != 0; }                 /* base.rs:52-54 */
static int i_is_zero(or_i320 a) { return !(a.v[0] | a.v[1] | a.v[2] | a.v[3] | a.v[4]); }
static int i_eq(or_i320 a, or_i320 b) { return memcmp(a.v, b.v, sizeof a.v) == 0; }  /* base.rs:153-157 */
static or_i320 i_from_u128(u128 x) { or_i320 r = {{(uint64_t)x, (uint64_t)(x >> 64), 0, 0, 0}}; return r; } /* base.rs:186-192 */
static u128 i_to_u128(or_i320 a) { return ((u128)a.v[1] << 64) | a.v[0]; }   /* base.rs:194-198 */

or_i320 or_i320_add(or_i320 a, or_i320 b) {                                  /* base.rs:57-71 */
    or_i320 r; uint64_t carry = 0;
    for (int i = 0; i < 5; ++i) {
        uint64_t s = a.v[i] + b.v[i]; uint64_t c1 = s < a.v[i];
        uint64_t s2 = s + carry;      uint64_t c2 = s2 < s;
        r.v[i] = s2; carry = c1 + c2;
    }
    return r;
}
or_i320 or_i320_neg(or_i320 a) {                                             /* base.rs:81-92 */
    or_i320 r; uint64_t carry = 1;
    for (int i = 0; i < 5; ++i) { uint64_t x = ~a.v[i]; r.v[i] = x + carry; carry = r.v[i] < x; }
    return r;
}
or_i320 or_i320_sub(or_i320 a, or_i320 b) { return or_i320_add(a, or_i320_neg(b)); }  /* base.rs:73-79 */
static or_i320 i_abs(or_i320 a) { return i_sign(a) ? or_i320_neg(a) : a; }   /* base.rs:44-50 */

or_i320 or_i320_mul(or_i320 a, or_i320 b) {                                  /* base.rs:94-123 */
    /* sign-magnitude 5x5 schoolbook into 10 limbs, truncated to the low 5.  Carries are
     * propagated correctly here; the reference's latent carry bug is unreachable
     * for field inputs and is not reproduced. */
    uint64_t res[10] = {0};
    int neg = i_sign(a) ^ i_sign(b);
    or_i320 l = i_abs(a), r = i_abs(b);
    for (int i = 0; i < 5; ++i) {
        uint64_t carry = 0;
        for (int j = 0; j < 5; ++j) {
            u128 t = (u128)l.v[i] * r.v[j] + res[i + j] + carry;
            res[i + j] = (uint64_t)t; carry = (uint64_t)(t >> 64);
        }
        res[i + 5] = carry;
    }
    or_i320 p = {{res[0], res[1], res[2], res[3], res[4]}};
    return neg ? or_i320_neg(p) : p;
}

/* Knuth algorithm D on little-endian u64 limbs: what ruint 1.10.1 `algorithms::div::div`
 * computes for base.rs:134 (quotient of the raw, unsigned limb values). */
static int limbs_len(const uint64_t* x, int n) { while (n > 0 && x[n - 1] == 0) --n; return n; }
static void knuth_divrem(const uint64_t* u, int un, const uint64_t* v, int vn, uint64_t* q, uint64_t* r) {
    /* requires vn >= 1, v[vn-1] != 0; q has un limbs (zero-filled by caller), r has vn limbs */
    if (un < vn) { for (int i = 0; i < vn; ++i) r[i] = i < un ? u[i] : 0; return; }
    if (vn == 1) {
        u128 rem = 0;
        for (int i = un - 1; i >= 0; --i) { u128 cur = (rem << 64) | u[i]; q[i] = (uint64_t)(cur / v[0]); rem = cur % v[0]; }
        r[0] = (uint64_t)rem; return;
    }
    uint64_t vnrm[6], unrm[12];
    int s = __builtin_clzll(v[vn - 1]);
    for (int i = vn - 1; i > 0; --i) vnrm[i] = s ? (v[i] << s) | (v[i - 1] >> (64 - s)) : v[i];
    vnrm[0] = v[0] << s;
    unrm[un] = s ? u[un - 1] >> (64 - s) : 0;
    for (int i = un - 1; i > 0; --i) unrm[i] = s ? (u[i] << s) | (u[i - 1] >> (64 - s)) : u[i];
    unrm[0] = u[0] << s;
    for (int j = un - vn; j >= 0; --j) {
        u128 num = ((u128)unrm[j + vn] << 64) | unrm[j + vn - 1];
        u128 qhat = num / vnrm[vn - 1], rhat = num % vnrm[vn - 1];
        while ((qhat >> 64) || (uint64_t)qhat * (u128)vnrm[vn - 2] > ((rhat << 64) | unrm[j + vn - 2])) {
            --qhat; rhat += vnrm[vn - 1];
            if (rhat >> 64) break;
        }
        /* multiply and subtract */
        uint64_t borrow = 0, carry = 0;
        for (int i = 0; i < vn; ++i) {
            u128 p = (u128)(uint64_t)qhat * vnrm[i] + carry;
            carry = (uint64_t)(p >> 64);
            uint64_t sub = (uint64_t)p;
            uint64_t t = unrm[i + j] - sub; uint64_t b1 = unrm[i + j] < sub;
            uint64_t t2 = t - borrow;        uint64_t b2 = t < borrow;
            unrm[i + j] = t2; borrow = b1 + b2;
        }
        uint64_t t = unrm[j + vn] - carry; uint64_t b1 = unrm[j + vn] < carry;
        uint64_t t2 = t - borrow;          uint64_t b2 = t < borrow;
        unrm[j + vn] = t2;
        if (b1 + b2) {                         /* qhat was one too large: add back */
            --qhat;
            uint64_t c = 0;
            for (int i = 0; i < vn; ++i) {
                u128 sum = (u128)unrm[i + j] + vnrm[i] + c;
                unrm[i + j] = (uint64_t)sum; c = (uint64_t)(sum >> 64);
            }
            unrm[j + vn] += c;
        }
        q[j] = (uint64_t)qhat;
    }
    for (int i = 0; i < vn; ++i) r[i] = s ? (unrm[i] >> s) | (unrm[i + 1] << (64 - s)) : unrm[i];
}
int or_i320_div(or_i320 a, or_i320 b, or_i320* q) {                          /* base.rs:125-137 (sign ignored) */
    if (i_is_zero(b)) return -2;                                             /* base.rs:129-131 */
    uint64_t qq[5] = {0}, rr[5] = {0};
    int un = limbs_len(a.v, 5), vn = limbs_len(b.v, 5);
    if (un > 0) knuth_divrem(a.v, un, b.v, vn, qq, rr);
    memcpy(q->v, qq, sizeof qq);
    return 0;
}
int or_i320_rem(or_i320 a, or_i320 b, or_i320* r) {                          /* base.rs:139-151 */
    or_i320 divisor = i_abs(b), dividend = i_abs(a), d;
    if (or_i320_div(dividend, divisor, &d)) return -2;
    if (i_sign(a)) { *r = or_i320_neg(or_i320_sub(dividend, or_i320_mul(d, divisor))); return 0; }
    *r = or_i320_sub(a, or_i320_mul(d, divisor));
    return 0;
}
int or_i320_cmp(or_i320 a, or_i320 b) {                                      /* base.rs:159-176 */
    int ls = i_sign(a), rs = i_sign(b);
    if (ls && !rs) return -1;
    if (!ls && rs) return 1;
    or_i320 l = i_abs(a), r = i_abs(b);
    int c = 0;
    for (int i = 4; i >= 0; --i) if (l.v[i] != r.v[i]) { c = l.v[i] < r.v[i] ? -1 : 1; break; }
    return ls ? -c : c;
}
or_i320 or_i320_pow(or_i320 a, or_i320 e) {                                  /* base.rs:30-42 (NOT modular) */
    or_i320 result = I_ONE, base = a, exp = e, two = {{2, 0, 0, 0, 0}}, m;
    while (or_i320_cmp(exp, I_ZERO) > 0) {
        or_i320_rem(exp, two, &m);
        if (i_eq(m, I_ONE)) result = or_i320_mul(result, base);
        base = or_i320_mul(base, base);
        or_i320_div(exp, two, &exp);
    }
    return result;
}
int or_ext_euclid(or_i320 a, or_i320 b, or_i320* g, or_i320* s_out, or_i320* t_out) {   /* utils.rs:3-18 */
    or_i320 old_r = a, r = b, old_s = I_ONE, s = I_ZERO, old_t = I_ZERO, t = I_ONE, q, tmp;
    for (;;) {
        if (or_i320_div(old_r, r, &q)) return -2;
        tmp = or_i320_sub(old_r, or_i320_mul(q, r)); old_r = r; r = tmp;
        tmp = or_i320_sub(old_s, or_i320_mul(q, s)); old_s = s; s = tmp;
        tmp = or_i320_sub(old_t, or_i320_mul(q, t)); old_t = t; t = tmp;
        if (i_is_zero(r)) { *g = old_r; *s_out = old_s; *t_out = old_t; return 0; }
    }
}
static u128 ref_add(u128 a, u128 b) {                                        /* field.rs:57-68 */
    or_i320 sum;
    or_i320_rem(or_i320_add(i_from_u128(a), i_from_u128(b)), i_from_u128(PRIME()), &sum);
    return fp_new(i_to_u128(sum));
}
static u128 ref_sub(u128 a, u128 b) {                                        /* field.rs:70-82 */
    or_i320 p = i_from_u128(PRIME()), d;
    or_i320_rem(or_i320_add(or_i320_sub(i_from_u128(a), i_from_u128(b)), p), p, &d);
    return fp_new(i_to_u128(d));
}
static u128 ref_mul(u128 a, u128 b) {                                        /* field.rs:84-95 */
    or_i320 m;
    or_i320_rem(or_i320_mul(i_from_u128(a), i_from_u128(b)), i_from_u128(PRIME()), &m);
    return fp_new(i_to_u128(m));
}
static int ref_div(u128 a, u128 b, u128* out) {                              /* field.rs:97-114 */
    if (b == 0) return -2;
    or_i320 p = i_from_u128(PRIME()), g, s, t, inv, quo;
    or_ext_euclid(i_from_u128(b), p, &g, &s, &t);
    or_i320_rem(or_i320_add(s, p), p, &inv);
    or_i320_rem(or_i320_mul(i_from_u128(a), inv), p, &quo);
    *out = fp_new(i_to_u128(quo));
    return 0;
}
void or_ref_fp_add(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]) { st(out, ref_add(ld(a), ld(b))); }
void or_ref_fp_sub(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]) { st(out, ref_sub(ld(a), ld(b))); }
void or_ref_fp_mul(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]) { st(out, ref_mul(ld(a), ld(b))); }
int or_ref_fp_div(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]) {
    u128 r; int rc = ref_div(ld(a), ld(b), &r); if (!rc) st(out, r); return rc;
}
void or_ref_fp_pow(const uint64_t a[2], const uint64_t e[2], uint64_t out[2]) {  /* field.rs:37-46 (Q1: wraps mod 2^320) */
    or_i320 pw = or_i320_pow(i_from_u128(ld(a)), i_from_u128(ld(e))), m;
    or_i320_rem(pw, i_from_u128(PRIME()), &m);
    st(out, fp_new(i_to_u128(m)));
}

/* ------------------------------------------------------------------------------------------
 * BLAKE2b-512 (RFC 7693), unkeyed — what crate blake2 0.10.6 `Blake2b512` computes
 * (merkle.rs:1, 47-54, 134-136).
 * ------------------------------------------------------------------------------------------ */
static const uint64_t B2B_IV[8] = {
    0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
    0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
static const uint8_t B2B_SIGMA[12][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
static inline uint64_t rotr64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
#define B2B_G(a, b, c, d, x, y) do { \
    v[a] = v[a] + v[b] + (x); v[d] = rotr64(v[d] ^ v[a], 32); \
    v[c] = v[c] + v[d];       v[b] = rotr64(v[b] ^ v[c], 24); \
    v[a] = v[a] + v[b] + (y); v[d] = rotr64(v[d] ^ v[a], 16); \
    v[c] = v[c] + v[d];       v[b] = rotr64(v[b] ^ v[c], 63); } while (0)
static void b2b_compress(uint64_t h[8], const uint8_t block[128], u128 t, int last) {
    uint64_t v[16], m[16];
    for (int i = 0; i < 8; ++i) { v[i] = h[i]; v[i + 8] = B2B_IV[i]; }
    v[12] ^= (uint64_t)t; v[13] ^= (uint64_t)(t >> 64);
    if (last) v[14] = ~v[14];
    for (int i = 0; i < 16; ++i) {
        uint64_t w = 0;
        for (int k = 7; k >= 0; --k) w = (w << 8) | block[8 * i + k];
        m[i] = w;
    }
    for (int r = 0; r < 12; ++r) {
        const uint8_t* s = B2B_SIGMA[r];
        B2B_G(0, 4, 8, 12, m[s[0]], m[s[1]]);   B2B_G(1, 5, 9, 13, m[s[2]], m[s[3]]);
        B2B_G(2, 6, 10, 14, m[s[4]], m[s[5]]);  B2B_G(3, 7, 11, 15, m[s[6]], m[s[7]]);
        B2B_G(0, 5, 10, 15, m[s[8]], m[s[9]]);  B2B_G(1, 6, 11, 12, m[s[10]], m[s[11]]);
        B2B_G(2, 7, 8, 13, m[s[12]], m[s[13]]); B2B_G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
    for (int i = 0; i < 8; ++i) h[i] ^= v[i] ^ v[i + 8];
}
void or_blake2b512(const uint8_t* in, size_t len, uint8_t out[64]) {
    uint64_t h[8];
    for (int i = 0; i < 8; ++i) h[i] = B2B_IV[i];
    h[0] ^= 0x01010000ULL ^ 64;                 /* digest_length=64, key_length=0, fanout=depth=1 */
    uint8_t block[128];
    u128 t = 0;
    while (len > 128) { t += 128; b2b_compress(h, in, t, 0); in += 128; len -= 128; }
    memset(block, 0, sizeof block);
    if (len) memcpy(block, in, len);
    t += len;
    b2b_compress(h, block, t, 1);
    for (int i = 0; i < 8; ++i) for (int k = 0; k < 8; ++k) out[8 * i + k] = (uint8_t)(h[i] >> (8 * k));
}

/* ------------------------------------------------------------------------------------------
 * SHA-256 (FIPS 180-4) — crate sha2 0.10.8 (proof_stream.rs:4, 52-55, 59-63)
 * ------------------------------------------------------------------------------------------ */
static const uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
static inline uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
static void sha256_block(uint32_t st8[8], const uint8_t blk[64]) {
    uint32_t w[64];
    for (int i = 0; i < 16; ++i) w[i] = ((uint32_t)blk[4 * i] << 24) | ((uint32_t)blk[4 * i + 1] << 16) | ((uint32_t)blk[4 * i + 2] << 8) | blk[4 * i + 3];
    for (int i = 16; i < 64; ++i) {
        uint32_t s0 = rotr32(w[i - 15], 7) ^ rotr32(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = rotr32(w[i - 2], 17) ^ rotr32(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = st8[0], b = st8[1], c = st8[2], d = st8[3], e = st8[4], f = st8[5], g = st8[6], h = st8[7];
    for (int i = 0; i < 64; ++i) {
        uint32_t S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25), ch = (e & f) ^ (~e & g);
        uint32_t t1 = h + S1 + ch + SHA_K[i] + w[i];
        uint32_t S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22), mj = (a & b) ^ (a & c) ^ (b & c);
        uint32_t t2 = S0 + mj;
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st8[0] += a; st8[1] += b; st8[2] += c; st8[3] += d; st8[4] += e; st8[5] += f; st8[6] += g; st8[7] += h;
}
void or_sha256(const uint8_t* in, size_t len, uint8_t out[32]) {
    uint32_t s[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    uint64_t bits = (uint64_t)len * 8;
    while (len >= 64) { sha256_block(s, in); in += 64; len -= 64; }
    uint8_t blk[128];
    memset(blk, 0, sizeof blk);
    if (len) memcpy(blk, in, len);
    blk[len] = 0x80;
    size_t tot = (len + 9 <= 64) ? 64 : 128;
    for (int k = 0; k < 8; ++k) blk[tot - 1 - k] = (uint8_t)(bits >> (8 * k));
    sha256_block(s, blk);
    if (tot == 128) sha256_block(s, blk + 64);
    for (int i = 0; i < 8; ++i) { out[4 * i] = (uint8_t)(s[i] >> 24); out[4 * i + 1] = (uint8_t)(s[i] >> 16); out[4 * i + 2] = (uint8_t)(s[i] >> 8); out[4 * i + 3] = (uint8_t)s[i]; }
}

/* ------------------------------------------------------------------------------------------
 * Merkle (merkle.rs)
 * ------------------------------------------------------------------------------------------ */
typedef struct { const uint8_t* in; uint8_t* out; } merkle_level_arg;
static void merkle_level_range(size_t lo, size_t hi, void* p) {
    merkle_level_arg* a = (merkle_level_arg*)p;
    for (size_t i = lo; i < hi; ++i) or_blake2b512(a->in + 128 * i, 128, a->out + 64 * i);
}
int or_merkle_commit(const uint8_t* leaves, size_t L, uint8_t root[64]) {    /* merkle.rs:35-68 */
    if (L == 0 || (L & (L - 1))) return -1;                                  /* merkle.rs:36 assert */
    uint8_t* hashes = (uint8_t*)malloc(L * 64);                              /* merkle.rs:38 */
    if (!hashes) return -4;
    memcpy(hashes, leaves, L * 64);
    size_t n = L;
    uint8_t* other = NULL;                                                   /* threaded legs ping-pong between two buffers */
    while (n > 1) {                                                          /* merkle.rs:48 */
        if (g_threads > 1 && n >= 4096 && (other || (other = (uint8_t*)malloc(L * 32)) != NULL)) {
            merkle_level_arg a = {hashes, other};
            par_for(n / 2, 1024, merkle_level_range, &a);
            uint8_t* t = hashes; hashes = other; other = t;
        } else {
            for (size_t i = 0; i < n / 2; ++i) {                             /* merkle.rs:51-56: H(left || right) */
                uint8_t d[64];
                or_blake2b512(hashes + 128 * i, 128, d);
                memcpy(hashes + 64 * i, d, 64);
            }
        }
        n /= 2;
    }
    memcpy(root, hashes, 64);                                                /* merkle.rs:62 (L==1: root==leaf) */
    free(hashes); free(other);
    return 0;
}
int or_merkle_open(const uint8_t* leaves, size_t L, size_t index, uint8_t* path, size_t* path_len) { /* merkle.rs:70-108 */
    if (L == 0 || (L & (L - 1))) return -1;                                  /* merkle.rs:72 */
    if (index >= L) return -6;                                               /* merkle.rs:75 */
    if (L < 2) return -6;                                                    /* merkle.rs:104 would index out of bounds (Q5) */
    size_t depth = 0; for (size_t t = L; t > 1; t >>= 1) ++depth;
    /* the reference collects sibling-subtree roots top-down then reverses => leaf-sibling first */
    const uint8_t* cur = leaves; size_t len = L, idx = index, k = depth;
    while (len > 2) {                                                        /* merkle.rs:85-101 */
        size_t mid = len / 2; --k;
        if (idx < mid) { or_merkle_commit(cur + mid * 64, mid, path + k * 64); }
        else { or_merkle_commit(cur, mid, path + k * 64); cur += mid * 64; idx -= mid; }
        len = mid;
    }
    memcpy(path, cur + (1 - idx) * 64, 64);                                  /* merkle.rs:104 */
    *path_len = depth;
    return 0;
}
int or_merkle_verify(const uint8_t root[64], const uint8_t leaf[64], const uint8_t* path, size_t path_len, size_t index) { /* merkle.rs:114-154 */
    if (path_len == 0) return -1;                                            /* merkle.rs:126 */
    if (path_len < 8 * sizeof(size_t) && index >= ((size_t)1 << path_len)) return -6;  /* merkle.rs:122 */
    uint8_t cur[64], buf[128];
    memcpy(cur, leaf, 64);
    for (size_t i = 0; i < path_len; ++i) {
        if ((index & 1) == 0) { memcpy(buf, cur, 64); memcpy(buf + 64, path + 64 * i, 64); }   /* merkle.rs:133-134 */
        else { memcpy(buf, path + 64 * i, 64); memcpy(buf + 64, cur, 64); }                   /* merkle.rs:135-136 */
        or_blake2b512(buf, 128, cur);
        index >>= 1;                                                         /* merkle.rs:146 */
    }
    return memcmp(cur, root, 64) == 0 ? 1 : 0;                               /* merkle.rs:153 */
}
typedef struct { const uint64_t* elems; uint8_t* out; } leaves_arg;
static void hash_leaves_range(size_t lo, size_t hi, void* p) {
    leaves_arg* a = (leaves_arg*)p;
    for (size_t i = lo; i < hi; ++i) {
        uint8_t le[16];
        for (int k = 0; k < 8; ++k) { le[k] = (uint8_t)(a->elems[2 * i] >> (8 * k)); le[8 + k] = (uint8_t)(a->elems[2 * i + 1] >> (8 * k)); }
        or_blake2b512(le, 16, a->out + 64 * i);
    }
}
void or_hash_leaves(const uint64_t* elems, size_t L, uint8_t* out) {         /* G2: Blake2b512(value.to_le_bytes()) */
    leaves_arg a = {elems, out};
    par_for(L, 2048, hash_leaves_range, &a);
}
int or_merkle_commit_elems(const uint64_t* elems, size_t L, uint8_t root[64]) {
    if (L == 0 || (L & (L - 1))) return -1;
    uint8_t* leaves = (uint8_t*)malloc(L * 64);
    if (!leaves) return -4;
    or_hash_leaves(elems, L, leaves);
    int rc = or_merkle_commit(leaves, L, root);
    free(leaves);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * G1 roots of unity, G7 NTT / coset LDE, G4 fold, G3 sampling, G5 round count
 * ------------------------------------------------------------------------------------------ */
static u128 generator(void) {                     /* G1: 85408008396924667383611388730472331217, order 2^119 */
    return ((u128)0x4040fbed12ee470fULL << 64) | 0xb5038f9c18f6f7d1ULL; /* checked in tests against the decimal literal */
}
void or_generator(uint64_t out[2]) { st(out, generator()); }
static int root_of_unity(uint32_t log_n, u128* w) {
    if (log_n > 119) return -1;
    u128 r = generator();
    for (uint32_t i = log_n; i < 119; ++i) r = fp_mul(r, r);        /* g^(2^(119-log_n)) */
    *w = r;
    return 0;
}
int or_root_of_unity(uint32_t log_n, uint64_t out[2]) { u128 w; int rc = root_of_unity(log_n, &w); if (!rc) st(out, w); return rc; }

int or_ntt_naive(const uint64_t* in, uint32_t log_n, int inverse, uint64_t* out) {   /* X[k] = sum_j x[j] w^(jk) */
    u128 w; if (root_of_unity(log_n, &w)) return -1;
    size_t n = (size_t)1 << log_n;
    if (inverse) w = fp_inv(w);
    u128 ninv = fp_inv(fp_new((u128)n));
    u128 wk = 1;
    for (size_t k = 0; k < n; ++k) {
        u128 acc = 0, wjk = 1;
        for (size_t j = 0; j < n; ++j) { acc = fp_add(acc, fp_mul(ld(in + 2 * j), wjk)); wjk = fp_mul(wjk, wk); }
        if (inverse) acc = fp_mul(acc, ninv);
        st(out + 2 * k, acc);
        wk = fp_mul(wk, w);
    }
    return 0;
}
int or_ntt(uint64_t* data, uint32_t log_n, int inverse) {          /* iterative radix-2 DIT, natural order in/out */
    u128 w; if (root_of_unity(log_n, &w)) return -1;
    size_t n = (size_t)1 << log_n;
    if (inverse) w = fp_inv(w);
    u128* a = (u128*)malloc(n * sizeof(u128));
    if (!a) return -4;
    for (size_t i = 0; i < n; ++i) {
        size_t r = 0; for (uint32_t b = 0; b < log_n; ++b) if (i >> b & 1) r |= (size_t)1 << (log_n - 1 - b);
        a[r] = ld(data + 2 * i);
    }
    u128* tw = (u128*)malloc((n / 2 + 1) * sizeof(u128));
    if (!tw) { free(a); return -4; }
    tw[0] = 1; for (size_t i = 1; i < n / 2; ++i) tw[i] = fp_mul(tw[i - 1], w);
    for (size_t len = 2; len <= n; len <<= 1) {
        size_t half = len / 2, step = n / len;
        for (size_t s = 0; s < n; s += len)
            for (size_t j = 0; j < half; ++j) {
                u128 u = a[s + j], v = fp_mul(a[s + j + half], tw[j * step]);
                a[s + j] = fp_add(u, v); a[s + j + half] = fp_sub(u, v);
            }
    }
    if (inverse) { u128 ninv = fp_inv(fp_new((u128)n)); for (size_t i = 0; i < n; ++i) a[i] = fp_mul(a[i], ninv); }
    for (size_t i = 0; i < n; ++i) st(data + 2 * i, a[i]);
    free(a); free(tw);
    return 0;
}
int or_coset_lde(const uint64_t* coeffs, uint32_t log_n, uint32_t log_expansion, const uint64_t offset[2], uint64_t* out) {
    /* G7: codeword[i] = f(offset * w_N^i), N = n << log_expansion: scale c[j] by offset^j, zero-pad, size-N NTT */
    uint32_t log_N = log_n + log_expansion;
    if (log_N > 119) return -1;
    size_t n = (size_t)1 << log_n, N = (size_t)1 << log_N;
    u128 off = ld(offset), pw = 1;
    for (size_t j = 0; j < n; ++j) { st(out + 2 * j, fp_mul(ld(coeffs + 2 * j), pw)); pw = fp_mul(pw, off); }
    memset(out + 2 * n, 0, (N - n) * 16);
    return or_ntt(out, log_N, 0);
}
typedef struct { const uint64_t* cw; uint64_t* out; size_t L; u128 a, inv2, winv, d0; } fold_arg;
static void fold_range(size_t lo, size_t hi, void* p) {
    fold_arg* f = (fold_arg*)p;
    u128 d = lo ? fp_mul(f->d0, fp_pow(f->winv, (u128)lo)) : f->d0;
    for (size_t i = lo; i < hi; ++i) {
        u128 t = fp_mul(f->a, d);
        u128 x = fp_mul(fp_add(1, t), ld(f->cw + 2 * i)), y = fp_mul(fp_sub(1, t), ld(f->cw + 2 * (f->L / 2 + i)));
        st(f->out + 2 * i, fp_mul(f->inv2, fp_add(x, y)));
        d = fp_mul(d, f->winv);
    }
}
int or_fri_fold(const uint64_t* cw, size_t L, const uint64_t alpha[2], const uint64_t offset[2], const uint64_t omega[2], uint64_t* out) {
    /* G4: cw'[i] = 2^-1 * ((1 + a/(off*w^i)) * cw[i] + (1 - a/(off*w^i)) * cw[L/2+i]) */
    if (L < 2 || (L & (L - 1))) return -1;
    u128 a = ld(alpha), off = ld(offset), w = ld(omega);
    if (off == 0 || w == 0) return -2;
    fold_arg fa = {cw, out, L, a, fp_inv(2), fp_inv(w), fp_inv(off)};   /* d_i = 1/(off*w^i) = off^-1 * (w^-1)^i */
    par_for(L / 2, 2048, fold_range, &fa);
    return 0;
}
int or_fri_fold_faithful(const uint64_t* cw, size_t L, const uint64_t alpha[2], const uint64_t offset[2], const uint64_t omega[2], uint64_t* out, int use_i320) {
    /* G4 evaluated literally with the reference's operators: one `/` (ext-Euclid) per element */
    if (L < 2 || (L & (L - 1))) return -1;
    u128 a = ld(alpha), off = ld(offset), w = ld(omega), wi = 1, inv2;
    if (off == 0 || w == 0) return -2;
    if (use_i320) ref_div(1, 2, &inv2); else inv2 = fp_inv(2);
    for (size_t i = 0; i < L / 2; ++i) {
        u128 ci = ld(cw + 2 * i), cj = ld(cw + 2 * (L / 2 + i)), r;
        if (use_i320) {
            u128 t; ref_div(a, ref_mul(off, wi), &t);
            r = ref_mul(inv2, ref_add(ref_mul(ref_add(1, t), ci), ref_mul(ref_sub(1, t), cj)));
            wi = ref_mul(wi, w);
        } else {
            u128 t = fp_mul(a, fp_inv(fp_mul(off, wi)));
            r = fp_mul(inv2, fp_add(fp_mul(fp_add(1, t), ci), fp_mul(fp_sub(1, t), cj)));
            wi = fp_mul(wi, w);
        }
        st(out + 2 * i, r);
    }
    return 0;
}
void or_sample(const uint8_t digest[32], uint64_t out[2]) {        /* G3: big-endian integer of the 32 bytes, mod p */
    /* (hi*2^128 + lo) mod p with hi, lo the two 128-bit halves */
    u128 hi = 0, lo = 0;
    for (int i = 0; i < 16; ++i) { hi = (hi << 8) | digest[i]; lo = (lo << 8) | digest[16 + i]; }
    u128 two128 = ((u128)0) - PRIME();                             /* 2^128 mod p = 2^128 - p */
    st(out, fp_add(fp_mul(fp_new(hi), two128), fp_new(lo)));
}
uint32_t or_fri_num_rounds(size_t N, uint32_t expansion, uint32_t num_colinearity_tests) {   /* G5 */
    uint32_t r = 0; size_t len = N;
    while (len > expansion && 4 * (size_t)num_colinearity_tests < len) { len /= 2; ++r; }
    return r;
}

/* ------------------------------------------------------------------------------------------
 * ProofStream (proof_stream.rs).  Items are kept as their compact-JSON fragments, which is
 * all `serialize` (proof_stream.rs:38-40, 67-69 -> serde_json::to_vec(&[S])) observes.
 * ------------------------------------------------------------------------------------------ */
struct or_ps { uint8_t** item; size_t* len; size_t n, cap, read_index; };
or_ps* or_ps_new(void) { return (or_ps*)calloc(1, sizeof(or_ps)); }
void or_ps_free(or_ps* ps) {
    if (!ps) return;
    for (size_t i = 0; i < ps->n; ++i) free(ps->item[i]);
    free(ps->item); free(ps->len); free(ps);
}
static int ps_push_owned(or_ps* ps, uint8_t* buf, size_t len) {
    if (ps->n == ps->cap) {
        size_t nc = ps->cap ? ps->cap * 2 : 8;
        ps->item = (uint8_t**)realloc(ps->item, nc * sizeof(uint8_t*));
        ps->len = (size_t*)realloc(ps->len, nc * sizeof(size_t));
        ps->cap = nc;
    }
    ps->item[ps->n] = buf; ps->len[ps->n] = len; ps->n++;
    return 0;
}
int or_ps_push_raw(or_ps* ps, const uint8_t* json, size_t len) {              /* proof_stream.rs:25-27 */
    uint8_t* b = (uint8_t*)malloc(len ? len : 1);
    memcpy(b, json, len);
    return ps_push_owned(ps, b, len);
}
int or_ps_push_root(or_ps* ps, const uint8_t root[64]) {    /* G6 + merkle.rs:9-13: {"MerkleRoot":{"inner":[b0,...,b63]}} */
    char* b = (char*)malloc(64 * 4 + 64);
    size_t o = (size_t)sprintf(b, "{\"MerkleRoot\":{\"inner\":[");
    for (int i = 0; i < 64; ++i) o += (size_t)sprintf(b + o, i ? ",%u" : "%u", (unsigned)root[i]);
    o += (size_t)sprintf(b + o, "]}}");
    return ps_push_owned(ps, (uint8_t*)b, o);
}
static size_t u128_dec(u128 v, char* out) {                 /* serde_json writes u128 as a bare decimal */
    char tmp[40]; size_t n = 0;
    if (v == 0) tmp[n++] = '0';
    while (v) { tmp[n++] = (char)('0' + (int)(v % 10)); v /= 10; }
    for (size_t i = 0; i < n; ++i) out[i] = tmp[n - 1 - i];
    return n;
}
int or_ps_push_codeword(or_ps* ps, const uint64_t* elems, size_t n) {         /* G6: {"Codeword":[dec,dec,...]} */
    char* b = (char*)malloc(n * 40 + 32);
    size_t o = (size_t)sprintf(b, "{\"Codeword\":[");
    for (size_t i = 0; i < n; ++i) { if (i) b[o++] = ','; o += u128_dec(ld(elems + 2 * i), b + o); }
    b[o++] = ']'; b[o++] = '}';
    return ps_push_owned(ps, (uint8_t*)b, o);
}
size_t or_ps_len(or_ps* ps) { return ps->n; }
int or_ps_pull(or_ps* ps, const uint8_t** json, size_t* len) {                /* proof_stream.rs:29-36 */
    if (ps->read_index >= ps->n) return -6;                                   /* OutOfBoundsReadIndexError */
    *json = ps->item[ps->read_index]; *len = ps->len[ps->read_index]; ps->read_index++;
    return 0;
}
static size_t ps_serialize_n(or_ps* ps, size_t count, uint8_t* out, size_t cap) {  /* proof_stream.rs:67-69 */
    size_t need = 2;
    for (size_t i = 0; i < count; ++i) need += ps->len[i] + (i ? 1 : 0);
    if (!out || cap < need) return need;
    size_t o = 0; out[o++] = '[';
    for (size_t i = 0; i < count; ++i) { if (i) out[o++] = ','; memcpy(out + o, ps->item[i], ps->len[i]); o += ps->len[i]; }
    out[o++] = ']';
    return need;
}
size_t or_ps_serialize(or_ps* ps, uint8_t* out, size_t cap) { return ps_serialize_n(ps, ps->n, out, cap); }
size_t or_ps_serialize_read(or_ps* ps, uint8_t* out, size_t cap) { return ps_serialize_n(ps, ps->read_index, out, cap); }
static int ps_fs(or_ps* ps, size_t count, uint8_t out[32]) {
    size_t need = ps_serialize_n(ps, count, NULL, 0);
    uint8_t* buf = (uint8_t*)malloc(need);
    if (!buf) return -4;
    ps_serialize_n(ps, count, buf, need);
    or_sha256(buf, need, out);
    free(buf);
    return 0;
}
int or_ps_prover_fiat_shamir(or_ps* ps, uint8_t out[32]) { return ps_fs(ps, ps->n, out); }            /* proof_stream.rs:51-56 */
int or_ps_verifier_fiat_shamir(or_ps* ps, uint8_t out[32]) { return ps_fs(ps, ps->read_index, out); } /* proof_stream.rs:58-64 */

/* ------------------------------------------------------------------------------------------
 * FRI commit (G5, G8)
 * ------------------------------------------------------------------------------------------ */
int or_fri_commit(const uint64_t* codeword, size_t N, uint32_t expansion, uint32_t num_colinearity_tests,
                  const uint64_t offset[2], or_ps* ps, int mode, uint8_t* roots_out, uint32_t* num_rounds,
                  uint64_t* last_codeword, size_t* last_len, uint64_t* alphas_out) {
    if (N == 0 || (N & (N - 1))) return -1;
    uint32_t log_N = 0; while (((size_t)1 << log_N) < N) ++log_N;
    uint32_t R = or_fri_num_rounds(N, expansion, num_colinearity_tests);
    u128 omega; if (root_of_unity(log_N, &omega)) return -1;
    u128 off = ld(offset);
    uint64_t* cw = (uint64_t*)malloc(N * 16), *nx = (uint64_t*)malloc(N * 8 + 16);
    if (!cw || !nx) { free(cw); free(nx); return -4; }
    memcpy(cw, codeword, N * 16);
    size_t L = N;
    for (uint32_t r = 0; r < R; ++r) {
        uint8_t root[64];
        int rc = or_merkle_commit_elems(cw, L, root);       /* leaf hash (G2) + merkle.rs:35-68 */
        if (rc) { free(cw); free(nx); return rc; }
        memcpy(roots_out + 64 * r, root, 64);
        or_ps_push_root(ps, root);                          /* proof_stream.rs:25 */
        if (r == R - 1) break;
        uint8_t dig[32]; uint64_t alpha[2], o2[2], w2[2];
        or_ps_prover_fiat_shamir(ps, dig);                  /* proof_stream.rs:51 */
        or_sample(dig, alpha);                              /* G3 */
        if (alphas_out) { alphas_out[2 * r] = alpha[0]; alphas_out[2 * r + 1] = alpha[1]; }
        st(o2, off); st(w2, omega);
        if (mode == 0) or_fri_fold(cw, L, alpha, o2, w2, nx);
        else           or_fri_fold_faithful(cw, L, alpha, o2, w2, nx, 1);
        L /= 2;
        memcpy(cw, nx, L * 16);
        omega = mode == 0 ? fp_mul(omega, omega) : ref_mul(omega, omega);
        off   = mode == 0 ? fp_mul(off, off)     : ref_mul(off, off);
    }
    or_ps_push_codeword(ps, cw, L);                         /* last codeword in the clear */
    memcpy(last_codeword, cw, L * 16);
    *last_len = L; *num_rounds = R;
    free(cw); free(nx);
    return 0;
}

void or_splitmix_elems(uint64_t seed, uint64_t* out, size_t n) {   /* SURVEY §8d */
    uint64_t s = seed;
    for (size_t i = 0; i < n; ++i) {
        uint64_t w[2];
        for (int k = 0; k < 2; ++k) {
            uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
            w[k] = z ^ (z >> 31);
        }
        st(out + 2 * i, fp_new(((u128)w[1] << 64) | w[0]));
    }
}
