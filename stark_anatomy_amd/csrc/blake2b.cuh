// blake2b.cuh — BLAKE2b-512 (RFC 7693, unkeyed, 64-byte digest) single-block compressions for the
// two shapes the FRI commit needs, on the CDNA4 VALU (64-bit add/xor/rotate on 32-bit lanes):
//   node : H(left || right), 128-byte message, t = 128, final   — merkle.rs:51-54 (one compression)
//   leaf : H(value.to_le_bytes()), 16-byte message, t = 16, final — SURVEY.md §8c G2
// Replaces crate blake2 0.10.6 `Blake2b512::{new, update, finalize_reset}` (merkle.rs:47-54).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct alignas(16) digest64 {
    uint64_t h[8];
};

#define B2B_IV0 0x6a09e667f3bcc908ULL
#define B2B_IV1 0xbb67ae8584caa73bULL
#define B2B_IV2 0x3c6ef372fe94f82bULL
#define B2B_IV3 0xa54ff53a5f1d36f1ULL
#define B2B_IV4 0x510e527fade682d1ULL
#define B2B_IV5 0x9b05688c2b3e6c1fULL
#define B2B_IV6 0x1f83d9abfb41bd6bULL
#define B2B_IV7 0x5be0cd19137e2179ULL
// h0 = IV0 ^ 0x01010040 (digest_length 64, key_length 0, fanout 1, depth 1)
#define B2B_H0 (B2B_IV0 ^ 0x01010040ULL)

// 64-bit rotate right.  Device pass: two v_alignbit_b32 on the halves, spelled out — left to the funnel-shift matcher, a rotate
// that is not fused with the xor feeding it comes out as v_lshrrev_b64 + v_lshlrev_b32 + v_or_b32.  n = 32 is a register swap.
__host__ __device__ __forceinline__ uint64_t b2b_rotr(uint64_t x, int n) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    if (n == 32) return ((uint64_t)lo << 32) | hi;
    if (n < 32) return ((uint64_t)__builtin_amdgcn_alignbit(lo, hi, n) << 32) | __builtin_amdgcn_alignbit(hi, lo, n);
    return ((uint64_t)__builtin_amdgcn_alignbit(hi, lo, n - 32) << 32) | __builtin_amdgcn_alignbit(lo, hi, n - 32);
#else
    return (x >> n) | (x << (64 - n));
#endif
}

#define B2B_G(a, b, c, d, x, y)                 \
    do {                                        \
        a = a + b + (x); d = b2b_rotr(d ^ a, 32); \
        c = c + d;       b = b2b_rotr(b ^ c, 24); \
        a = a + b + (y); d = b2b_rotr(d ^ a, 16); \
        c = c + d;       b = b2b_rotr(b ^ c, 63); \
    } while (0)

// One half-round = four independent G's (the columns, or the diagonals).  On gfx950 the two-operand 32-bit ops of G (the
// xors) issue at one wave-instruction per ~2.4 cycles per SIMD, every three-operand or 64-bit op (v_lshl_add_u64, v_alignbit)
// at one per ~4.3, and a stream that alternates between the two classes loses 15-35 % to the switches
// (profiles/r02_valu_opclass_microbench.txt, r02_mixorder_microbench.txt).  So the four G's advance in lock step, one
// operation at a time across all four, with scheduling barriers between the op classes: runs of 8 xors between runs of
// 8-16 adds/rotates instead of whatever interleaving the scheduler picks.  B2B_SB lets scalar and memory instructions
// through, nothing else; on the host pass it is nothing.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(B2B_NO_GROUP)
#define B2B_SB __builtin_amdgcn_sched_barrier(0x4 | 0x10 | 0x80)
#else
#define B2B_SB ((void)0)
#endif
#define B2B_G4(a0, b0, c0, d0, x0, y0, a1, b1, c1, d1, x1, y1, a2, b2, c2, d2, x2, y2, a3, b3, c3, d3, x3, y3)                       \
    do {                                                                                                                             \
        a0 = a0 + b0 + (x0); a1 = a1 + b1 + (x1); a2 = a2 + b2 + (x2); a3 = a3 + b3 + (x3);                                          \
        B2B_SB;                                                                                                                      \
        d0 = b2b_rotr(d0 ^ a0, 32); d1 = b2b_rotr(d1 ^ a1, 32); d2 = b2b_rotr(d2 ^ a2, 32); d3 = b2b_rotr(d3 ^ a3, 32);              \
        B2B_SB;                                                                                                                      \
        c0 = c0 + d0; c1 = c1 + d1; c2 = c2 + d2; c3 = c3 + d3;                                                                      \
        B2B_SB;                                                                                                                      \
        b0 ^= c0; b1 ^= c1; b2 ^= c2; b3 ^= c3;                                                                                      \
        B2B_SB;                                                                                                                      \
        b0 = b2b_rotr(b0, 24); b1 = b2b_rotr(b1, 24); b2 = b2b_rotr(b2, 24); b3 = b2b_rotr(b3, 24);                                  \
        a0 = a0 + b0 + (y0); a1 = a1 + b1 + (y1); a2 = a2 + b2 + (y2); a3 = a3 + b3 + (y3);                                          \
        B2B_SB;                                                                                                                      \
        d0 ^= a0; d1 ^= a1; d2 ^= a2; d3 ^= a3;                                                                                      \
        B2B_SB;                                                                                                                      \
        d0 = b2b_rotr(d0, 16); d1 = b2b_rotr(d1, 16); d2 = b2b_rotr(d2, 16); d3 = b2b_rotr(d3, 16);                                  \
        c0 = c0 + d0; c1 = c1 + d1; c2 = c2 + d2; c3 = c3 + d3;                                                                      \
        B2B_SB;                                                                                                                      \
        b0 ^= c0; b1 ^= c1; b2 ^= c2; b3 ^= c3;                                                                                      \
        B2B_SB;                                                                                                                      \
        b0 = b2b_rotr(b0, 63); b1 = b2b_rotr(b1, 63); b2 = b2b_rotr(b2, 63); b3 = b2b_rotr(b3, 63);                                  \
    } while (0)

#define B2B_ROUND(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15)                                             \
    do {                                                                                                                             \
        B2B_G4(v0, v4, v8, v12, m[s0], m[s1], v1, v5, v9, v13, m[s2], m[s3], v2, v6, v10, v14, m[s4], m[s5], v3, v7, v11, v15, m[s6], m[s7]);     \
        B2B_G4(v0, v5, v10, v15, m[s8], m[s9], v1, v6, v11, v12, m[s10], m[s11], v2, v7, v8, v13, m[s12], m[s13], v3, v4, v9, v14, m[s14], m[s15]); \
    } while (0)

// One final-block compression from the initial chaining value, `t` message bytes.  m[] is indexed
// with compile-time constants only, so it lives in registers and zero words fold away.
__host__ __device__ __forceinline__ void b2b_compress_final(const uint64_t m[16], uint64_t t, uint64_t out[8]) {
    uint64_t v0 = B2B_H0, v1 = B2B_IV1, v2 = B2B_IV2, v3 = B2B_IV3, v4 = B2B_IV4, v5 = B2B_IV5, v6 = B2B_IV6, v7 = B2B_IV7;
    uint64_t v8 = B2B_IV0, v9 = B2B_IV1, v10 = B2B_IV2, v11 = B2B_IV3, v12 = B2B_IV4 ^ t, v13 = B2B_IV5, v14 = ~B2B_IV6, v15 = B2B_IV7;
    B2B_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    B2B_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3);
    B2B_ROUND(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4);
    B2B_ROUND(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8);
    B2B_ROUND(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13);
    B2B_ROUND(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9);
    B2B_ROUND(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11);
    B2B_ROUND(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10);
    B2B_ROUND(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5);
    B2B_ROUND(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0);
    B2B_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    B2B_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3);
    out[0] = B2B_H0 ^ v0 ^ v8;   out[1] = B2B_IV1 ^ v1 ^ v9;  out[2] = B2B_IV2 ^ v2 ^ v10; out[3] = B2B_IV3 ^ v3 ^ v11;
    out[4] = B2B_IV4 ^ v4 ^ v12; out[5] = B2B_IV5 ^ v5 ^ v13; out[6] = B2B_IV6 ^ v6 ^ v14; out[7] = B2B_IV7 ^ v7 ^ v15;
}

// node = Blake2b512(left || right)                              [merkle.rs:51-54]
__host__ __device__ __forceinline__ digest64 b2b_node(const digest64& l, const digest64& r) {
    uint64_t m[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) { m[i] = l.h[i]; m[8 + i] = r.h[i]; }
    digest64 o;
    b2b_compress_final(m, 128, o.h);
    return o;
}
// leaf = Blake2b512(16-byte little-endian canonical value)       [SURVEY.md §8c G2]
__host__ __device__ __forceinline__ digest64 b2b_leaf(uint64_t lo, uint64_t hi) {
    uint64_t m[16];
    m[0] = lo; m[1] = hi;
#pragma unroll
    for (int i = 2; i < 16; ++i) m[i] = 0;
    digest64 o;
    b2b_compress_final(m, 16, o.h);
    return o;
}
