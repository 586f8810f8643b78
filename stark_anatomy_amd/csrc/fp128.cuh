// fp128.cuh — prime-field arithmetic for p = 1 + 407·2^119 (reference: src/field/types/field.rs:15)
// on the CDNA4 integer VALU.  Replaces the reference's I320 widen / schoolbook / Knuth-division /
// ext-Euclid path (src/field/types/base.rs:94-151, src/field/utils.rs:3-18) wholesale; results
// are the same canonical residues in [0,p) that field.rs:57-114 returns.
//
// Representation: 4 x u32 little-endian limbs (16 B, one dwordx4 per element).  Data in HBM is
// always the CANONICAL value.  Multiplication is Montgomery (R = 2^128): mont_mul(a,b) = a·b·R^-1.
// Twiddles/constants are kept in Montgomery form (w·R), so data(canonical) × twiddle(Montgomery)
// -> canonical with a single mont_mul and no conversions.
//
// Reduction exploits the shape of p:  p ≡ 1 (mod 2^119)  =>  p^-1 ≡ 1 − 407·2^119 (mod 2^128),
// so M = T_lo·p^-1 mod 2^128 only touches the top 9 bits of T_lo, and M·p = M + (407·M)·2^119
// costs four 32x32 multiplies by the constant 407.  Subtractive form:
//     a·b·R^-1 = T_hi − floor(M·p / 2^128)   (+p if negative)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FP_HD __host__ __device__ __forceinline__

struct alignas(16) fp {
    uint32_t w[4];
};

// p limbs
#define FP_P0 0x00000001u
#define FP_P3 0xcb800000u
// R mod p = 2^128 - p
#define FP_R0 0xffffffffu
#define FP_R3 0x347fffffu

FP_HD fp fp_make(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { fp r; r.w[0] = a; r.w[1] = b; r.w[2] = c; r.w[3] = d; return r; }
FP_HD fp fp_zero() { return fp_make(0, 0, 0, 0); }
FP_HD fp fp_one() { return fp_make(1, 0, 0, 0); }
FP_HD fp fp_prime() { return fp_make(FP_P0, 0, 0, FP_P3); }
FP_HD fp fp_mont_one() { return fp_make(FP_R0, 0xffffffffu, 0xffffffffu, FP_R3); }                         // R mod p
FP_HD fp fp_mont_r2() { return fp_make(0x0e778236u, 0x5bd53a7fu, 0x1a6aedc2u, 0xaaf4ad9au); }             // R^2 mod p
FP_HD fp fp_from_u64x2(uint64_t lo, uint64_t hi) { return fp_make((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)); }
FP_HD uint64_t fp_lo64(fp a) { return (uint64_t)a.w[0] | ((uint64_t)a.w[1] << 32); }
FP_HD uint64_t fp_hi64(fp a) { return (uint64_t)a.w[2] | ((uint64_t)a.w[3] << 32); }
FP_HD bool fp_is_zero(fp a) { return (a.w[0] | a.w[1] | a.w[2] | a.w[3]) == 0; }
FP_HD bool fp_eq(fp a, fp b) { return a.w[0] == b.w[0] && a.w[1] == b.w[1] && a.w[2] == b.w[2] && a.w[3] == b.w[3]; }

typedef unsigned __int128 fp_u128;
FP_HD fp_u128 fp_to_u128(fp a) { return ((fp_u128)fp_hi64(a) << 64) | fp_lo64(a); }
FP_HD fp fp_from_u128(fp_u128 v) { return fp_from_u64x2((uint64_t)v, (uint64_t)(v >> 64)); }
FP_HD fp_u128 fp_prime_u128() { return ((fp_u128)0xcb80000000000000ULL << 64) | 1u; }

// (a + b) mod p for canonical a, b           [field.rs:57-68]
// Device pass: 32-bit carry chains spelled out.  Through unsigned __int128 the compiler builds the comparisons from 64-bit compares
// and selects every limb twice: 24 VALU instructions for an add and as many for a subtract; the chains below are 15 and 9 (the
// subtract adds p back as "carry-in = borrow" on limb 0 and a masked top limb — p = 2^127·1.59… + 1 has only those two limbs).
// A butterfly is one of each, so an NTT pass loses ~100 of its ~430 instructions per element.
FP_HD fp fp_add(fp a, fp b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FP_ADDSUB_C)
    uint32_t c, br, s0, s1, s2, s3, d0, d1, d2, d3;
    s0 = __builtin_addc(a.w[0], b.w[0], 0u, &c);
    s1 = __builtin_addc(a.w[1], b.w[1], c, &c);
    s2 = __builtin_addc(a.w[2], b.w[2], c, &c);
    s3 = __builtin_addc(a.w[3], b.w[3], c, &c);
    d0 = __builtin_subc(s0, FP_P0, 0u, &br);
    d1 = __builtin_subc(s1, 0u, br, &br);
    d2 = __builtin_subc(s2, 0u, br, &br);
    d3 = __builtin_subc(s3, FP_P3, br, &br);
    const bool ge = c | !br;                     // the sum left 128 bits, or did not go below p
    return fp_make(ge ? d0 : s0, ge ? d1 : s1, ge ? d2 : s2, ge ? d3 : s3);
#else
    fp_u128 x = fp_to_u128(a), y = fp_to_u128(b), s = x + y, d = s - fp_prime_u128();
    bool ge = (s < x) | (s >= fp_prime_u128());
    return fp_from_u128(ge ? d : s);
#endif
}
// (a - b) mod p for canonical a, b           [field.rs:70-82]
FP_HD fp fp_sub(fp a, fp b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FP_ADDSUB_C)
    uint32_t br, c, d0, d1, d2, d3;
    d0 = __builtin_subc(a.w[0], b.w[0], 0u, &br);
    d1 = __builtin_subc(a.w[1], b.w[1], br, &br);
    d2 = __builtin_subc(a.w[2], b.w[2], br, &br);
    d3 = __builtin_subc(a.w[3], b.w[3], br, &br);
    const uint32_t m3 = br ? FP_P3 : 0u;         // + p when it went negative: limb 0 gets the borrow itself (p's limb 0 is 1)
    fp r;
    r.w[0] = __builtin_addc(d0, br, 0u, &c);
    r.w[1] = __builtin_addc(d1, 0u, c, &c);
    r.w[2] = __builtin_addc(d2, 0u, c, &c);
    r.w[3] = d3 + m3 + c;
    return r;
#else
    fp_u128 x = fp_to_u128(a), y = fp_to_u128(b), d = x - y;
    return fp_from_u128(x < y ? d + fp_prime_u128() : d);
#endif
}
// (u + t, u - t) mod p in one block: the three carry chains of a butterfly (the sum, the sum minus p, the difference) interleaved
// by hand, so that every carry's write and read are two instructions apart without wait-state fillers, the "sum >= p" decision is
// one scalar or-not of two carry masks, and the difference takes its "+p" as carry-in = borrow: 21 VALU instructions where
// fp_add + fp_sub compile to 14 + 9 (the compiler turns "subtract 0 with borrow" into a select and a subtract).  Canonical in,
// canonical out, bit-equal to the pair (tools/mm_bench.hip).  The scalar or-not writes SCC: the clobber list must say so — without
// it the scheduler parks the block between an s_cmp and its s_cbranch_scc / s_cselect (DESIGN.md section 8 item 3 tells that story).
__device__ __forceinline__ void fp_butterfly(const fp u, const fp t, fp& sum, fp& diff) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FP_ADDSUB_C) && !defined(FP_BUTTERFLY_C)
    uint32_t s0, s1, s2, s3, d0, d1, d2, d3, x0, x1, x2, x3, m3;
    uint64_t k1, k2, k3, k4;
    asm("v_add_co_u32_e64 %[s0], %[k1], %[u0], %[t0]\n\t"
        "v_sub_co_u32_e64 %[x0], %[k3], %[u0], %[t0]\n\t"
        "v_subrev_co_u32_e64 %[d0], %[k2], 1, %[s0]\n\t"
        "v_addc_co_u32_e64 %[s1], %[k1], %[u1], %[t1], %[k1]\n\t"
        "v_subb_co_u32_e64 %[x1], %[k3], %[u1], %[t1], %[k3]\n\t"
        "v_subbrev_co_u32_e64 %[d1], %[k2], 0, %[s1], %[k2]\n\t"
        "v_addc_co_u32_e64 %[s2], %[k1], %[u2], %[t2], %[k1]\n\t"
        "v_subb_co_u32_e64 %[x2], %[k3], %[u2], %[t2], %[k3]\n\t"
        "v_subbrev_co_u32_e64 %[d2], %[k2], 0, %[s2], %[k2]\n\t"
        "v_addc_co_u32_e64 %[s3], %[k1], %[u3], %[t3], %[k1]\n\t"
        "v_subb_co_u32_e64 %[x3], %[k3], %[u3], %[t3], %[k3]\n\t"
        "v_subb_co_u32_e64 %[d3], %[k2], %[s3], %[p3], %[k2]\n\t"
        "s_orn2_b64 vcc, %[k1], %[k2]\n\t"                                  // sum left 128 bits, or did not go below p  (writes SCC)
        "v_cndmask_b32_e64 %[m3], 0, %[p3], %[k3]\n\t"
        "v_addc_co_u32_e64 %[x0], %[k4], 0, %[x0], %[k3]\n\t"               // difference negative: + p = + borrow on limb 0 ...
        "v_cndmask_b32_e32 %[s0], %[s0], %[d0], vcc\n\t"
        "v_cndmask_b32_e32 %[s1], %[s1], %[d1], vcc\n\t"
        "v_addc_co_u32_e64 %[x1], %[k4], 0, %[x1], %[k4]\n\t"
        "v_cndmask_b32_e32 %[s2], %[s2], %[d2], vcc\n\t"
        "v_cndmask_b32_e32 %[s3], %[s3], %[d3], vcc\n\t"
        "v_addc_co_u32_e64 %[x2], %[k4], 0, %[x2], %[k4]\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32_e64 %[x3], %[k4], %[x3], %[m3], %[k4]\n\t"           // ... and the masked top limb
        : [s0] "=&v"(s0), [s1] "=&v"(s1), [s2] "=&v"(s2), [s3] "=&v"(s3), [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3),
          [x0] "=&v"(x0), [x1] "=&v"(x1), [x2] "=&v"(x2), [x3] "=&v"(x3), [m3] "=&v"(m3),
          [k1] "=&s"(k1), [k2] "=&s"(k2), [k3] "=&s"(k3), [k4] "=&s"(k4)
        : [u0] "v"(u.w[0]), [u1] "v"(u.w[1]), [u2] "v"(u.w[2]), [u3] "v"(u.w[3]), [t0] "v"(t.w[0]), [t1] "v"(t.w[1]), [t2] "v"(t.w[2]), [t3] "v"(t.w[3]),
          [p3] "v"(FP_P3)
        : "vcc", "scc");
    sum = fp_make(s0, s1, s2, s3);
    diff = fp_make(x0, x1, x2, x3);
#else
    sum = fp_add(u, t);
    diff = fp_sub(u, t);
#endif
}
FP_HD fp fp_neg(fp a) { return fp_is_zero(a) ? a : fp_from_u128(fp_prime_u128() - fp_to_u128(a)); }
// a / 2 mod p
FP_HD fp fp_half(fp a) {
    fp_u128 x = fp_to_u128(a);
    if (x & 1) {                       // (x + p) / 2 ; x + p may carry past 2^128
        fp_u128 s = x + fp_prime_u128();
        fp_u128 c = (s < x) ? ((fp_u128)1 << 127) : 0;
        return fp_from_u128((s >> 1) | c);
    }
    return fp_from_u128(x >> 1);
}
// reduce an arbitrary 128-bit value into [0,p)  [field.rs:30-35 `new`]; p > 2^127 so one subtraction
FP_HD fp fp_reduce_once(fp a) {
    fp_u128 x = fp_to_u128(a);
    return fp_from_u128(x >= fp_prime_u128() ? x - fp_prime_u128() : x);
}

// mont_mul(a, b) = a·b·2^-128 mod p, canonical result.  Requires b < p; a may be any 128-bit value.
// [replaces field.rs:84-95 -> base.rs:94-151 + ruint div]
FP_HD fp mont_mul_c(fp a, fp b) {
    uint32_t t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint64_t x = (uint64_t)a.w[j] * b.w[i] + t[i + j] + c;   // <= 2^64-1, no overflow
            t[i + j] = (uint32_t)x;
            c = x >> 32;
        }
        t[i + 4] = (uint32_t)c;
    }
    // M = T_lo · p^-1 mod 2^128 = T_lo − ((T_lo·407 mod 2^9) << 119)
    uint32_t k9 = (t[0] * 407u) & 511u;
    uint32_t m0 = t[0], m1 = t[1], m2 = t[2], m3 = t[3] - (k9 << 23);
    // U = 407·M (137 bits)
    uint64_t x0 = (uint64_t)m0 * 407u;
    uint64_t x1 = (uint64_t)m1 * 407u + (x0 >> 32);
    uint64_t x2 = (uint64_t)m2 * 407u + (x1 >> 32);
    uint64_t x3 = (uint64_t)m3 * 407u + (x2 >> 32);
    uint32_t u0 = (uint32_t)x0, u1 = (uint32_t)x1, u2 = (uint32_t)x2, u3 = (uint32_t)x3, u4 = (uint32_t)(x3 >> 32);
    // floor(M·p / 2^128) = (U >> 9) + carry,  carry = [ (M >> 119) + (U & 511) >= 512 ]
    uint32_t cy = ((m3 >> 23) + (u0 & 511u)) >> 9;
    uint32_t q0 = (u0 >> 9) | (u1 << 23), q1 = (u1 >> 9) | (u2 << 23), q2 = (u2 >> 9) | (u3 << 23), q3 = (u3 >> 9) | (u4 << 23);
    fp_u128 q = (((fp_u128)(((uint64_t)q3 << 32) | q2)) << 64) | (((uint64_t)q1 << 32) | q0);
    q += cy;
    fp_u128 th = (((fp_u128)(((uint64_t)t[7] << 32) | t[6])) << 64) | (((uint64_t)t[5] << 32) | t[4]);
    fp_u128 r = th - q;
    if (th < q) r += fp_prime_u128();
    return fp_from_u128(r);
}
// The same product on the device, laid out for gfx950's 64-bit multiply-add: v_mad_u64_u32 wants its 64-bit addend in an
// even-aligned register pair, so the textbook row-by-row form above compiles with a v_mov for every "carry word into the low
// half of a fresh pair" (36 v_mov in 106 VALU instructions).  Here the sixteen partial products accumulate in place in seven
// 64-bit columns — E_k at words 2k,2k+1 for i+j even, O_k at words 2k+1,2k+2 for i+j odd — whose overflow bit is caught from
// the instruction's own carry-out (one v_addc each), and the columns are merged by two 32-bit carry chains.  The 407·M of the
// reduction is four independent products merged the same way.  71 VALU instructions, no v_mov (tools/mm_bench.hip checks
// bit-equality with the C form and times both).  The assembler's hazard pass does not see inside an asm
// statement, so the two wait states gfx950 wants between a VALU write of an SGPR carry and its VALU read are kept by instruction
// order inside each statement (three instructions between every carry's write and its read); no carry crosses statements.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FP_MONT_MUL_C)
__device__ __forceinline__ uint32_t fp_addc(uint32_t x, uint32_t y, uint32_t cin, uint32_t* cout) { return __builtin_addc(x, y, cin, cout); }
__device__ __forceinline__ uint32_t fp_subc(uint32_t x, uint32_t y, uint32_t bin, uint32_t* bout) { return __builtin_subc(x, y, bin, bout); }
__device__ __forceinline__ fp mont_mul_dev_eo(fp a, fp b) {
    // One asm statement per row a_i·b: its carries are made and used inside it, its accumulators go in and out as operands,
    // so the compiler is free to retire a_i and finished columns between rows (one 25-instruction block held 27 registers
    // at once and cost the NTT kernels three waves of occupancy).
    uint64_t E0, E1, E2, E3, O0, O1, O2, c0, c1, c2;
    uint32_t vE1, vE2, vO0, vO1, vO2, w1, w2;
    asm("v_mad_u64_u32 %[E0], vcc, %[a], %[b0], 0\n\t"
        "v_mad_u64_u32 %[O0], vcc, %[a], %[b1], 0\n\t"
        "v_mad_u64_u32 %[E1], vcc, %[a], %[b2], 0\n\t"
        "v_mad_u64_u32 %[O1], vcc, %[a], %[b3], 0\n\t"
        : [E0] "=&v"(E0), [O0] "=&v"(O0), [E1] "=&v"(E1), [O1] "=&v"(O1)
        : [a] "v"(a.w[0]), [b0] "v"(b.w[0]), [b1] "v"(b.w[1]), [b2] "v"(b.w[2]), [b3] "v"(b.w[3])
        : "vcc");
    asm("v_mad_u64_u32 %[O0], %[c0], %[a], %[b0], %[O0]\n\t"
        "v_mad_u64_u32 %[E1], %[c1], %[a], %[b1], %[E1]\n\t"
        "v_mad_u64_u32 %[O1], %[c2], %[a], %[b2], %[O1]\n\t"
        "v_mad_u64_u32 %[E2], vcc, %[a], %[b3], 0\n\t"
        "v_addc_co_u32_e64 %[vO0], vcc, 0, 0, %[c0]\n\t"
        "v_addc_co_u32_e64 %[vE1], vcc, 0, 0, %[c1]\n\t"
        "v_addc_co_u32_e64 %[vO1], vcc, 0, 0, %[c2]\n\t"
        : [O0] "+v"(O0), [E1] "+v"(E1), [O1] "+v"(O1), [E2] "=&v"(E2), [vO0] "=&v"(vO0), [vE1] "=&v"(vE1), [vO1] "=&v"(vO1),
          [c0] "=&s"(c0), [c1] "=&s"(c1), [c2] "=&s"(c2)
        : [a] "v"(a.w[1]), [b0] "v"(b.w[0]), [b1] "v"(b.w[1]), [b2] "v"(b.w[2]), [b3] "v"(b.w[3])
        : "vcc");
    asm("v_mad_u64_u32 %[E1], %[c0], %[a], %[b0], %[E1]\n\t"
        "v_mad_u64_u32 %[O1], %[c1], %[a], %[b1], %[O1]\n\t"
        "v_mad_u64_u32 %[E2], %[c2], %[a], %[b2], %[E2]\n\t"
        "v_mad_u64_u32 %[O2], vcc, %[a], %[b3], 0\n\t"
        "v_addc_co_u32_e64 %[vE1], vcc, 0, %[vE1], %[c0]\n\t"
        "v_addc_co_u32_e64 %[vO1], vcc, 0, %[vO1], %[c1]\n\t"
        "v_addc_co_u32_e64 %[vE2], vcc, 0, 0, %[c2]\n\t"
        : [E1] "+v"(E1), [O1] "+v"(O1), [E2] "+v"(E2), [O2] "=&v"(O2), [vE1] "+v"(vE1), [vO1] "+v"(vO1), [vE2] "=&v"(vE2),
          [c0] "=&s"(c0), [c1] "=&s"(c1), [c2] "=&s"(c2)
        : [a] "v"(a.w[2]), [b0] "v"(b.w[0]), [b1] "v"(b.w[1]), [b2] "v"(b.w[2]), [b3] "v"(b.w[3])
        : "vcc");
    asm("v_mad_u64_u32 %[O1], %[c0], %[a], %[b0], %[O1]\n\t"
        "v_mad_u64_u32 %[E2], %[c1], %[a], %[b1], %[E2]\n\t"
        "v_mad_u64_u32 %[O2], %[c2], %[a], %[b2], %[O2]\n\t"
        "v_mad_u64_u32 %[E3], vcc, %[a], %[b3], 0\n\t"
        "v_addc_co_u32_e64 %[vO1], vcc, 0, %[vO1], %[c0]\n\t"
        "v_addc_co_u32_e64 %[vE2], vcc, 0, %[vE2], %[c1]\n\t"
        "v_addc_co_u32_e64 %[vO2], vcc, 0, 0, %[c2]\n\t"
        : [O1] "+v"(O1), [E2] "+v"(E2), [O2] "+v"(O2), [E3] "=&v"(E3), [vO1] "+v"(vO1), [vE2] "+v"(vE2), [vO2] "=&v"(vO2),
          [c0] "=&s"(c0), [c1] "=&s"(c1), [c2] "=&s"(c2)
        : [a] "v"(a.w[3]), [b0] "v"(b.w[0]), [b1] "v"(b.w[1]), [b2] "v"(b.w[2]), [b3] "v"(b.w[3])
        : "vcc");
    (void)w1; (void)w2;
    // T = E + (O << 32) + overflow words (vO0 at word 3, vE1 at 4, vO1 at 5, vE2 at 6, vO2 at 7)
    uint32_t cy, t0, t1, t2, t3, t4, t5, t6, t7;
    t0 = (uint32_t)E0;
    t1 = fp_addc((uint32_t)(E0 >> 32), (uint32_t)O0, 0, &cy);
    t2 = fp_addc((uint32_t)E1, (uint32_t)(O0 >> 32), cy, &cy);
    t3 = fp_addc((uint32_t)(E1 >> 32), (uint32_t)O1, cy, &cy);
    t4 = fp_addc((uint32_t)E2, (uint32_t)(O1 >> 32), cy, &cy);
    t5 = fp_addc((uint32_t)(E2 >> 32), (uint32_t)O2, cy, &cy);
    t6 = fp_addc((uint32_t)E3, (uint32_t)(O2 >> 32), cy, &cy);
    t7 = (uint32_t)(E3 >> 32) + cy;
    t3 = fp_addc(t3, vO0, 0, &cy);
    t4 = fp_addc(t4, vE1, cy, &cy);
    t5 = fp_addc(t5, vO1, cy, &cy);
    t6 = fp_addc(t6, vE2, cy, &cy);
    t7 = t7 + vO2 + cy;
    // M = T_lo · p^-1 mod 2^128 = T_lo − ((T_lo·407 mod 2^9) << 119);  U = 407·M from four independent products
    const uint64_t x0 = (uint64_t)t0 * 407u;
    const uint32_t k9 = (uint32_t)x0 & 511u;
    const uint32_t m3 = t3 - (k9 << 23);
    const uint64_t x1 = (uint64_t)t1 * 407u, x2 = (uint64_t)t2 * 407u, x3 = (uint64_t)m3 * 407u;
    uint32_t u0 = (uint32_t)x0, u1, u2, u3, u4;
    u1 = fp_addc((uint32_t)(x0 >> 32), (uint32_t)x1, 0, &cy);
    u2 = fp_addc((uint32_t)x2, (uint32_t)(x1 >> 32), cy, &cy);
    u3 = fp_addc((uint32_t)(x2 >> 32), (uint32_t)x3, cy, &cy);
    u4 = (uint32_t)(x3 >> 32) + cy;
    // floor(M·p / 2^128) = (U >> 9) + carry,  carry = [ (M >> 119) + (U & 511) >= 512 ]
    const uint32_t cq = ((m3 >> 23) + (u0 & 511u)) >> 9;
    uint32_t q0 = __builtin_amdgcn_alignbit(u1, u0, 9), q1 = __builtin_amdgcn_alignbit(u2, u1, 9), q2 = __builtin_amdgcn_alignbit(u3, u2, 9), q3 = __builtin_amdgcn_alignbit(u4, u3, 9);
    q0 = fp_addc(q0, cq, 0, &cy); q1 = fp_addc(q1, 0, cy, &cy); q2 = fp_addc(q2, 0, cy, &cy); q3 += cy;
    uint32_t bw, r0, r1, r2, r3;
    r0 = fp_subc(t4, q0, 0, &bw); r1 = fp_subc(t5, q1, bw, &bw); r2 = fp_subc(t6, q2, bw, &bw); r3 = fp_subc(t7, q3, bw, &bw);
    // negative: add p = 1 + 0xcb800000·2^96
    const uint32_t m = 0u - bw;                                   // all ones when negative
    r0 = fp_addc(r0, m & 1u, 0, &cy); r1 = fp_addc(r1, 0, cy, &cy); r2 = fp_addc(r2, 0, cy, &cy); r3 = r3 + (m & FP_P3) + cy;
    return fp_make(r0, r1, r2, r3);
}
// Round 3, late: the product by PRODUCT SCANNING.  One 64-bit accumulator walks the seven columns of the 4 x 4 limb product; a
// column's overflow bits (the multiply-add's own carry-out, one v_addc each) are counted straight into the word that becomes the
// accumulator's high half for the next column, so handing a column over costs one full-rate v_mov (the old high word into the
// low half of the next even-aligned pair) where the even/odd-column form above merges its seven accumulators with twelve
// half-rate carry-chain adds.  The reduction folds its carry bit into the subtraction's borrow-in and the final "+p" takes the
// borrow as carry-in (p's limb 0 is 1).  61 VALU instructions, 6 of them full-rate (71, none); bit-equal (tools/mm_bench.hip).
// Carries never cross asm statements; inside one, two instructions or an s_nop sit between a carry's write and its read.
#ifndef FP_MONT_MUL_EO
#define FP_MM_COL2(acc, cc, x0, y0, x1, y1)                                                             \
    asm("v_mad_u64_u32 %[A], vcc, %[X0], %[Y0], %[A]\n\t"                                             \
        "v_mad_u64_u32 %[A], %[C1], %[X1], %[Y1], %[A]\n\t"                                           \
        "s_nop 1\n\t"                                                                                 \
        "v_addc_co_u32_e64 %[CC], vcc, 0, 0, %[C1]\n\t"                                               \
        : [A] "+v"(acc), [CC] "=&v"(cc), [C1] "=&s"(c1)                                               \
        : [X0] "v"(x0), [Y0] "v"(y0), [X1] "v"(x1), [Y1] "v"(y1) : "vcc")
#define FP_MM_COL3(acc, cc, x0, y0, x1, y1, x2, y2)                                                     \
    asm("v_mad_u64_u32 %[A], vcc, %[X0], %[Y0], %[A]\n\t"                                             \
        "v_mad_u64_u32 %[A], %[C1], %[X1], %[Y1], %[A]\n\t"                                           \
        "v_mad_u64_u32 %[A], %[C2], %[X2], %[Y2], %[A]\n\t"                                           \
        "s_nop 0\n\t"                                                                                 \
        "v_addc_co_u32_e64 %[CC], vcc, 0, 0, %[C1]\n\t"                                               \
        "v_addc_co_u32_e64 %[CC], vcc, 0, %[CC], %[C2]\n\t"                                           \
        : [A] "+v"(acc), [CC] "=&v"(cc), [C1] "=&s"(c1), [C2] "=&s"(c2)                               \
        : [X0] "v"(x0), [Y0] "v"(y0), [X1] "v"(x1), [Y1] "v"(y1), [X2] "v"(x2), [Y2] "v"(y2) : "vcc")
#define FP_MM_COL4(acc, cc, x0, y0, x1, y1, x2, y2, x3, y3)                                             \
    asm("v_mad_u64_u32 %[A], vcc, %[X0], %[Y0], %[A]\n\t"                                             \
        "v_mad_u64_u32 %[A], %[C1], %[X1], %[Y1], %[A]\n\t"                                           \
        "v_mad_u64_u32 %[A], %[C2], %[X2], %[Y2], %[A]\n\t"                                           \
        "v_mad_u64_u32 %[A], %[C3], %[X3], %[Y3], %[A]\n\t"                                           \
        "v_addc_co_u32_e64 %[CC], vcc, 0, 0, %[C1]\n\t"                                               \
        "v_addc_co_u32_e64 %[CC], vcc, 0, %[CC], %[C2]\n\t"                                           \
        "v_addc_co_u32_e64 %[CC], vcc, 0, %[CC], %[C3]\n\t"                                           \
        : [A] "+v"(acc), [CC] "=&v"(cc), [C1] "=&s"(c1), [C2] "=&s"(c2), [C3] "=&s"(c3)               \
        : [X0] "v"(x0), [Y0] "v"(y0), [X1] "v"(x1), [Y1] "v"(y1), [X2] "v"(x2), [Y2] "v"(y2), [X3] "v"(x3), [Y3] "v"(y3) : "vcc")
__device__ __forceinline__ uint64_t fp_mm_next(uint64_t acc, uint32_t cc) { return (acc >> 32) | ((uint64_t)cc << 32); }
__device__ __forceinline__ fp mont_mul_dev(fp a, fp b) {
    uint64_t c1, c2, c3, acc;
    uint32_t cc, t0, t1, t2, t3, t4, t5, t6, t7;
    acc = (uint64_t)a.w[0] * b.w[0];
    t0 = (uint32_t)acc; acc >>= 32;
    FP_MM_COL2(acc, cc, a.w[0], b.w[1], a.w[1], b.w[0]);
    t1 = (uint32_t)acc; acc = fp_mm_next(acc, cc);
    FP_MM_COL3(acc, cc, a.w[0], b.w[2], a.w[1], b.w[1], a.w[2], b.w[0]);
    t2 = (uint32_t)acc; acc = fp_mm_next(acc, cc);
    FP_MM_COL4(acc, cc, a.w[0], b.w[3], a.w[1], b.w[2], a.w[2], b.w[1], a.w[3], b.w[0]);
    t3 = (uint32_t)acc; acc = fp_mm_next(acc, cc);
    FP_MM_COL3(acc, cc, a.w[1], b.w[3], a.w[2], b.w[2], a.w[3], b.w[1]);
    t4 = (uint32_t)acc; acc = fp_mm_next(acc, cc);
    FP_MM_COL2(acc, cc, a.w[2], b.w[3], a.w[3], b.w[2]);
    t5 = (uint32_t)acc; acc = fp_mm_next(acc, cc);
    acc = (uint64_t)a.w[3] * b.w[3] + acc;                       // < 2^64: the whole product is
    t6 = (uint32_t)acc; t7 = (uint32_t)(acc >> 32);
    // reduction as before
    uint32_t cy;
    const uint64_t x0 = (uint64_t)t0 * 407u;
    const uint32_t k9 = (uint32_t)x0 & 511u;
    const uint32_t m3 = t3 - (k9 << 23);
    const uint64_t x1 = (uint64_t)t1 * 407u, x2 = (uint64_t)t2 * 407u, x3 = (uint64_t)m3 * 407u;
    uint32_t u0 = (uint32_t)x0, u1, u2, u3, u4;
    u1 = fp_addc((uint32_t)(x0 >> 32), (uint32_t)x1, 0, &cy);
    u2 = fp_addc((uint32_t)x2, (uint32_t)(x1 >> 32), cy, &cy);
    u3 = fp_addc((uint32_t)(x2 >> 32), (uint32_t)x3, cy, &cy);
    u4 = (uint32_t)(x3 >> 32) + cy;
    const uint32_t cq = ((m3 >> 23) + (u0 & 511u)) >> 9;
    uint32_t q0 = __builtin_amdgcn_alignbit(u1, u0, 9), q1 = __builtin_amdgcn_alignbit(u2, u1, 9), q2 = __builtin_amdgcn_alignbit(u3, u2, 9), q3 = __builtin_amdgcn_alignbit(u4, u3, 9);
    uint32_t bw, r0, r1, r2, r3;
    r0 = fp_subc(t4, q0, cq, &bw); r1 = fp_subc(t5, q1, bw, &bw); r2 = fp_subc(t6, q2, bw, &bw); r3 = fp_subc(t7, q3, bw, &bw);
    const uint32_t m3p = bw ? FP_P3 : 0u;
    r0 = fp_addc(r0, 0u, bw, &cy); r1 = fp_addc(r1, 0, cy, &cy); r2 = fp_addc(r2, 0, cy, &cy); r3 = r3 + m3p + cy;
    return fp_make(r0, r1, r2, r3);
}

#else
__device__ __forceinline__ fp mont_mul_dev(fp a, fp b) { return mont_mul_dev_eo(a, b); }
#endif
#endif
FP_HD fp mont_mul(fp a, fp b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FP_MONT_MUL_C)
    return mont_mul_dev(a, b);
#else
    return mont_mul_c(a, b);
#endif
}
FP_HD fp fp_to_mont(fp a) { return mont_mul(a, fp_mont_r2()); }             // a·R
FP_HD fp fp_from_mont(fp a) { return mont_mul(a, fp_one()); }               // a·R^-1
// canonical × canonical -> canonical (two Montgomery multiplies)          [field.rs:84-95]
FP_HD fp fp_mul(fp a, fp b) { return mont_mul(mont_mul(a, b), fp_mont_r2()); }
// a^e for canonical a (true modular exponentiation; SURVEY.md Q1 — NOT field.rs:37-46's wrapping pow)
FP_HD fp fp_pow(fp a, fp_u128 e) {
    fp base = fp_to_mont(a), acc = fp_mont_one();
    while (e) {
        if (e & 1) acc = mont_mul(acc, base);
        base = mont_mul(base, base);
        e >>= 1;
    }
    return fp_from_mont(acc);
}
// a^-1 = a^(p-2): the unique inverse in [1,p) that field.rs:109-110 derives from ext-Euclid. a != 0.
FP_HD fp fp_inv(fp a) { return fp_pow(a, fp_prime_u128() - 2); }
// Inverse inside the Montgomery domain: am = a·R -> a^-1·R, am != 0.  p − 2 = 406·2^119 + (2^119 − 1) is 110010110b
// followed by 119 ones, so the ladder is walked from the top with a 7-bit window over the run of ones:
// 133 squarings + 27 multiplies instead of the 128 + 125 of the plain ladder.
FP_HD fp mont_inv(fp am) {
    fp w = am;                                    // am^(2^k − 1), k = 1..7
#pragma unroll 1
    for (int k = 1; k < 7; ++k) w = mont_mul(mont_mul(w, w), am);
    fp r = am;                                    // am^406: bits after the leading one of 110010110b
#pragma unroll 1
    for (int k = 7; k >= 0; --k) {
        r = mont_mul(r, r);
        if ((406u >> k) & 1u) r = mont_mul(r, am);
    }
#pragma unroll 1
    for (int g = 0; g < 17; ++g) {                // 17 x 7 = 119 ones
#pragma unroll 1
        for (int k = 0; k < 7; ++k) r = mont_mul(r, r);
        r = mont_mul(r, w);
    }
    return r;
}
