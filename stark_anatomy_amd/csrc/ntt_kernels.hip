// ntt_kernels.hip — radix-2^b multi-pass NTT / INTT and coset pre-scale on gfx950.
// Not in the reference; built from field.rs Add/Sub/Mul semantics only, natural order in and out:
//   X[k] = sum_j x[j]·w^(jk),  w = g^(2^119/n).
//
// Decomposition (generalised four-step, no separate transposes).  log_n = b_0 + … + b_{P-1}, each
// b_i <= 8.  Index digits j = (j_0 … j_{P-1}), j_0 most significant; k = k_0 + R_0·k_1 + ….
//   pass p < P-1 : inside every contiguous sub-block of m_p = n/(R_0…R_{p-1}) elements, R_p-point DFTs
//                  over the stride-(m_p/R_p) digit for C contiguous columns at a time, then the
//                  inter-pass twiddle w_{m_p}^(col·k_p); in place (each tile owns its addresses).
//   final pass   : R_{P-1}-point DFTs over the contiguous digit, for C adjacent values of k_0, and the
//                  digit-reversing scatter to natural order — reads C rows of R contiguous elements,
//                  writes R rows of C contiguous elements (the transpose happens inside LDS).
// A tile (R rows × C columns, <= 4096 elements = 64 KiB) lives in LDS for all log2(R) butterfly
// levels; HBM sees one coalesced read and one coalesced write of the array per pass.
// Butterflies are decimation-in-frequency, so rows come out bit-reversed and the reversal is
// absorbed into the store address.  Twiddles come from a two-level table of Montgomery-form powers
// (4096 + n/4096 entries, L2-resident): w^e = lo[e & 4095]·hi[e >> 12].
// Two forms of a pass.  k_ntt_pass_reg (default wherever a pass has 2^6, 2^7 or 2^8 rows and 16-column tiles, i.e. every pass of
// every transform from 2^12 up except 2^17's last): four butterfly levels per wave in registers, row bits moved between
// registers and lanes with v_permlane32_swap / v_permlane16_swap, ONE trip through LDS per tile — see the comment above that
// kernel.  k_ntt_pass (all other shapes, and everything under STARK_HIP_NTT_REG=0): the tile stays in LDS for all its levels,
// two levels per LDS round trip.  Same bits either way (tests/test_gpu_parity.py::test_ntt_register_exchange_pass_matches_lds_pass).
// mont_mul is the asm form of fp128.cuh (a third fewer instructions).  Round 2 first kept the plain-C form here: with the asm
// form k_ntt_pass took 79 VGPRs and a 1024-lane workgroup no longer fitted twice on a CU (1.16 ms against 1.005 at 2^24).
// Since the last level group is a specialisation of its own (ntt_levels<S, true>) the kernel needs 60, and the asm form
// wins: 0.95 -> 0.90 ms (-DFP_MONT_MUL_C brings the C form back).
#include "kernels.h"
#include <string.h>
#ifdef NTT_TRACE
#include <vector>
#include <cstdio>
#endif

struct NttPassParams {
    const fp* in;
    fp* out;
    const fp* tw;                 // [lo | hi]
    uint32_t log_n, log_m, log_r, log_c;
    uint32_t is_final, log_r0;    // final pass: bits of digit 0
    uint32_t num_mid, mid_bits[NTT_MAX_PASSES];   // final pass: middle digit widths, most significant first
    uint32_t apply_scale;
    uint32_t log_tiles;           // tiles per transform (log2); blockIdx.x >> log_tiles = index in the batch
    fp scale;                     // Montgomery form
    // first pass of a coset LDE (fuse_log_e > 0): `in` holds only the n = 2^(log_n - fuse_log_e) coefficients; the rows of a
    // tile that fall into the zero padding are not read, and a coefficient is scaled by offset^j on its way into LDS.
    // Montgomery-form power tables: pow_row[r] = offset^(r·n/R... the row stride), pow_chunk[k] = offset^(k·C), pow_col[c] = offset^c.
    uint32_t fuse_log_e;
    const fp *pow_row, *pow_chunk, *pow_col;
    const fp* tw_pass;            // non-final pass: direct table of its inter-pass twiddles, [k][col] (NttPassTables), or null
#ifdef NTT_TRACE
    unsigned long long* trace;    // experiment build: per workgroup 8 stamps of the 100 MHz clock (start, filled, 4 level groups, drained) + XCC/CU id
#endif
};

__device__ __forceinline__ fp ntt_load_tw(const fp* __restrict__ tw, uint32_t log_n, size_t e) {
    fp lo = tw[e & ((1u << NTT_TW_LO_BITS) - 1)];
    if (log_n <= NTT_TW_LO_BITS) return lo;
    fp hi = tw[(1u << NTT_TW_LO_BITS) + (e >> NTT_TW_LO_BITS)];
    return mont_mul(lo, hi);
}

#ifndef NTT_THREADS
#define NTT_THREADS 1024
#endif
#ifndef NTT_TILE_LOG
#define NTT_TILE_LOG 12               // elements per tile (log2)
#endif
#ifndef NTT_MAX_PASS_BITS
#define NTT_MAX_PASS_BITS 8           // radix bits per pass (rows of a tile = 2^bits)
#endif
#ifndef NTT_RADIX_LOG
#define NTT_RADIX_LOG 2
#endif

// Levels lo+S-1 … lo of the in-place DIF over the row index of an R x C tile in LDS (row stride rs), S levels per LDS
// round trip: a work item is the 2^S rows that differ in row bits [lo, lo+S) at one column; it is read once, takes
// its S butterfly levels in registers and is written back to the same addresses (no barrier inside).
// Level with pair distance 2^h multiplies the difference by w_R^(j << (log_r-1-h)), j = row mod 2^h (none for h = 0).
// LO0: the group is the last one (lo == 0), where j = 0 — a multiplication by one — for every butterfly of level h whose
// position inside the work item is a multiple of 2^h: known at compile time and skipped (a quarter of a multiply per element
// and pass; exact, since the operand is canonical).
template <int S, bool LO0>
__device__ __forceinline__ void ntt_levels(fp* __restrict__ x, const fp* __restrict__ twr, uint32_t rs, uint32_t log_r, uint32_t log_c,
                                           uint32_t lo, uint32_t T, uint32_t tid, uint32_t nthreads) {
    const uint32_t C = 1u << log_c;
    for (uint32_t w = tid; w < (T >> S); w += nthreads) {
        const uint32_t c = w & (C - 1), q = w >> log_c;
        const uint32_t r_low = q & ((1u << lo) - 1), r0 = ((q >> lo) << (lo + S)) | r_low;
        fp v[1 << S];
#pragma unroll
        for (int e = 0; e < (1 << S); ++e) v[e] = x[(r0 + ((uint32_t)e << lo)) * rs + c];
#pragma unroll
        for (int s = S - 1; s >= 0; --s) {
            const uint32_t sh = log_r - 1 - (lo + s);
#pragma unroll
            for (int e = 0; e < (1 << S); ++e) {
                if (e & (1 << s)) continue;
                const fp u = v[e], t = v[e | (1 << s)];
                fp d;
                fp_butterfly(u, t, v[e], d);
                if (LO0 ? (e & ((1 << s) - 1)) != 0 : true) d = mont_mul(d, twr[(r_low + ((uint32_t)(e & ((1 << s) - 1)) << lo)) << sh]);
                v[e | (1 << s)] = d;
            }
        }
#pragma unroll
        for (int e = 0; e < (1 << S); ++e) x[(r0 + ((uint32_t)e << lo)) * rs + c] = v[e];
    }
}
#ifndef NTT_WAVES_PER_EU
#define NTT_WAVES_PER_EU 8
#endif
__global__ __launch_bounds__(NTT_THREADS, NTT_WAVES_PER_EU) void k_ntt_pass(NttPassParams p) {      // 8 waves per SIMD = two workgroups per CU: at most 64 VGPRs
    extern __shared__ __align__(16) unsigned char lds_raw[];
    fp* x = reinterpret_cast<fp*>(lds_raw);
    const uint32_t R = 1u << p.log_r, C = 1u << p.log_c, T = R * C;
#ifdef NTT_PAD_ALL
    const uint32_t rs = p.log_c ? C + 1 : C;
#else
    const uint32_t rs = (p.is_final && p.log_c) ? C + 1 : C;      // padded row stride: transposed fill stays conflict-free
#endif
    fp* twr = x + (size_t)R * rs;                                 // w_R^e, e < R/2 (Montgomery form)
    const uint32_t tid = threadIdx.x;
#ifdef NTT_TRACE
#define NTT_STAMP(i) do { if (p.trace && tid == 0) p.trace[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    uint32_t stamp_i = 2;
    NTT_STAMP(0);
    if (p.trace && tid == 0) p.trace[(size_t)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 32);
#else
#define NTT_STAMP(i) do { } while (0)
#endif
    const size_t tile = blockIdx.x & ((1u << p.log_tiles) - 1);
    const size_t xform = (size_t)(blockIdx.x >> p.log_tiles) << p.log_n;   // batched: transforms are contiguous
    const fp* __restrict__ in = p.in + (p.fuse_log_e ? xform >> p.fuse_log_e : xform);   // fused LDE: the inputs are the short coefficient vectors
    fp* __restrict__ out = p.out + xform;

    for (uint32_t e = tid; e < (R >> 1); e += blockDim.x) twr[e] = ntt_load_tw(p.tw, p.log_n, (size_t)e << (p.log_n - p.log_r));

    size_t base, row_stride_in;
    uint32_t colbase = 0;
    if (!p.is_final) {
        const uint32_t log_mp = p.log_m - p.log_r;                // sub-sub-block (column count), log2
        const size_t chunks = (size_t)1 << (log_mp - p.log_c);
        const size_t block = tile / chunks, chunk = tile % chunks;
        colbase = (uint32_t)(chunk << p.log_c);
        base = (block << p.log_m) + colbase;
        row_stride_in = (size_t)1 << log_mp;
        if (p.fuse_log_e) {                                       // j = r·row_stride + colbase + c;  x = j < n ? coeff[j]·offset^j : 0
            fp* pc = twr + (R >> 1);                              // offset^(colbase + c), c < C
            if (tid < C) pc[tid] = mont_mul(p.pow_chunk[chunk], p.pow_col[tid]);
            __syncthreads();
            const uint32_t r_n = R >> p.fuse_log_e;               // rows below it hold coefficients
            for (uint32_t idx = tid; idx < T; idx += blockDim.x) {
                const uint32_t r = idx >> p.log_c, c = idx & (C - 1);
                fp v = fp_zero();
                if (r < r_n) v = mont_mul(in[base + (size_t)r * row_stride_in + c], mont_mul(p.pow_row[r], pc[c]));
                x[r * rs + c] = v;
            }
        } else
        for (uint32_t idx = tid; idx < T; idx += blockDim.x) {
            const uint32_t r = idx >> p.log_c, c = idx & (C - 1);
            x[r * rs + c] = in[base + (size_t)r * row_stride_in + c];
        }
    } else {
        const size_t chunks = (size_t)1 << (p.log_r0 - p.log_c);  // chunks of digit-0 values
        const size_t mid = tile / chunks, chunk = tile % chunks;
        colbase = (uint32_t)(chunk << p.log_c);
        base = mid << p.log_r;
        row_stride_in = (size_t)1 << (p.log_n - p.log_r0);        // stride of digit 0 in the in-place layout
        for (uint32_t idx = tid; idx < T; idx += blockDim.x) {
            const uint32_t c = idx >> p.log_r, r = idx & (R - 1); // contiguous digit fastest: coalesced rows
            x[r * rs + c] = in[(size_t)(colbase + c) * row_stride_in + base + r];
        }
    }
    __syncthreads();
    NTT_STAMP(1);

    // log2(R) decimation-in-frequency levels, NTT_RADIX_LOG at a time in registers: one LDS round trip and one
    // barrier per group of levels instead of per level.
#ifdef NTT_SKIP_LEVELS
    for (uint32_t hb = 0; hb > 0;) {            // timing experiment only: no butterflies (results are wrong)
#else
    for (uint32_t hb = p.log_r; hb > 0;) {
#endif
        const uint32_t S = hb < NTT_RADIX_LOG ? hb : NTT_RADIX_LOG, lo = hb - S;
        if (lo) ntt_levels<NTT_RADIX_LOG, false>(x, twr, rs, p.log_r, p.log_c, lo, T, tid, blockDim.x);      // lo > 0: a full group
        else switch (S) {
            case 1: ntt_levels<1, true>(x, twr, rs, p.log_r, p.log_c, 0, T, tid, blockDim.x); break;
            case 2: ntt_levels<2, true>(x, twr, rs, p.log_r, p.log_c, 0, T, tid, blockDim.x); break;
#if NTT_RADIX_LOG >= 3
            case 3: ntt_levels<3, true>(x, twr, rs, p.log_r, p.log_c, 0, T, tid, blockDim.x); break;
#endif
#if NTT_RADIX_LOG >= 4
            case 4: ntt_levels<4, true>(x, twr, rs, p.log_r, p.log_c, 0, T, tid, blockDim.x); break;
#endif
        }
        __syncthreads();
#ifdef NTT_TRACE
        if (stamp_i < 6) { NTT_STAMP(stamp_i); ++stamp_i; }
#endif
        hb = lo;
    }

    if (!p.is_final) {
        const uint32_t tw_shift = p.log_n - p.log_m;              // w_m = w_n^(n/m)
        for (uint32_t idx = tid; idx < T; idx += blockDim.x) {
            const uint32_t r = idx >> p.log_c, c = idx & (C - 1);
            const uint32_t k = __brev(r) >> (32 - p.log_r);
            fp v = x[r * rs + c];
            const size_t e = ((size_t)(colbase + c) * k) << tw_shift;
#ifndef NTT_SKIP_TWIDDLE
            v = mont_mul(v, p.tw_pass ? p.tw_pass[(size_t)k * row_stride_in + colbase + c] : ntt_load_tw(p.tw, p.log_n, e));
#else
            (void)e;
#endif
            out[base + (size_t)k * row_stride_in + c] = v;
        }
    } else {
        // out index = k_0 + R_0·(midrev + M·k_last)
        const size_t chunks = (size_t)1 << (p.log_r0 - p.log_c);
        size_t mid = tile / chunks, midrev = 0;
        uint32_t log_M = 0;
        for (int q = (int)p.num_mid - 1; q >= 0; --q) { log_M += p.mid_bits[q]; }
        {
            uint32_t sh = 0;                                       // digits are stored most-significant-first in `mid`
            size_t rem = mid;
            for (int q = (int)p.num_mid - 1; q >= 0; --q) {        // peel least significant (= last middle digit) first
                const size_t dq = rem & (((size_t)1 << p.mid_bits[q]) - 1);
                rem >>= p.mid_bits[q];
                sh += p.mid_bits[q];
                midrev |= dq << (log_M - sh);
            }
        }
        for (uint32_t idx = tid; idx < T; idx += blockDim.x) {
            const uint32_t r = idx >> p.log_c, c = idx & (C - 1);
            const uint32_t k = p.log_r ? (__brev(r) >> (32 - p.log_r)) : 0;
            fp v = x[r * rs + c];
            if (p.apply_scale) v = mont_mul(v, p.scale);
            out[(size_t)(colbase + c) + ((((size_t)k << log_M) + midrev) << p.log_r0)] = v;
        }
    }
#ifdef NTT_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    NTT_STAMP(6);
#endif
}


// ---------------------------------------------------------------------------------------------------------------------
// k_ntt_pass_reg — a pass of R = 2^LR rows (LR = 6, 7, 8) x 16 columns whose butterflies never go through LDS between level groups.
// A wave's 64 lanes are (t: lane bits 5,4) x (c: the 16 columns, lane bits 3..0) and every lane holds four elements e, so a wave
// holds 16 rows x 16 columns: FOUR row bits (two in e, two in t).  gfx950's v_permlane32_swap / v_permlane16_swap exchange
// register halves between lanes that differ in lane bit 5 / bit 4 — one instruction per dword does the 2 x 2 transpose
// "register bit <-> lane bit" (tools/permlane_probe.hip prints the lane maps) — so after a radix-4 step over the register bits,
// 16 swaps bring the lane's row bits into the registers and the next radix-4 step runs on them: four levels per wave with no
// memory in between.  The columns sit in the lane bits the swaps never touch, so a wave's global loads and stores stay 256-byte
// rows and a level's twiddle is one LDS word per 16 lanes, as in k_ntt_pass.
//   phase 1: wave w holds rows (top four bits | w), loaded straight from HBM: levels LR-1, LR-2 | swap | levels LR-3, LR-4
//   one trip through LDS (the only one; k_ntt_pass makes five): rows regrouped so that wave W holds rows (W | low bits)
//   phase 2: the LR-4 low levels — LR = 8: 3,2 | swap | 1,0;  LR = 7: 2,1 | half swap | 0;  LR = 6: 1,0 — then the inter-pass
//   twiddle and the store, straight from registers.
// Two workgroup barriers (after the w_R table, inside the LDS trip) instead of six; between them the waves run free: VALUBusy
// 87-94 % against 73 % (profiles/r03_ntt_register_exchange.txt).
// FINAL: the pass over the contiguous digit.  Its rows arrive as runs along r, so the tile takes one extra trip through LDS
// (padded rows, as in k_ntt_pass) to turn the lanes from "along r" to "along the 16 columns"; from there on the same two phases,
// and the digit-reversing scatter to natural order goes out of the registers in 256-byte runs.  Fused coset LDE with expansion
// 4 (zpad): rows R/4.. are zero padding, so the top two levels are three multiplies per lane and no add or subtract.
template <int LR, bool LO0>
__device__ __forceinline__ void ntt_radix4(fp (&v)[4], const fp* __restrict__ twr, uint32_t r_low, uint32_t lo) {
    // levels lo+1, lo of the 2^LR-point DIF on v[e] = row (… | e << lo | r_low); as ntt_levels<2, LO0> with log_r = LR
#pragma unroll
    for (int s = 1; s >= 0; --s) {
        const uint32_t sh = LR - 1 - (lo + s);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (e & (1 << s)) continue;
            const fp u = v[e], t = v[e | (1 << s)];
            fp d;
            fp_butterfly(u, t, v[e], d);
            if (LO0 ? (e & ((1 << s) - 1)) != 0 : true) d = mont_mul(d, twr[(r_low + ((uint32_t)(e & ((1 << s) - 1)) << lo)) << sh]);
            v[e | (1 << s)] = d;
        }
    }
}
// register bit 1 <-> lane bit 5
__device__ __forceinline__ void ntt_swap_hi(fp (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e0 = 0; e0 < 2; ++e0) {
            auto r = __builtin_amdgcn_permlane32_swap(v[e0].w[i], v[e0 | 2].w[i], false, false);
            v[e0].w[i] = r[0]; v[e0 | 2].w[i] = r[1];
        }
}
// register bit 0 <-> lane bit 4
__device__ __forceinline__ void ntt_swap_lo(fp (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e1 = 0; e1 < 4; e1 += 2) {
            auto r = __builtin_amdgcn_permlane16_swap(v[e1].w[i], v[e1 | 1].w[i], false, false);
            v[e1].w[i] = r[0]; v[e1 | 1].w[i] = r[1];
        }
}
// v[e] of lane t  <->  v[t] of lane e   (t = lane bits 5,4): the 4 x 4 block that four lanes of one column hold, transposed
__device__ __forceinline__ void ntt_swap_reg_lane(fp (&v)[4]) { ntt_swap_hi(v); ntt_swap_lo(v); }

template <int LR, bool FINAL, bool FUSE, bool DIRECT_TW>      // FUSE: first pass of a fused coset LDE; DIRECT_TW: the pass has a direct table of its inter-pass twiddles
__global__ __launch_bounds__(4 << LR, 8) void k_ntt_pass_reg(NttPassParams p) {
    static_assert(LR >= 6 && LR <= 8, "rows");
    constexpr uint32_t R = 1u << LR, NT = 4u << LR, L2 = LR - 4;  // L2: row bits of the second phase = log2(waves)
    extern __shared__ __align__(16) unsigned char lds_raw[];
    fp* x = reinterpret_cast<fp*>(lds_raw);                       // R rows x 16 columns (row stride 17 in the final pass)
    constexpr uint32_t RS = FINAL ? 17 : 16;
    fp* twr = x + R * RS;                                         // w_R^e, e < R/2 (Montgomery form)
    const uint32_t tid = threadIdx.x, wave = tid >> 6, t = (tid >> 4) & 3, c = tid & 15;
    const size_t tile = blockIdx.x & ((1u << p.log_tiles) - 1);
    const size_t xform = (size_t)(blockIdx.x >> p.log_tiles) << p.log_n;
    const fp* __restrict__ in = p.in + (FUSE ? xform >> p.fuse_log_e : xform);
    fp* __restrict__ out = p.out + xform;
    fp v[4];
    size_t base = 0, row_stride = 0, chunk = 0;
    uint32_t colbase = 0;
    bool zpad = false;
    if (!FINAL) {
        const uint32_t log_mp = p.log_m - LR;
        const size_t chunks = (size_t)1 << (log_mp - 4);
        const size_t block = tile / chunks;
        chunk = tile % chunks;
        colbase = (uint32_t)(chunk << 4);
        base = (block << p.log_m) + colbase + c;
        row_stride = (size_t)1 << log_mp;
        const uint32_t r_n = FUSE ? R >> p.fuse_log_e : R;        // fused LDE: rows from r_n on are zero padding
        zpad = FUSE && p.fuse_log_e == 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t r = ((uint32_t)e << (LR - 2)) | (t << L2) | wave;
            v[e] = r < r_n ? in[base + (size_t)r * row_stride] : fp_zero();
        }
    } else {
        const size_t chunks = (size_t)1 << (p.log_r0 - 4);        // chunks of digit-0 values
        const size_t mid = tile / chunks;
        chunk = tile % chunks;
        colbase = (uint32_t)(chunk << 4);
        base = mid << LR;
        row_stride = (size_t)1 << (p.log_n - p.log_r0);           // stride of digit 0 in the in-place layout
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t idx = tid + q * NT, cc = idx >> LR, r = idx & (R - 1);
            v[q] = in[(size_t)(colbase + cc) * row_stride + base + r];
        }
    }
    if (tid < R / 2) twr[tid] = ntt_load_tw(p.tw, p.log_n, (size_t)tid << (p.log_n - LR));
    if (!FINAL && FUSE) {                                         // coefficient j = r·row_stride + colbase + c takes offset^j
        const fp pcc = mont_mul(p.pow_chunk[chunk], p.pow_col[c]);
        const uint32_t r_n = R >> p.fuse_log_e;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t r = ((uint32_t)e << (LR - 2)) | (t << L2) | wave;
            if (r < r_n) v[e] = mont_mul(v[e], mont_mul(p.pow_row[r], pcc));
        }
    }
    if (FINAL) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t idx = tid + q * NT, cc = idx >> LR, r = idx & (R - 1);
            x[r * RS + cc] = v[q];
        }
    }
    __syncthreads();
    if (FINAL) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = x[(((uint32_t)e << (LR - 2)) | (t << L2) | wave) * RS + c];
    }
    // ---- phase 1: row = e << (LR-2) | t << L2 | wave
    if (zpad) {                                                   // v[1..3] == 0: out = (a, a·w^(2j), a·w^j, a·w^(3j)), j = row mod R/4
        const uint32_t j = (t << L2) | wave;
        v[2] = mont_mul(v[0], twr[j]);
        v[1] = mont_mul(v[0], twr[j << 1]);
        v[3] = mont_mul(v[2], twr[j << 1]);
    } else {
        ntt_radix4<LR, false>(v, twr, (t << L2) | wave, LR - 2);
    }
    ntt_swap_reg_lane(v);                                         // now v[e] = row (t << (LR-2) | e << L2 | wave)
    ntt_radix4<LR, false>(v, twr, wave, L2);
#pragma unroll
    for (int e = 0; e < 4; ++e) x[((t << (LR - 2)) | ((uint32_t)e << L2) | wave) * RS + c] = v[e];    // the rows this wave read: no barrier before
    __syncthreads();
    // ---- phase 2: wave W holds rows W << 4 | (16 rows): the low L2 bits and, for L2 < 4, 4 - L2 of the bits phase 1 has finished
    uint32_t rows[4];                                             // row of v[e] when the levels are done
    if (LR == 8) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = x[((wave << 4) | ((uint32_t)e << 2) | t) * RS + c];
        ntt_radix4<LR, false>(v, twr, t, 2);
        ntt_swap_reg_lane(v);
        ntt_radix4<LR, true>(v, twr, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) rows[e] = (wave << 4) | (t << 2) | (uint32_t)e;
    } else if (LR == 7) {                                         // e = (b2, b1), lane bit 4 = b0, lane bit 5 = b3
        const uint32_t b0 = t & 1, b3 = t >> 1;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = x[((wave << 4) | (b3 << 3) | ((uint32_t)e << 1) | b0) * RS + c];
        ntt_radix4<LR, false>(v, twr, b0, 1);
        ntt_swap_lo(v);                                           // now register bit 0 = b0, lane bit 4 = b1
#pragma unroll
        for (int e = 0; e < 4; e += 2) { const fp u = v[e], w = v[e | 1]; fp_butterfly(u, w, v[e], v[e | 1]); }   // level 0: no twiddle
#pragma unroll
        for (int e = 0; e < 4; ++e) rows[e] = (wave << 4) | (b3 << 3) | ((uint32_t)(e >> 1) << 2) | (b0 << 1) | (uint32_t)(e & 1);   // lane bit 4 now holds b1
    } else {                                                      // LR == 6: e = (b1, b0), t = (b3, b2)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = x[((wave << 4) | (t << 2) | (uint32_t)e) * RS + c];
        ntt_radix4<LR, true>(v, twr, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) rows[e] = (wave << 4) | (t << 2) | (uint32_t)e;
    }
    if (!FINAL) {
        const uint32_t tw_shift = p.log_n - p.log_m;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t k = __brev(rows[e]) >> (32 - LR);
            const fp w = DIRECT_TW ? p.tw_pass[(size_t)k * row_stride + colbase + c] : ntt_load_tw(p.tw, p.log_n, ((size_t)(colbase + c) * k) << tw_shift);
            out[base + (size_t)k * row_stride] = mont_mul(v[e], w);
        }
    } else {
        // out index = k_0 + R_0·(midrev + M·k_last)
        const size_t chunks = (size_t)1 << (p.log_r0 - 4);
        size_t mid = tile / chunks, midrev = 0;
        uint32_t log_M = 0;
        for (int q = (int)p.num_mid - 1; q >= 0; --q) log_M += p.mid_bits[q];
        {
            uint32_t sh = 0;
            size_t rem = mid;
            for (int q = (int)p.num_mid - 1; q >= 0; --q) {
                const size_t dq = rem & (((size_t)1 << p.mid_bits[q]) - 1);
                rem >>= p.mid_bits[q];
                sh += p.mid_bits[q];
                midrev |= dq << (log_M - sh);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t k = __brev(rows[e]) >> (32 - LR);
            fp o = v[e];
            if (p.apply_scale) o = mont_mul(o, p.scale);
            out[(size_t)(colbase + c) + ((((size_t)k << log_M) + midrev) << p.log_r0)] = o;
        }
    }
}
template <int LR>
static hipError_t launch_pass_reg(const NttPassParams& p, size_t tiles, hipStream_t s) {
    const dim3 g((unsigned)tiles), b(4u << LR);
    static bool attr_set[64] = {};             // per device: the attribute belongs to the function on the current device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (LR == 8 && !attr_set[dev]) {           // only the 2^8-row tiles need more than the default 64 KiB
        const void* regs[] = {reinterpret_cast<const void*>(k_ntt_pass_reg<LR, false, false, false>), reinterpret_cast<const void*>(k_ntt_pass_reg<LR, false, false, true>),
                              reinterpret_cast<const void*>(k_ntt_pass_reg<LR, false, true, false>), reinterpret_cast<const void*>(k_ntt_pass_reg<LR, false, true, true>),
                              reinterpret_cast<const void*>(k_ntt_pass_reg<LR, true, false, false>)};
        for (const void* f : regs) {
            const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            if (e != hipSuccess) return e;
        }
        attr_set[dev] = true;
    }
    const size_t R = (size_t)1 << LR;
    if (p.is_final) hipLaunchKernelGGL((k_ntt_pass_reg<LR, true, false, false>), g, b, (R * 17 + R / 2) * sizeof(fp), s, p);
    else {
        const size_t l = (R * 16 + R / 2) * sizeof(fp);
        if (p.fuse_log_e) { if (p.tw_pass) hipLaunchKernelGGL((k_ntt_pass_reg<LR, false, true, true>), g, b, l, s, p); else hipLaunchKernelGGL((k_ntt_pass_reg<LR, false, true, false>), g, b, l, s, p); }
        else { if (p.tw_pass) hipLaunchKernelGGL((k_ntt_pass_reg<LR, false, false, true>), g, b, l, s, p); else hipLaunchKernelGGL((k_ntt_pass_reg<LR, false, false, false>), g, b, l, s, p); }
    }
    return hipSuccess;
}

// ---------------------------------------------------------------------------------------------------------------------
// k_ntt_pass_pipe — the same pass as a PERSISTENT workgroup with the next tile's loads in flight during this tile's levels.
// Why: with the butterflies taken out, the three passes of a 2^24 transform still take 0.30 ms (the tile fills and drains: 5.4 TB/s),
// with them 0.86 ms — the kernel above runs its memory phases and its arithmetic one after the other, and two workgroups per CU do
// not interleave them well enough (profiles/r03_ntt_phase_experiment.txt).  Here a CU holds ONE workgroup of 1024 threads, which may
// then use 128 VGPRs: 16 of them carry the thread's four elements of the NEXT tile, loaded right after the current tile has been
// parked in LDS and consumed when the current tile has been stored.  The w_R table is built once per workgroup, not per tile.
struct NttTileAddr {
    size_t base, row_stride_in, xform_in, xform_out, chunk;
    uint32_t colbase;
};
__device__ __forceinline__ NttTileAddr ntt_tile_addr(const NttPassParams& p, size_t t) {
    NttTileAddr a;
    const size_t tile = t & (((size_t)1 << p.log_tiles) - 1);
    const size_t xform = (t >> p.log_tiles) << p.log_n;
    a.xform_in = p.fuse_log_e ? xform >> p.fuse_log_e : xform;
    a.xform_out = xform;
    if (!p.is_final) {
        const uint32_t log_mp = p.log_m - p.log_r;
        const size_t chunks = (size_t)1 << (log_mp - p.log_c);
        const size_t block = tile / chunks;
        a.chunk = tile % chunks;
        a.colbase = (uint32_t)(a.chunk << p.log_c);
        a.base = (block << p.log_m) + a.colbase;
        a.row_stride_in = (size_t)1 << log_mp;
    } else {
        const size_t chunks = (size_t)1 << (p.log_r0 - p.log_c);
        const size_t mid = tile / chunks;
        a.chunk = tile % chunks;
        a.colbase = (uint32_t)(a.chunk << p.log_c);
        a.base = mid << p.log_r;
        a.row_stride_in = (size_t)1 << (p.log_n - p.log_r0);
    }
    return a;
}
#define NTT_PIPE_PER_THREAD ((1u << NTT_TILE_LOG) / NTT_THREADS)
__global__ __launch_bounds__(NTT_THREADS, 4) void k_ntt_pass_pipe(NttPassParams p, uint32_t n_tiles) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    fp* x = reinterpret_cast<fp*>(lds_raw);
    const uint32_t R = 1u << p.log_r, C = 1u << p.log_c, T = R * C;
    const uint32_t rs = (p.is_final && p.log_c) ? C + 1 : C;
    fp* twr = x + (size_t)R * rs;
    fp* pc = twr + (R >> 1);                                      // fused LDE: offset^(colbase + c), c < C
    const uint32_t tid = threadIdx.x;
    for (uint32_t e = tid; e < (R >> 1); e += blockDim.x) twr[e] = ntt_load_tw(p.tw, p.log_n, (size_t)e << (p.log_n - p.log_r));
    const uint32_t r_n = p.fuse_log_e ? R >> p.fuse_log_e : R;   // fused LDE: rows below it hold coefficients, the rest is zero padding

    // the thread's elements of tile tt -> registers (loads left in flight); T == NTT_THREADS · NTT_PIPE_PER_THREAD (launcher)
    fp nxt0, nxt1, nxt2, nxt3;
    static_assert(NTT_PIPE_PER_THREAD == 4, "four prefetch registers per thread");
#define NTT_FETCH_ONE(dst, q)                                                                                   \
    do {                                                                                                        \
        const uint32_t idx_ = tid + (q) * NTT_THREADS;                                                          \
        if (!p.is_final) {                                                                                      \
            const uint32_t r_ = idx_ >> p.log_c, c_ = idx_ & (C - 1);                                           \
            dst = r_ < r_n ? in_[fa.base + (size_t)r_ * fa.row_stride_in + c_] : fp_zero();                     \
        } else {                                                                                                \
            const uint32_t c_ = idx_ >> p.log_r, r_ = idx_ & (R - 1);                                           \
            dst = in_[(size_t)(fa.colbase + c_) * fa.row_stride_in + fa.base + r_];                             \
        }                                                                                                       \
    } while (0)
#define NTT_FETCH(tt)                                                                                           \
    do {                                                                                                        \
        const NttTileAddr fa = ntt_tile_addr(p, (tt));                                                          \
        const fp* __restrict__ in_ = p.in + fa.xform_in;                                                        \
        NTT_FETCH_ONE(nxt0, 0); NTT_FETCH_ONE(nxt1, 1); NTT_FETCH_ONE(nxt2, 2); NTT_FETCH_ONE(nxt3, 3);         \
    } while (0)
#define NTT_PARK_ONE(src, q)                                                                                    \
    do {                                                                                                        \
        const uint32_t idx_ = tid + (q) * NTT_THREADS;                                                          \
        if (!p.is_final) {                                                                                      \
            const uint32_t r_ = idx_ >> p.log_c, c_ = idx_ & (C - 1);                                           \
            fp v_ = src;                                                                                        \
            if (p.fuse_log_e && r_ < r_n) v_ = mont_mul(v_, mont_mul(p.pow_row[r_], pc[c_]));                   \
            x[r_ * rs + c_] = v_;                                                                               \
        } else {                                                                                                \
            const uint32_t c_ = idx_ >> p.log_r, r_ = idx_ & (R - 1);                                           \
            x[r_ * rs + c_] = src;                                                                              \
        }                                                                                                       \
    } while (0)
    size_t t = blockIdx.x;
    if (t < n_tiles) NTT_FETCH(t);
    for (; t < n_tiles; t += gridDim.x) {
        const NttTileAddr a = ntt_tile_addr(p, t);
        fp* __restrict__ out = p.out + a.xform_out;
        if (p.fuse_log_e) {
            if (tid < C) pc[tid] = mont_mul(p.pow_chunk[a.chunk], p.pow_col[tid]);
            __syncthreads();
        }
        // park the prefetched tile in LDS
        NTT_PARK_ONE(nxt0, 0); NTT_PARK_ONE(nxt1, 1); NTT_PARK_ONE(nxt2, 2); NTT_PARK_ONE(nxt3, 3);
        __syncthreads();
        if (t + gridDim.x < n_tiles) NTT_FETCH(t + gridDim.x);   // in flight during the levels below

        for (uint32_t hb = p.log_r; hb > 0;) {
            const uint32_t S = hb < NTT_RADIX_LOG ? hb : NTT_RADIX_LOG, lo = hb - S;
            if (lo) ntt_levels<NTT_RADIX_LOG, false>(x, twr, rs, p.log_r, p.log_c, lo, T, tid, blockDim.x);
            else switch (S) {
                case 1: ntt_levels<1, true>(x, twr, rs, p.log_r, p.log_c, 0, T, tid, blockDim.x); break;
                case 2: ntt_levels<2, true>(x, twr, rs, p.log_r, p.log_c, 0, T, tid, blockDim.x); break;
#if NTT_RADIX_LOG >= 3
                case 3: ntt_levels<3, true>(x, twr, rs, p.log_r, p.log_c, 0, T, tid, blockDim.x); break;
#endif
            }
            __syncthreads();
            hb = lo;
        }

        if (!p.is_final) {
            const uint32_t tw_shift = p.log_n - p.log_m;
            for (uint32_t idx = tid; idx < T; idx += blockDim.x) {
                const uint32_t r = idx >> p.log_c, c = idx & (C - 1);
                const uint32_t k = __brev(r) >> (32 - p.log_r);
                fp v = x[r * rs + c];
                const size_t e = ((size_t)(a.colbase + c) * k) << tw_shift;
                v = mont_mul(v, p.tw_pass ? p.tw_pass[(size_t)k * a.row_stride_in + a.colbase + c] : ntt_load_tw(p.tw, p.log_n, e));
                out[a.base + (size_t)k * a.row_stride_in + c] = v;
            }
        } else {
            const size_t tile = t & (((size_t)1 << p.log_tiles) - 1);
            const size_t chunks = (size_t)1 << (p.log_r0 - p.log_c);
            size_t mid = tile / chunks, midrev = 0;
            uint32_t log_M = 0;
            for (int q = (int)p.num_mid - 1; q >= 0; --q) { log_M += p.mid_bits[q]; }
            {
                uint32_t sh = 0;
                size_t rem = mid;
                for (int q = (int)p.num_mid - 1; q >= 0; --q) {
                    const size_t dq = rem & (((size_t)1 << p.mid_bits[q]) - 1);
                    rem >>= p.mid_bits[q];
                    sh += p.mid_bits[q];
                    midrev |= dq << (log_M - sh);
                }
            }
            for (uint32_t idx = tid; idx < T; idx += blockDim.x) {
                const uint32_t r = idx >> p.log_c, c = idx & (C - 1);
                const uint32_t k = p.log_r ? (__brev(r) >> (32 - p.log_r)) : 0;
                fp v = x[r * rs + c];
                if (p.apply_scale) v = mont_mul(v, p.scale);
                out[(size_t)(a.colbase + c) + ((((size_t)k << log_M) + midrev) << p.log_r0)] = v;
            }
        }
        __syncthreads();                                         // the tile has left LDS: the next one may be parked
    }
}

void ntt_make_plan(uint32_t log_n, NttPlan* plan, size_t batch) {
    plan->log_n = log_n;
    uint32_t P = log_n <= NTT_MAX_PASS_BITS ? 1 : (log_n + NTT_MAX_PASS_BITS - 1) / NTT_MAX_PASS_BITS;
    plan->num_passes = P;
    for (uint32_t i = 0; i < NTT_MAX_PASSES; ++i) plan->bits[i] = 0;
    uint32_t rem = log_n;
    for (uint32_t i = 0; i < P; ++i) {                 // spread evenly, larger digits first
        uint32_t b = (rem + (P - i) - 1) / (P - i);
        plan->bits[i] = b;
        rem -= b;
    }
    // Tile size, per pass.  2^12 elements (64 KiB, 1024 threads, two per CU) is the best shape when every CU has tiles to spare; a 2^18
    // transform has only 64 of them.  Halve the tile (and its workgroup) while fewer than 1024 tiles exist, as long as its rows
    // stay >= 16 elements (256 bytes) long: 2^18 0.046 -> 0.030 ms (profiles/r03_ntt_phase_experiment.txt §6).
    // STARK_HIP_NTT_TILE_LOG: one value for every pass, or a comma-separated value per pass (A/B runs).
    uint32_t tl = NTT_TILE_LOG;
    while (tl > 10 && tl - 1 >= plan->bits[0] + 4 && (((batch ? batch : 1) << log_n) >> tl) < 1024) --tl;
    // and no pass gets a tile wider than 16 columns x its rows: four 2^11 tiles per CU hide each other's waits better than two 2^12
    // ones (2^22: 0.204 -> 0.196 ms), what a tile must keep is the 256-byte row
    for (uint32_t i = 0; i < NTT_MAX_PASSES; ++i) plan->tile_log[i] = (i < P && plan->bits[i] + 4 < tl && plan->bits[i] + 4 >= 10) ? plan->bits[i] + 4 : tl;
    if (const char* e = getenv("STARK_HIP_NTT_TILE_LOG")) {
        uint32_t last = 0;
        for (uint32_t i = 0; i < P; ++i) {
            uint32_t v = (uint32_t)atoi(e);
            if (v == 0) v = last;
            last = v;
            if (v >= 10 && v <= NTT_TILE_LOG && v >= plan->bits[i] + 2) plan->tile_log[i] = v;
            const char* c = strchr(e, ',');
            if (c) e = c + 1;
        }
    }
}

static size_t pass_lds_bytes(uint32_t log_r, uint32_t log_c, bool is_final) {
    const size_t R = (size_t)1 << log_r, C = (size_t)1 << log_c;
#ifdef NTT_PAD_ALL
    const size_t rs = log_c ? C + 1 : C;
#else
    const size_t rs = (is_final && log_c) ? C + 1 : C;
#endif
    return (R * rs + (R >> 1) + 1 + C) * sizeof(fp);           // tile | w_R powers | C column powers of a fused LDE load
}

// out[k] = base^k in Montgomery form, k < count (base_mont = base·2^128)
__global__ __launch_bounds__(256) void k_pow_table(fp* __restrict__ out, uint32_t count, fp base_mont) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    fp acc = fp_mont_one(), b = base_mont;
    for (uint32_t e = k; e; e >>= 1) {
        if (e & 1) acc = mont_mul(acc, b);
        b = mont_mul(b, b);
    }
    out[k] = acc;
}
hipError_t launch_pow_table(fp* out, size_t count, fp base_mont, hipStream_t s) {
    if (!count) return hipSuccess;
    if (count > 0xffffffffu) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_pow_table, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, out, (uint32_t)count, base_mont);
    return hipGetLastError();
}
bool ntt_lde_fusable(const NttPlan& plan, uint32_t log_expansion, NttLdeShape* shape) {
    if (plan.num_passes < 2 || log_expansion == 0 || log_expansion > plan.bits[0]) return false;
    const uint32_t log_r = plan.bits[0], log_mp = plan.log_n - log_r;
    const uint32_t log_c = (plan.tile_log[0] - log_r) < log_mp ? (plan.tile_log[0] - log_r) : log_mp;
    shape->rows = (size_t)1 << (log_r - log_expansion);
    shape->chunks = (size_t)1 << (log_mp - log_c);
    shape->cols = (size_t)1 << log_c;
    shape->log_row_stride = log_mp;
    shape->log_cols = log_c;
    return true;
}

// out[(k << (log_m - log_r)) + col] = w_m^(col·k), k < 2^log_r, col < 2^(log_m - log_r): the inter-pass twiddles of the pass
// that works on sub-blocks of m = 2^log_m elements, in the order its tiles read them (col fastest: coalesced)
__global__ __launch_bounds__(256) void k_pass_twiddles(fp* __restrict__ out, const fp* __restrict__ tw, uint32_t log_n, uint32_t log_m, uint32_t log_r) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >> log_m) return;
    const uint32_t log_mp = log_m - log_r;
    const size_t k = i >> log_mp, col = i & (((size_t)1 << log_mp) - 1);
    out[i] = ntt_load_tw(tw, log_n, (col * k) << (log_n - log_m));
}
hipError_t launch_pass_twiddles(fp* out, const fp* tw, uint32_t log_n, uint32_t log_m, uint32_t log_r, hipStream_t s) {
    if (log_m > log_n || log_r > log_m || log_m > 30) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_pass_twiddles, dim3((unsigned)((((size_t)1 << log_m) + 255) / 256)), dim3(256), 0, s, out, tw, log_n, log_m, log_r);
    return hipGetLastError();
}

hipError_t launch_ntt(const NttPlan& plan, fp* data, fp* scratch, const fp* tw, fp scale_mont, bool apply_scale, size_t batch, hipStream_t s, const NttLdeFuse* fuse,
                      const NttPassTables* ptab) {
    static bool attr_set[64] = {};             // per device: the attribute belongs to the function on the current device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_ntt_pass), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_ntt_pass_pipe), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    const uint32_t P = plan.num_passes, log_n = plan.log_n;
    uint32_t consumed = 0;
    for (uint32_t pi = 0; pi < P; ++pi) {
        NttPassParams p{};
        p.tw = tw;
        p.log_n = log_n;
        p.log_r = plan.bits[pi];
        p.apply_scale = 0;
        p.scale = scale_mont;
        const bool is_final = (pi == P - 1);
        p.is_final = is_final;
        if (!is_final) {
            p.log_m = log_n - consumed;
            const uint32_t log_mp = p.log_m - p.log_r;
            p.log_c = (plan.tile_log[pi] - p.log_r) < log_mp ? (plan.tile_log[pi] - p.log_r) : log_mp;
            p.tw_pass = ptab ? ptab->t[pi] : nullptr;
            p.in = data;                                               // non-final passes work in place on `data` ...
            p.out = (pi + 2 == P) ? scratch : data;                    // ... except the last one, which lands in scratch
            if (pi == 0 && fuse) {                                     // coset LDE: the first pass reads the coefficients themselves
                p.in = fuse->coeffs; p.fuse_log_e = fuse->log_expansion;
                p.pow_row = fuse->pow_row; p.pow_chunk = fuse->pow_chunk; p.pow_col = fuse->pow_col;
            }
        } else {
            p.log_r0 = (P >= 2) ? plan.bits[0] : 0;
            p.log_c = (plan.tile_log[pi] - p.log_r) < p.log_r0 ? (plan.tile_log[pi] - p.log_r) : p.log_r0;
            p.num_mid = (P >= 2) ? P - 2 : 0;
            for (uint32_t q = 0; q < p.num_mid; ++q) p.mid_bits[q] = plan.bits[1 + q];
            p.apply_scale = apply_scale ? 1 : 0;
            p.in = (P >= 2) ? scratch : data;
            p.out = data;
        }
        p.log_tiles = log_n - (p.log_r + p.log_c);
        const size_t tiles = ((size_t)1 << p.log_tiles) * batch;
        if (tiles > 0x7fffffffu) return hipErrorInvalidValue;
        const size_t lds = pass_lds_bytes(p.log_r, p.log_c, is_final);
        // Persistent, pipelined form for passes with enough full-size tiles to keep every CU's one workgroup busy for several tiles.
        // Opt-in (STARK_HIP_NTT_PIPE=1): measured 0.862-0.872 ms against 0.822-0.827 ms for the plain form at 2^24 on one box —
        // hiding the fills does not pay for halving the waves per SIMD (profiles/r03_ntt_phase_experiment.txt).
        const char* pipe_env = getenv("STARK_HIP_NTT_PIPE");
        const int pipe_mode = pipe_env ? atoi(pipe_env) : 0;
        static const unsigned pipe_wgs = [] { const char* e = getenv("STARK_HIP_NTT_PIPE_WGS"); const int v = e ? atoi(e) : 0; return (unsigned)(v > 0 ? v : 256); }();
        // STARK_HIP_NTT_REG (read per call, A/B and tests): bit 0 = register-exchange form for non-final passes, bit 1 = for the final pass, bit 2 = for 2^6- and 2^7-row passes too
        const char* reg_env = getenv("STARK_HIP_NTT_REG");
        const int reg_mode = reg_env ? atoi(reg_env) : 7;
        // register-exchange form: 2^6 .. 2^8 rows x 16 columns
        const bool reg_shape = p.log_r >= 6 && p.log_r <= 8 && p.log_c == 4 && (p.log_r == 8 || (reg_mode & 4));
        if (reg_shape && (is_final ? ((reg_mode & 2) && P >= 2) : (reg_mode & 1) != 0)) {
            const hipError_t e = p.log_r == 8 ? launch_pass_reg<8>(p, tiles, s) : p.log_r == 7 ? launch_pass_reg<7>(p, tiles, s) : launch_pass_reg<6>(p, tiles, s);
            if (e != hipSuccess) return e;
        } else if (pipe_mode && (p.log_r + p.log_c) == NTT_TILE_LOG && tiles >= 4 * (size_t)pipe_wgs) {
            hipLaunchKernelGGL(k_ntt_pass_pipe, dim3(pipe_wgs), dim3(NTT_THREADS), lds, s, p, (uint32_t)tiles);
        } else {
#ifdef NTT_TRACE
        static unsigned long long* d_trace = nullptr;
        static int trace_left = getenv("STARK_HIP_NTT_TRACE") ? atoi(getenv("STARK_HIP_NTT_TRACE")) : 0;
        const bool tracing = trace_left > 0 && tiles >= 1024 && tiles <= 8192;
        if (tracing && !d_trace && hipMalloc(&d_trace, 8192 * 64) != hipSuccess) return hipErrorOutOfMemory;
        p.trace = tracing ? d_trace : nullptr;
        static const size_t lds_extra = getenv("STARK_HIP_NTT_LDS_EXTRA") ? (size_t)atoi(getenv("STARK_HIP_NTT_LDS_EXTRA")) : 0;    // e.g. 24576: one workgroup per CU
        hipLaunchKernelGGL(k_ntt_pass, dim3((unsigned)tiles), dim3(NTT_THREADS >> (NTT_TILE_LOG - plan.tile_log[pi])), lds + lds_extra, s, p);
#else
        hipLaunchKernelGGL(k_ntt_pass, dim3((unsigned)tiles), dim3(NTT_THREADS >> (NTT_TILE_LOG - plan.tile_log[pi])), lds, s, p);
#endif
#ifdef NTT_TRACE
        if (tracing) {
            --trace_left;
            std::vector<unsigned long long> h(tiles * 8);
            if (hipStreamSynchronize(s) != hipSuccess || hipMemcpy(h.data(), d_trace, tiles * 64, hipMemcpyDeviceToHost) != hipSuccess) return hipErrorUnknown;
            unsigned long long t_min = ~0ull, t_max = 0;
            for (size_t w = 0; w < tiles; ++w) { if (h[w * 8] < t_min) t_min = h[w * 8]; if (h[w * 8 + 6] > t_max) t_max = h[w * 8 + 6]; }
            double sum[7] = {0};
            for (size_t w = 0; w < tiles; ++w) for (int i = 1; i < 7; ++i) sum[i] += (double)(h[w * 8 + i] - h[w * 8 + i - 1]) / 100.0;
            fprintf(stderr, "[ntt trace] pass %u (%s, R=2^%u C=2^%u, %zu tiles): kernel %.1f us; per workgroup (us, mean): fill %.2f | groups %.2f %.2f %.2f %.2f | twiddle+drain %.2f | total %.2f\n", pi,
                    is_final ? "final" : "non-final", p.log_r, p.log_c, tiles, (double)(t_max - t_min) / 100.0, sum[1] / tiles, sum[2] / tiles, sum[3] / tiles, sum[4] / tiles, sum[5] / tiles,
                    sum[6] / tiles, (sum[1] + sum[2] + sum[3] + sum[4] + sum[5] + sum[6]) / tiles);
            // the first 6 workgroups that ran on the CU of workgroup 0: start / end offsets
            const unsigned long long cu0 = h[7];
            int shown = 0;
            for (size_t w = 0; w < tiles && shown < 10; ++w) if (h[w * 8 + 7] == cu0) {
                fprintf(stderr, "[ntt trace]   wg %5zu on that CU: start %7.2f filled %7.2f g %7.2f %7.2f %7.2f %7.2f end %7.2f\n", w, (double)(h[w * 8] - t_min) / 100.0, (double)(h[w * 8 + 1] - t_min) / 100.0,
                        (double)(h[w * 8 + 2] - t_min) / 100.0, (double)(h[w * 8 + 3] - t_min) / 100.0, (double)(h[w * 8 + 4] - t_min) / 100.0, (double)(h[w * 8 + 5] - t_min) / 100.0,
                        (double)(h[w * 8 + 6] - t_min) / 100.0);
                ++shown;
            }
        }
#endif
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        consumed += p.log_r;
    }
    return hipSuccess;
}

// out[j] = j < n ? coeffs[j]·offset^j : 0 ; each lane owns 4 consecutive j: one square-and-multiply
// for offset^j0, then three steps.
#define SCALE_PER_THREAD 4
__global__ __launch_bounds__(256) void k_coset_scale_pad(const fp* __restrict__ coeffs, size_t n, fp offset, fp* __restrict__ out, size_t N) {
    const size_t j0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * SCALE_PER_THREAD;
    if (j0 >= N) return;
    coeffs += (size_t)blockIdx.y * n;                                    // blockIdx.y: index in a batch of polynomials stored back to back
    out += (size_t)blockIdx.y * N;
    if (j0 >= n) {
#pragma unroll
        for (int q = 0; q < SCALE_PER_THREAD; ++q) if (j0 + q < N) out[j0 + q] = fp_zero();
        return;
    }
    fp step = fp_to_mont(offset), base = step, t = fp_mont_one();
    for (size_t e = j0; e; e >>= 1) {
        if (e & 1) t = mont_mul(t, base);
        base = mont_mul(base, base);
    }
#pragma unroll 1
    for (int q = 0; q < SCALE_PER_THREAD && j0 + q < N; ++q) {
        out[j0 + q] = (j0 + q < n) ? mont_mul(coeffs[j0 + q], t) : fp_zero();
        t = mont_mul(t, step);
    }
}

hipError_t launch_coset_scale_pad(const fp* coeffs, size_t n, fp offset, fp* out, size_t N, hipStream_t s, size_t batch) {
    if (!batch) return hipSuccess;
    if (batch > 65535) return hipErrorInvalidValue;
    const size_t threads = (N + SCALE_PER_THREAD - 1) / SCALE_PER_THREAD;
    hipLaunchKernelGGL(k_coset_scale_pad, dim3((unsigned)((threads + 255) / 256), (unsigned)batch), dim3(256), 0, s, coeffs, n, offset, out, N);
    return hipGetLastError();
}

// data[r][c] *= w_n^((row0 + r)·(col0 + c)) for a rows × 2^log_cols row-major block: the twiddle between the
// two transform steps of a four-step NTT whose steps run as separate batched transforms (dist_ntt.py).
__global__ __launch_bounds__(256) void k_twiddle_scale(fp* __restrict__ data, size_t total, uint32_t log_cols, uint64_t row0, uint64_t col0,
                                                       const fp* __restrict__ tw, uint32_t log_n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const uint64_t r = row0 + (i >> log_cols), c = col0 + (i & (((size_t)1 << log_cols) - 1));
    const uint64_t nmask = ((uint64_t)1 << log_n) - 1;
    const uint64_t e = ((r & nmask) * (c & nmask)) & nmask;      // r, c < 2^32: the product fits 64 bits
    data[i] = mont_mul(data[i], ntt_load_tw(tw, log_n, e));
}

hipError_t launch_twiddle_scale(fp* data, size_t rows, uint32_t log_cols, uint64_t row0, uint64_t col0, const fp* tw, uint32_t log_n, hipStream_t s) {
    const size_t total = rows << log_cols;
    if (!total) return hipSuccess;
    hipLaunchKernelGGL(k_twiddle_scale, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, data, total, log_cols, row0, col0, tw, log_n);
    return hipGetLastError();
}

// data[i] *= start·base^i (start in Montgomery form = base^j0): the coset pre-scale of a slice [j0, j0+count) of a
// coefficient vector that is spread over ranks.
__global__ __launch_bounds__(256) void k_scale_powers(fp* __restrict__ data, size_t count, fp start_mont, fp base) {
    const size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * SCALE_PER_THREAD;
    if (i0 >= count) return;
    fp step = fp_to_mont(base), b = step, t = start_mont;
    for (size_t e = i0; e; e >>= 1) {
        if (e & 1) t = mont_mul(t, b);
        b = mont_mul(b, b);
    }
#pragma unroll 1
    for (int q = 0; q < SCALE_PER_THREAD && i0 + q < count; ++q) {
        data[i0 + q] = mont_mul(data[i0 + q], t);
        t = mont_mul(t, step);
    }
}

hipError_t launch_scale_powers(fp* data, size_t count, fp start_mont, fp base, hipStream_t s) {
    const size_t threads = (count + SCALE_PER_THREAD - 1) / SCALE_PER_THREAD;
    if (!threads) return hipSuccess;
    hipLaunchKernelGGL(k_scale_powers, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, data, count, start_mont, base);
    return hipGetLastError();
}
