// field_kernels.hip — batched FieldElement operators (field.rs:57-114) on gfx950.
// One element per lane, one dwordx4 load per operand and one dwordx4 store: HBM-bound for
// add/sub/mul (48 B per element), VALU-bound for div/inv (Fermat, a^(p-2)).
#include "kernels.h"

template <int OP>
__global__ __launch_bounds__(256) void k_fp_binop(const fp* __restrict__ a, const fp* __restrict__ b, fp* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        fp x = a[i], r;
        if (OP == FPOP_ADD) r = fp_add(x, b[i]);
        else if (OP == FPOP_SUB) r = fp_sub(x, b[i]);
        else if (OP == FPOP_MUL) r = fp_mul(x, b[i]);
        else r = fp_reduce_once(x);
        out[i] = r;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Batched inversion by Montgomery's trick: FPOP_INV out = 1/a, FPOP_DIV out = a/b (field.rs:97-114 per
// element), FPOP_INV_MONT out = R/a (a factor for mont_mul).  A zero divisor raises *flag (field.rs:101-103) and gets 0;
// the other elements of the batch are unaffected.
// A workgroup inverts 256·T divisors with ONE field inversion: each lane multiplies up its T divisors (strided by 256, so
// loads coalesce), the 256 lane products go up a product tree in LDS, lane 0 inverts the root, the way down the tree hands
// every lane the inverse of its own product (inverse of the root · product of all other lanes), and the lane walks its T
// divisors back.  3 multiplies per element (+1 for the quotient) + ~2/T for the tree + 160/(256·T) for the inversion,
// where the per-element Fermat ladder took ~190.
// Raw canonical values are read as Montgomery forms (of x/R): mont_mul then needs no conversion on the way in, and the
// leftover power of R is taken out of the ONE inverted value per workgroup.
// MODE 0: self-contained, as described.  The one inversion per chunk is a single lane walking ~160 dependent multiplies
// (~21 us) while the other 255 wait, which is what a large call spends its time on; so from 2^22 elements up the work is
// split (api.hip: dev_fp_op): MODE 1 writes only each chunk's product to ext[chunk]; those few thousand products are
// inverted by one small MODE 0 launch; MODE 2 is MODE 0 with the chunk's inverted product read from ext[chunk] instead of
// computed.  One more pass of lane products and a tree, and no lane ever waits for an inversion.
#define BINV_T 8
template <int OP, int MODE>
__global__ __launch_bounds__(256) void k_fp_batch_inv(const fp* __restrict__ a, const fp* __restrict__ b, fp* __restrict__ out, size_t n, uint32_t* __restrict__ flag,
                                                      fp* __restrict__ ext) {
    __shared__ fp s_pre[2][256], s_suf[2][256];
    const uint32_t tid = threadIdx.x;
    const fp* __restrict__ div = (OP == FPOP_DIV) ? b : a;
    for (size_t base = (size_t)blockIdx.x * (256 * BINV_T); base < n; base += (size_t)gridDim.x * (256 * BINV_T)) {
        const size_t chunk = base / (256 * BINV_T);
        fp x[BINV_T], p[BINV_T];
        uint32_t zero_mask = 0;
        fp run = fp_mont_one();
#pragma unroll
        for (int j = 0; j < BINV_T; ++j) {
            const size_t i = base + (size_t)j * 256 + tid;
            fp v = fp_mont_one();                                         // beyond n, or a zero divisor: a factor of one
            if (i < n) {
                v = fp_reduce_once(div[i]);
                if (fp_is_zero(v)) { zero_mask |= 1u << j; v = fp_mont_one(); }
            }
            x[j] = v;
            p[j] = run;                                                   // product of this lane's divisors before j
            run = mont_mul(run, v);
        }
        if (zero_mask) *flag = 1u;
        if (MODE == 1) {                                                  // the chunk's product, nothing else
            s_pre[0][tid] = run;
            __syncthreads();
#pragma unroll 1
            for (uint32_t off = 128; off; off >>= 1) {
                if (tid < off) s_pre[0][tid] = mont_mul(s_pre[0][tid], s_pre[0][tid + off]);
                __syncthreads();
            }
            if (tid == 0) ext[chunk] = s_pre[0][0];
            __syncthreads();
            continue;
        }
        // Product tree over the 256 lane products (heap layout: node i has children 2i, 2i+1, leaves at 256 + lane): up to the
        // chunk's product at the root, then down again with out[child] = out[parent] · sibling, seeded with the inverse of the
        // root — a leaf ends with inverse(root) · (product of all other lanes) = the inverse of its own lane product.  One
        // thread per node, so a level of <= 64 nodes costs one wave one multiply: 21 wave-multiplies a chunk, against the 72 of
        // a two-ended Hillis-Steele scan over all 256 lanes (which this kernel used first).
        fp* const tr = &s_pre[0][0];                                      // 512 nodes each (s_pre and s_suf are 2 x 256)
        fp* const dn = &s_suf[0][0];
        tr[256 + tid] = run;
        __syncthreads();
#pragma unroll 1
        for (uint32_t size = 128; size; size >>= 1) {
            if (tid < size) tr[size + tid] = mont_mul(tr[2 * (size + tid)], tr[2 * (size + tid) + 1]);
            __syncthreads();
        }
        if (tid == 0) {
            fp t;
            if (MODE == 2) t = ext[chunk];
            else {
                t = mont_inv(tr[1]);                                      // never zero: zeros were replaced by one
                t = fp_from_mont(t);                                      // R/x per element ...
                if (OP == FPOP_INV) t = fp_from_mont(t);                  // ... or 1/x
            }
            dn[1] = t;
        }
        __syncthreads();
#pragma unroll 1
        for (uint32_t size = 2; size <= 256; size <<= 1) {               // `size` nodes on this level, one thread each
            if (tid < size) dn[size + tid] = mont_mul(dn[(size + tid) >> 1], tr[(size + tid) ^ 1]);
            __syncthreads();
        }
        fp inv = dn[256 + tid];
        __syncthreads();                                                  // the tree buffers are reused by the next chunk
#pragma unroll
        for (int j = BINV_T - 1; j >= 0; --j) {
            const size_t i = base + (size_t)j * 256 + tid;
            fp r = j ? mont_mul(inv, p[j]) : inv;                         // p[0] is one
            inv = mont_mul(inv, x[j]);
            if (i < n) {
                if (zero_mask & (1u << j)) r = fp_zero();
                else if (OP == FPOP_DIV) r = mont_mul(fp_reduce_once(a[i]), r);
                out[i] = r;
            }
        }
    }
}

hipError_t launch_fp_binop(int op, const fp* a, const fp* b, fp* out, size_t n, uint32_t* d_flag, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const int block = 256;
    if (op == FPOP_DIV || op == FPOP_INV || op == FPOP_INV_MONT) {
        if (!d_flag) return hipErrorInvalidValue;
        size_t chunks = (n + 256 * BINV_T - 1) / (256 * BINV_T);
        if (chunks > 256 * 8) chunks = 256 * 8;
        dim3 g((unsigned)chunks), t(block);
        if (op == FPOP_DIV) hipLaunchKernelGGL((k_fp_batch_inv<FPOP_DIV, 0>), g, t, 0, s, a, b, out, n, d_flag, (fp*)nullptr);
        else if (op == FPOP_INV) hipLaunchKernelGGL((k_fp_batch_inv<FPOP_INV, 0>), g, t, 0, s, a, b, out, n, d_flag, (fp*)nullptr);
        else hipLaunchKernelGGL((k_fp_batch_inv<FPOP_INV_MONT, 0>), g, t, 0, s, a, b, out, n, d_flag, (fp*)nullptr);
        return hipGetLastError();
    }
    size_t blocks = (n + block - 1) / block;
    if (blocks > 256 * 8) blocks = 256 * 8;
    dim3 g((unsigned)blocks), t(block);
    switch (op) {
    case FPOP_ADD: hipLaunchKernelGGL(k_fp_binop<FPOP_ADD>, g, t, 0, s, a, b, out, n); break;
    case FPOP_SUB: hipLaunchKernelGGL(k_fp_binop<FPOP_SUB>, g, t, 0, s, a, b, out, n); break;
    case FPOP_MUL: hipLaunchKernelGGL(k_fp_binop<FPOP_MUL>, g, t, 0, s, a, b, out, n); break;
    case FPOP_NEW: hipLaunchKernelGGL(k_fp_binop<FPOP_NEW>, g, t, 0, s, a, b, out, n); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// The split form of a batched inversion (see k_fp_batch_inv): chunk products -> ext[chunk]; caller inverts ext (FPOP_INV for
// op == FPOP_INV, else FPOP_INV_MONT) with launch_fp_binop; finish with the inverted products.
size_t fp_batch_inv_chunks(size_t n) { return (n + 256 * BINV_T - 1) / (256 * BINV_T); }
hipError_t launch_fp_batch_inv_products(int op, const fp* a, const fp* b, size_t n, fp* ext, uint32_t* d_flag, hipStream_t s) {
    if (!n) return hipSuccess;
    if (!ext || !d_flag || (op != FPOP_DIV && op != FPOP_INV && op != FPOP_INV_MONT)) return hipErrorInvalidValue;
    size_t chunks = fp_batch_inv_chunks(n);
    if (chunks > 256 * 16) chunks = 256 * 16;
    if (op == FPOP_DIV) hipLaunchKernelGGL((k_fp_batch_inv<FPOP_DIV, 1>), dim3((unsigned)chunks), dim3(256), 0, s, a, b, (fp*)nullptr, n, d_flag, ext);
    else hipLaunchKernelGGL((k_fp_batch_inv<FPOP_INV, 1>), dim3((unsigned)chunks), dim3(256), 0, s, a, b, (fp*)nullptr, n, d_flag, ext);
    return hipGetLastError();
}
hipError_t launch_fp_batch_inv_finish(int op, const fp* a, const fp* b, fp* out, size_t n, fp* ext, uint32_t* d_flag, hipStream_t s) {
    if (!n) return hipSuccess;
    if (!ext || !d_flag) return hipErrorInvalidValue;
    size_t chunks = fp_batch_inv_chunks(n);
    if (chunks > 256 * 16) chunks = 256 * 16;
    const dim3 g((unsigned)chunks), t(256);
    if (op == FPOP_DIV) hipLaunchKernelGGL((k_fp_batch_inv<FPOP_DIV, 2>), g, t, 0, s, a, b, out, n, d_flag, ext);
    else if (op == FPOP_INV) hipLaunchKernelGGL((k_fp_batch_inv<FPOP_INV, 2>), g, t, 0, s, a, b, out, n, d_flag, ext);
    else if (op == FPOP_INV_MONT) hipLaunchKernelGGL((k_fp_batch_inv<FPOP_INV_MONT, 2>), g, t, 0, s, a, b, out, n, d_flag, ext);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
