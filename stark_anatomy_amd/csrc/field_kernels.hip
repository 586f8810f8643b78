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
// loads coalesce), the 256 lane products are scanned from both ends in LDS, lane 0 inverts the workgroup product, every
// lane gets the inverse of its own product as inv_total · (product before it) · (product after it) and walks its T
// divisors back.  3 multiplies per element (+1 for the quotient) + 18/T for the scans + 160/(256·T) for the inversion,
// where the per-element Fermat ladder took ~190.
// Raw canonical values are read as Montgomery forms (of x/R): mont_mul then needs no conversion on the way in, and the
// leftover power of R is taken out of the ONE inverted value per workgroup.
#define BINV_T 8
template <int OP>
__global__ __launch_bounds__(256) void k_fp_batch_inv(const fp* __restrict__ a, const fp* __restrict__ b, fp* __restrict__ out, size_t n, uint32_t* __restrict__ flag) {
    __shared__ fp s_pre[2][256], s_suf[2][256];
    __shared__ fp s_inv;
    const uint32_t tid = threadIdx.x;
    const fp* __restrict__ div = (OP == FPOP_DIV) ? b : a;
    for (size_t base = (size_t)blockIdx.x * (256 * BINV_T); base < n; base += (size_t)gridDim.x * (256 * BINV_T)) {
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
        // inclusive scans of the lane products from the front (s_pre) and from the back (s_suf), ping-pong buffers
        s_pre[0][tid] = run;
        s_suf[0][tid] = run;
        __syncthreads();
        int cur = 0;
#pragma unroll 1
        for (uint32_t off = 1; off < 256; off <<= 1) {
            fp f = s_pre[cur][tid], g = s_suf[cur][tid];
            if (tid >= off) f = mont_mul(s_pre[cur][tid - off], f);
            if (tid + off < 256) g = mont_mul(g, s_suf[cur][tid + off]);
            s_pre[cur ^ 1][tid] = f;
            s_suf[cur ^ 1][tid] = g;
            __syncthreads();
            cur ^= 1;
        }
        if (tid == 0) {
            fp t = mont_inv(s_pre[cur][255]);                             // never zero: zeros were replaced by one
            t = fp_from_mont(t);                                          // R/x per element ...
            if (OP == FPOP_INV) t = fp_from_mont(t);                      // ... or 1/x
            s_inv = t;
        }
        __syncthreads();
        fp inv = s_inv;
        if (tid > 0) inv = mont_mul(inv, s_pre[cur][tid - 1]);
        if (tid < 255) inv = mont_mul(inv, s_suf[cur][tid + 1]);
        __syncthreads();                                                  // the scan buffers are reused by the next chunk
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
        if (op == FPOP_DIV) hipLaunchKernelGGL(k_fp_batch_inv<FPOP_DIV>, g, t, 0, s, a, b, out, n, d_flag);
        else if (op == FPOP_INV) hipLaunchKernelGGL(k_fp_batch_inv<FPOP_INV>, g, t, 0, s, a, b, out, n, d_flag);
        else hipLaunchKernelGGL(k_fp_batch_inv<FPOP_INV_MONT>, g, t, 0, s, a, b, out, n, d_flag);
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
