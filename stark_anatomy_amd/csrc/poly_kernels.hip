// poly_kernels.hip — the polynomial layer's own kernels.  The reference crate has no polynomial code
// (SURVEY.md §0); the operations are those a STARK prover needs next to the FRI commit: products through the NTT, the
// quotient by a zerofier X^n − c on an evaluation coset and in coefficient form.  Inversions go through the batched
// kernel in field_kernels.hip.
#include "kernels.h"

__global__ __launch_bounds__(256) void k_zerofier_values(fp* __restrict__ z, size_t count, fp start, fp w_mont, fp c) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    fp acc = start, base = w_mont;                       // start · w^k: canonical × Montgomery powers stays canonical
    for (size_t e = k; e; e >>= 1) {
        if (e & 1) acc = mont_mul(acc, base);
        base = mont_mul(base, base);
    }
    z[k] = fp_sub(acc, c);
}

__global__ __launch_bounds__(256) void k_mul_periodic(const fp* __restrict__ in, const fp* __restrict__ factor, size_t mask, fp* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = mont_mul(fp_reduce_once(in[i]), factor[i & mask]);
}

// One lane per residue class t mod n walks its coefficients from the top: q_{k−n} = f_k + c·q_k, r_t = f_t + c·q_t.
// Neighbouring lanes touch neighbouring coefficients, so every step of the walk is a coalesced row of n elements.
__global__ __launch_bounds__(256) void k_divmod_zerofier(const fp* __restrict__ f, size_t nf, size_t n, fp c_mont, fp* __restrict__ q, fp* __restrict__ r) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    if (t >= nf) { r[t] = fp_zero(); return; }
    fp qk = fp_zero();
    size_t k = t + ((nf - 1 - t) / n) * n;
    for (; k >= n; k -= n) {
        qk = fp_add(fp_reduce_once(f[k]), mont_mul(qk, c_mont));
        q[k - n] = qk;
    }
    r[t] = fp_add(fp_reduce_once(f[t]), mont_mul(qk, c_mont));
}

hipError_t launch_zerofier_values(fp* z, size_t count, fp start, fp w_mont, fp c, hipStream_t s) {
    if (!count) return hipSuccess;
    hipLaunchKernelGGL(k_zerofier_values, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, z, count, start, w_mont, c);
    return hipGetLastError();
}
hipError_t launch_mul_periodic(const fp* in, const fp* factor, size_t mask, fp* out, size_t n, hipStream_t s) {
    if (!n) return hipSuccess;
    size_t blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_mul_periodic, dim3((unsigned)blocks), dim3(256), 0, s, in, factor, mask, out, n);
    return hipGetLastError();
}
hipError_t launch_divmod_zerofier(const fp* f, size_t nf, size_t n, fp c_mont, fp* q, fp* r, hipStream_t s) {
    if (!n) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_divmod_zerofier, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, f, nf, n, c_mont, q, r);
    return hipGetLastError();
}
