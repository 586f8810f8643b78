// mm_bench.hip — mont_mul variants on gfx950: bit-equality on random and edge operands, and throughput (full chip,
// four independent chains per lane).  build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I stark_anatomy_amd/csrc tools/mm_bench.hip -o tools/mm_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "fp128.cuh"

#if defined(__HIP_DEVICE_COMPILE__)
template <int V> __device__ __forceinline__ fp mm(fp a, fp b) { return V == 0 ? mont_mul_c(a, b) : V == 1 ? mont_mul_dev_eo(a, b) : mont_mul(a, b); }
#else
template <int V> __device__ __forceinline__ fp mm(fp a, fp b) { return mont_mul_c(a, b); }
#endif

template <int V>
__global__ __launch_bounds__(256) void k_chain(const fp* __restrict__ in, fp* __restrict__ out, int iters) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    fp x0 = in[4 * i], x1 = in[4 * i + 1], x2 = in[4 * i + 2], x3 = in[4 * i + 3];
    const fp y0 = fp_reduce_once(x1), y1 = fp_reduce_once(x2), y2 = fp_reduce_once(x3), y3 = fp_reduce_once(x0);
    for (int k = 0; k < iters; ++k) { x0 = mm<V>(x0, y0); x1 = mm<V>(x1, y1); x2 = mm<V>(x2, y2); x3 = mm<V>(x3, y3); }
    out[4 * i] = x0; out[4 * i + 1] = x1; out[4 * i + 2] = x2; out[4 * i + 3] = x3;
}
template <int V>
__global__ void k_pairs(const fp* __restrict__ a, const fp* __restrict__ b, fp* __restrict__ o, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = mm<V>(a[i], b[i]);
}
__global__ void k_bfly(const fp* __restrict__ a, const fp* __restrict__ b, fp* __restrict__ o, size_t n) {     // o[i].w[0], w[1]: 0 where fp_butterfly == (fp_add, fp_sub)
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const fp u = fp_reduce_once(a[i]), t = b[i];
    fp s, d;
    fp_butterfly(u, t, s, d);
    const fp s2 = fp_add(u, t), d2 = fp_sub(u, t);
    o[i] = fp_make(fp_eq(s, s2) ? 0u : 1u, fp_eq(d, d2) ? 0u : 1u, 0, 0);
}
// radix-4 steps as in ntt_kernels.hip (butterflies straight into Montgomery multiplies, the register <-> lane exchange in between, 8 waves
// per SIMD), with the hand-scheduled butterfly (BF = 1) or fp_add / fp_sub (BF = 0): the two must agree bit for bit under full load too
template <int BF>
__global__ __launch_bounds__(1024, 8) void k_radix(const fp* __restrict__ in, const fp* __restrict__ tw, fp* __restrict__ out, int iters) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    fp v[4], w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = fp_reduce_once(in[4 * i + e]); w[e] = tw[(4 * i + e) & 4095]; }
    for (int k = 0; k < iters; ++k) {
#pragma unroll
        for (int s = 1; s >= 0; --s)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (e & (1 << s)) continue;
                const fp u = v[e], t = v[e | (1 << s)];
                fp d;
                if (BF) fp_butterfly(u, t, v[e], d); else { v[e] = fp_add(u, t); d = fp_sub(u, t); }
                if ((k & 1) || (e & ((1 << s) - 1)) != 0) d = mont_mul(d, w[(e + s + k) & 3]);
                v[e | (1 << s)] = d;
            }
        if (k & 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int e0 = 0; e0 < 2; ++e0) { auto r = __builtin_amdgcn_permlane32_swap(v[e0].w[q], v[e0 | 2].w[q], false, false); v[e0].w[q] = r[0]; v[e0 | 2].w[q] = r[1]; }
#pragma unroll
                for (int e1 = 0; e1 < 4; e1 += 2) { auto r = __builtin_amdgcn_permlane16_swap(v[e1].w[q], v[e1 | 1].w[q], false, false); v[e1].w[q] = r[0]; v[e1 | 1].w[q] = r[1]; }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) out[4 * i + e] = v[e];
}
static uint64_t rs = 88172645463325252ULL;
static uint64_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }
int main() {
    const size_t n = 1 << 22;
    std::vector<fp> a(n), b(n), o0(n), o1(n);
    const uint32_t edge[] = {0u, 1u, 2u, 0xffffffffu, 0xfffffffeu, 0x80000000u, 0x7fffffffu, 0xcb800000u, 0xcb7fffffu, 0x347fffffu, 0x00800000u, 0x197u};
    for (size_t i = 0; i < n; ++i) {
        for (int w = 0; w < 4; ++w) {
            a[i].w[w] = (i & 1) ? edge[rnd() % 12] : (uint32_t)rnd();
            b[i].w[w] = (i & 2) ? edge[rnd() % 12] : (uint32_t)rnd();
        }
        b[i] = fp_reduce_once(b[i]);                       // second operand < p; the first may be any 128-bit value
    }
    fp *da, *db, *d0, *d1;
    hipMalloc(&da, n * 16); hipMalloc(&db, n * 16); hipMalloc(&d0, n * 16); hipMalloc(&d1, n * 16);
    hipMemcpy(da, a.data(), n * 16, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_pairs<0>, dim3(n / 256), dim3(256), 0, 0, da, db, d0, n);
    fp* d2; hipMalloc(&d2, n * 16);
    std::vector<fp> o2(n);
    hipLaunchKernelGGL(k_pairs<1>, dim3(n / 256), dim3(256), 0, 0, da, db, d1, n);
    hipLaunchKernelGGL(k_pairs<2>, dim3(n / 256), dim3(256), 0, 0, da, db, d2, n);
    hipMemcpy(o2.data(), d2, n * 16, hipMemcpyDeviceToHost);
    hipMemcpy(o0.data(), d0, n * 16, hipMemcpyDeviceToHost); hipMemcpy(o1.data(), d1, n * 16, hipMemcpyDeviceToHost);
    size_t bad_bf = 0;
    {
        hipLaunchKernelGGL(k_bfly, dim3(n / 256), dim3(256), 0, 0, da, db, d2, n);
        hipMemcpy(o2.data(), d2, n * 16, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < n; ++i) bad_bf += o2[i].w[0] + o2[i].w[1];
        printf("butterfly: %zu of %zu (sum, difference) pairs differ from fp_add / fp_sub\n", bad_bf, 2 * n);
        const size_t lanes_r = (size_t)1024 * 1024;               // 4 elements per lane = n
        hipLaunchKernelGGL(k_radix<0>, dim3(lanes_r / 1024), dim3(1024), 0, 0, da, db, d0, 64);
        hipLaunchKernelGGL(k_radix<1>, dim3(lanes_r / 1024), dim3(1024), 0, 0, da, db, d1, 64);
        hipMemcpy(o0.data(), d0, n * 16, hipMemcpyDeviceToHost); hipMemcpy(o1.data(), d1, n * 16, hipMemcpyDeviceToHost);
        size_t bad_r = 0;
        for (size_t i = 0; i < n; ++i) if (!fp_eq(o0[i], o1[i])) ++bad_r;
        printf("radix-4 chains under load (64 steps, lane swaps, 8 waves per SIMD): %zu of %zu outputs differ between fp_butterfly and fp_add / fp_sub\n", bad_r, n);
        bad_bf += bad_r;
        hipLaunchKernelGGL(k_pairs<0>, dim3(n / 256), dim3(256), 0, 0, da, db, d0, n);
        hipLaunchKernelGGL(k_pairs<1>, dim3(n / 256), dim3(256), 0, 0, da, db, d1, n);
        hipLaunchKernelGGL(k_pairs<2>, dim3(n / 256), dim3(256), 0, 0, da, db, d2, n);
        hipMemcpy(o0.data(), d0, n * 16, hipMemcpyDeviceToHost); hipMemcpy(o1.data(), d1, n * 16, hipMemcpyDeviceToHost); hipMemcpy(o2.data(), d2, n * 16, hipMemcpyDeviceToHost);
    }
    size_t bad = 0, bad_host = 0;
    for (size_t i = 0; i < n; ++i) {
        if (!fp_eq(o0[i], o1[i]) || !fp_eq(o0[i], o2[i])) ++bad;
        if (i < (1 << 16) && !fp_eq(o0[i], mont_mul_c(a[i], b[i]))) ++bad_host;
    }
    printf("pairs: %zu of %zu differ between the C form and the two device forms; %zu of 65536 differ from the host\n", bad, n, bad_host);
    const int iters = 256;
    const size_t lanes = (size_t)256 * 8 * 256;             // 8 workgroups of 256 per CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int v = 0; v < 3; ++v) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            if (v == 0) hipLaunchKernelGGL(k_chain<0>, dim3(lanes / 256), dim3(256), 0, 0, da, d0, iters);
            else if (v == 1) hipLaunchKernelGGL(k_chain<1>, dim3(lanes / 256), dim3(256), 0, 0, da, d1, iters);
            else hipLaunchKernelGGL(k_chain<2>, dim3(lanes / 256), dim3(256), 0, 0, da, d1, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%-28s %8.3f ms  %7.1f G mont_mul/s\n", v == 0 ? "mont_mul_c (plain C)" : v == 1 ? "even/odd columns (71 VALU)" : "product scanning (61 VALU)", best, (double)lanes * 4 * iters / best / 1e6);
    }
    hipMemcpy(o0.data(), d0, lanes * 4 * 16, hipMemcpyDeviceToHost); hipMemcpy(o1.data(), d1, lanes * 4 * 16, hipMemcpyDeviceToHost);
    bad = 0;
    for (size_t i = 0; i < lanes * 4; ++i) if (!fp_eq(o0[i], o1[i])) ++bad;
    printf("chains: %zu of %zu differ\n", bad, lanes * 4);
    return bad || bad_host || bad_bf ? 1 : 0;
}
