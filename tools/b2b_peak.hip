// b2b_peak.hip — ceiling of the compiled Blake2b compression code on gfx950: pure compression loops, no tree
// bookkeeping, negligible memory traffic.  Build: hipcc --offload-arch=gfx950 -O3 -Istark_anatomy_amd/csrc tools/b2b_peak.hip -o tools/b2b_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "blake2b.cuh"

template <int MODE, int WAVES>
__global__ __launch_bounds__(256, WAVES) void k(uint64_t* out, int iters, uint64_t seed) {
    const uint64_t t = blockIdx.x * 256 + threadIdx.x;
    digest64 acc;
    const uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < 8; ++i) acc.h[i] = seed * (i + 1) + t;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {                       // leaf: 16-byte message
            digest64 d = b2b_leaf(acc.h[0] + it, acc.h[1]);
            for (int i = 0; i < 8; ++i) acc.h[i] ^= d.h[i];
        } else {                               // node: chained
            digest64 r;
            for (int i = 0; i < 8; ++i) r.h[i] = seed + i + it;
            acc = b2b_node(acc, r);
        }
    }
    uint64_t x = 0;
    for (int i = 0; i < 8; ++i) x ^= acc.h[i];
    out[t] = x;
    if (threadIdx.x == 0 && (blockIdx.x & 255) == 0) {          // diagnostic build only: in-kernel clock = dcycles / drealtime * 100 MHz
        out[(size_t)8192 * 256 + 2 * (blockIdx.x >> 8)] = __builtin_amdgcn_s_memtime() - c0;
        out[(size_t)8192 * 256 + 2 * (blockIdx.x >> 8) + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}
template <int MODE, int WAVES>
static void run(const char* name, uint64_t* buf, int blocks, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, WAVES>), dim3(blocks), dim3(256), 0, 0, buf, iters, 12345ull);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, WAVES>), dim3(blocks), dim3(256), 0, 0, buf, iters, 12345ull);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    uint64_t st[2]; (void)hipMemcpy(st, buf + (size_t)8192 * 256 + 2 * ((blocks >> 8) - 1), 16, hipMemcpyDeviceToHost);
    printf("%-20s blocks=%5d iters=%4d  %8.3f ms  %7.2f G compress/s  in-kernel clock %.2f GHz\n", name, blocks, iters, ms, (double)blocks * 256 * iters / ms / 1e6, (double)st[0] / (double)st[1] * 0.1);
}
int main() {
    uint64_t* buf; (void)hipMalloc(&buf, (size_t)8192 * 256 * 8 + 4096);
    for (int iters : {4, 16, 128, 1024}) {
        run<0, 1>("leaf (no bound)", buf, 4096, iters);
        run<1, 1>("node (no bound)", buf, 4096, iters);
        run<1, 1>("node (no bound)", buf, 1024, iters);
    }
    return 0;
}
