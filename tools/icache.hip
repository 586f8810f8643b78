// icache.hip — does straight-line code size (instruction-cache footprint) change VALU issue rate on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define X(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define A(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 7\n"
#define M8 X(0) A(1) X(2) X(3) A(4) X(5) X(6) A(7)          /* 5 VOP2 + 3 VOP3 = 44 bytes */
#define M64 M8 M8 M8 M8 M8 M8 M8 M8
#define M512 M64 M64 M64 M64 M64 M64 M64 M64
#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)
template <int SZ>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19, b = seed | 1;
    for (int it = 0; it < iters; ++it) {
        if (SZ == 64) asm volatile(M64 OPS);
        if (SZ == 512) asm volatile(M512 OPS);
        if (SZ == 2048) { asm volatile(M512 OPS); asm volatile(M512 OPS); asm volatile(M512 OPS); asm volatile(M512 OPS); }
        if (SZ == 4096) { asm volatile(M512 OPS); asm volatile(M512 OPS); asm volatile(M512 OPS); asm volatile(M512 OPS); asm volatile(M512 OPS); asm volatile(M512 OPS); asm volatile(M512 OPS); asm volatile(M512 OPS); }
        if (SZ == 8192) { for (int r = 0; r < 1; ++r) { asm volatile(M512 OPS); asm volatile(M512 OPS); asm volatile(M512 OPS); asm volatile(M512 OPS); asm volatile(M512 OPS); asm volatile(M512 OPS); asm volatile(M512 OPS); asm volatile(M512 OPS);
                                                          asm volatile(M512 OPS); asm volatile(M512 OPS); asm volatile(M512 OPS); asm volatile(M512 OPS); asm volatile(M512 OPS); asm volatile(M512 OPS); asm volatile(M512 OPS); asm volatile(M512 OPS); } }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
template <int SZ>
static void run(uint32_t* buf, int wps) {
    const int total = 1 << 18, iters = total / SZ, blocks = 256 * wps;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SZ>), dim3(blocks), dim3(256), 0, 0, buf, iters, 7u);
    (void)hipDeviceSynchronize(); (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<SZ>), dim3(blocks), dim3(256), 0, 0, buf, iters, 7u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)total * 44 / 8;
    printf("body=%5d instr (%6.1f KB) waves/SIMD=%d  %7.3f ms  %5.3f cycles/byte/SIMD @2.4GHz\n", SZ, SZ * 5.5 / 1024, wps, ms, ms * 1e-3 * 2.4e9 / (bytes * wps));
}
int main() {
    uint32_t* buf; (void)hipMalloc(&buf, (size_t)2048 * 256 * 4);
    for (int wps : {4, 6, 8}) { run<64>(buf, wps); run<512>(buf, wps); run<2048>(buf, wps); run<4096>(buf, wps); run<8192>(buf, wps); }
    return 0;
}
