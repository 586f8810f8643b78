// mixorder.hip — does the ORDER in which 4-byte (VOP2) and 8-byte (VOP3) VALU instructions are interleaved change
// the issue rate on gfx950?  Same multiset per 24-instruction group (16 VOP2 + 8 VOP3 = 128 B), different orders.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define X(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define A(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 7\n"
#define L(i) "v_lshl_add_u64 %" #i ", %" #i ", 0, %8\n"
#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)
#define P_ALT   X(0) X(1) A(2) X(3) X(4) A(5) X(6) X(7) A(0) X(1) X(2) A(3) X(4) X(5) A(6) X(7) X(0) A(1) X(2) X(3) A(4) X(5) X(6) A(7)
#define P_PAIR  X(0) X(1) X(2) X(3) A(4) A(5) X(6) X(7) X(0) X(1) A(2) A(3) X(4) X(5) X(6) X(7) A(0) A(1) X(2) X(3) X(4) X(5) A(6) A(7)
#define P_GRP4  X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) A(0) A(1) A(2) A(3) X(4) X(5) X(6) X(7) X(0) X(1) X(2) X(3) A(4) A(5) A(6) A(7)
#define P_GRP8  X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) A(0) A(1) A(2) A(3) A(4) A(5) A(6) A(7)
#define R4(p) p p p p
template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19, b = seed | 1;
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) asm volatile(R4(P_ALT) OPS);
        if (KIND == 1) asm volatile(R4(P_PAIR) OPS);
        if (KIND == 2) asm volatile(R4(P_GRP4) OPS);
        if (KIND == 3) asm volatile(R4(P_GRP8) OPS);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
template <int KIND>
static void run(const char* name, uint32_t* buf, int wps) {
    const int iters = 2000, blocks = 256 * wps;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), 0, 0, buf, iters, 7u);
    (void)hipDeviceSynchronize(); (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), 0, 0, buf, iters, 7u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-22s waves/SIMD=%d %7.3f ms  %5.3f cycles/byte/SIMD @2.4GHz\n", name, wps, ms, ms * 1e-3 * 2.4e9 / ((double)wps * iters * 4 * 128));
}
int main() {
    uint32_t* buf; (void)hipMalloc(&buf, (size_t)2048 * 256 * 4);
    for (int wps : {3, 4, 6, 8}) { run<0>("alternating 2:1", buf, wps); run<1>("pairs", buf, wps); run<2>("groups of 4/8", buf, wps); run<3>("16 VOP2 then 8 VOP3", buf, wps); }
    return 0;
}
