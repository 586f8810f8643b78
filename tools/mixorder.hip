// mixorder.hip — does the ORDER in which 4-byte (VOP2) and 8-byte (VOP3) VALU instructions are interleaved change
// the issue rate on gfx950?  Same multiset per 24-instruction group (16 VOP2 + 8 VOP3 = 128 B), different orders.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define X(i) "v_xor_b32 %" #i ", %" #i ", %[b]\n"
#define A(i) "v_alignbit_b32 %" #i ", %" #i ", %[b], 7\n"
#define L(i) "v_lshl_add_u64 %[w" #i "], %[w" #i "], 0, %[z]\n"
#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : [b] "v"(b)
#define OPS64 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), [w0] "+v"(w0), [w1] "+v"(w1), [w2] "+v"(w2), [w3] "+v"(w3), [w4] "+v"(w4), [w5] "+v"(w5), [w6] "+v"(w6), [w7] "+v"(w7) : [b] "v"(b), [z] "v"(z)
#define P_ALT   X(0) X(1) A(2) X(3) X(4) A(5) X(6) X(7) A(0) X(1) X(2) A(3) X(4) X(5) A(6) X(7) X(0) A(1) X(2) X(3) A(4) X(5) X(6) A(7)
#define P_PAIR  X(0) X(1) X(2) X(3) A(4) A(5) X(6) X(7) X(0) X(1) A(2) A(3) X(4) X(5) X(6) X(7) A(0) A(1) X(2) X(3) X(4) X(5) A(6) A(7)
#define P_GRP4  X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) A(0) A(1) A(2) A(3) X(4) X(5) X(6) X(7) X(0) X(1) X(2) X(3) A(4) A(5) A(6) A(7)
#define P_GRP8  X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) A(0) A(1) A(2) A(3) A(4) A(5) A(6) A(7)
#define X8 X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define A8 A(0) A(1) A(2) A(3) A(4) A(5) A(6) A(7)
#define L8 L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7)
#define P_GRP16 X8 X8 X8 X8 A8 A8 X8 X8 X8 X8 A8 A8
#define P_GRP32 X8 X8 X8 X8 X8 X8 X8 X8 A8 A8 A8 A8
// the shape of a Blake2b G over 4 columns as it can be grouped: 8 adds(64b) | 8 xor | 4 add | 8 xor 8 rot | 8 add | 8 xor 8 rot | 4 add | 8 xor 8 rot
#define P_G4 L8 X8 L(0) L(1) L(2) L(3) X8 A8 L8 X8 A8 L(4) L(5) L(6) L(7) X8 A8
#define R4(p) p p p p
template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19, b = seed | 1;
    uint64_t w0 = a0, w1 = a1 * 3ull, w2 = a2 * 5ull, w3 = a3 * 7ull, w4 = a4 * 9ull, w5 = a5 * 11ull, w6 = a6 * 13ull, w7 = a7 * 15ull, z = 0x123456789abcdefull + seed;
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) asm volatile(R4(P_ALT) OPS);
        if (KIND == 1) asm volatile(R4(P_PAIR) OPS);
        if (KIND == 2) asm volatile(R4(P_GRP4) OPS);
        if (KIND == 3) asm volatile(R4(P_GRP8) OPS);
        if (KIND == 4) asm volatile(P_GRP16 OPS);
        if (KIND == 5) asm volatile(P_GRP32 OPS);
        if (KIND == 6) asm volatile(P_G4 OPS64);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(w0 ^ w1 ^ w2 ^ w3 ^ w4 ^ w5 ^ w6 ^ w7);
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
    for (int wps : {3, 4, 6, 8}) { run<0>("alternating 2:1", buf, wps); run<1>("pairs", buf, wps); run<2>("groups of 4/8", buf, wps); run<3>("16 VOP2 then 8 VOP3", buf, wps); run<4>("32 VOP2 then 16 VOP3", buf, wps); run<5>("64 VOP2 then 32 VOP3", buf, wps); run<6>("G x4 grouped (32 X, 24 L, 24 A)", buf, wps); }
    return 0;
}
