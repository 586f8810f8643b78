// valu_dep.hip — how much VALU issue rate a wave loses to back-to-back DEPENDENT instructions on gfx950,
// as a function of independent chains per wave (ILP) and waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define X(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define A(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 7\n"
#define R8(a,b,c,d,e,f,g,h) a b c d e f g h
#define B8(INS,a,b,c,d,e,f,g,h) R8(INS(a),INS(b),INS(c),INS(d),INS(e),INS(f),INS(g),INS(h))
#define BLK(INS,a,b,c,d,e,f,g,h) B8(INS,a,b,c,d,e,f,g,h) B8(INS,a,b,c,d,e,f,g,h) B8(INS,a,b,c,d,e,f,g,h) B8(INS,a,b,c,d,e,f,g,h) B8(INS,a,b,c,d,e,f,g,h) B8(INS,a,b,c,d,e,f,g,h) B8(INS,a,b,c,d,e,f,g,h) B8(INS,a,b,c,d,e,f,g,h)
#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)
template <int KIND, int ILP>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19, b = seed | 1;
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0 && ILP == 1) asm volatile(BLK(X,0,0,0,0,0,0,0,0) OPS);
        if (KIND == 0 && ILP == 2) asm volatile(BLK(X,0,1,0,1,0,1,0,1) OPS);
        if (KIND == 0 && ILP == 4) asm volatile(BLK(X,0,1,2,3,0,1,2,3) OPS);
        if (KIND == 0 && ILP == 8) asm volatile(BLK(X,0,1,2,3,4,5,6,7) OPS);
        if (KIND == 1 && ILP == 1) asm volatile(BLK(A,0,0,0,0,0,0,0,0) OPS);
        if (KIND == 1 && ILP == 2) asm volatile(BLK(A,0,1,0,1,0,1,0,1) OPS);
        if (KIND == 1 && ILP == 4) asm volatile(BLK(A,0,1,2,3,0,1,2,3) OPS);
        if (KIND == 1 && ILP == 8) asm volatile(BLK(A,0,1,2,3,4,5,6,7) OPS);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
template <int KIND, int ILP>
static void run(uint32_t* buf, int wps) {
    const int iters = 2000, blocks = 256 * wps;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, ILP>), dim3(blocks), dim3(256), 0, 0, buf, iters, 7u);
    (void)hipDeviceSynchronize(); (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, ILP>), dim3(blocks), dim3(256), 0, 0, buf, iters, 7u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double instr_per_simd = (double)wps * iters * 64;           // wave-instructions issued per SIMD
    printf("%-14s ILP=%d waves/SIMD=%d  %7.3f ms  %6.2f T lane-ops/s  %5.2f cycles/instr/SIMD @2.4GHz\n", KIND ? "v_alignbit" : "v_xor", ILP, wps, ms,
           (double)blocks * 256 * iters * 64 / ms / 1e9, ms * 1e-3 * 2.4e9 / instr_per_simd);
}
int main() {
    uint32_t* buf; (void)hipMalloc(&buf, (size_t)2048 * 256 * 4);
    for (int wps : {1, 2, 3, 4, 6, 8}) {
        run<0, 1>(buf, wps); run<0, 2>(buf, wps); run<0, 4>(buf, wps); run<0, 8>(buf, wps);
        run<1, 1>(buf, wps); run<1, 2>(buf, wps); run<1, 4>(buf, wps); run<1, 8>(buf, wps);
    }
    return 0;
}
