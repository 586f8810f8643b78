// cndmask_peak.hip — what a lane select costs on gfx950: v_cndmask_b32 with VCC or an SGPR pair as the mask, against a
// select on a VGPR mask (v_bfi_b32) and against plain full-rate / half-rate ops, full chip, 8 waves per SIMD.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/cndmask_peak.hip -o tools/cndmask_peak
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define R8(x) x x x x x x x x
#define I8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define BLOCK(OP) R8(I8(OP))
#define XOR_I(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define CND_VCC_I(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define CND_SGPR_I(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, %9\n"
#define BFI_I(i) "v_bfi_b32 %" #i ", %9, %" #i ", %8\n"
#define CMP_CND_I(i) "v_cmp_lt_u32 vcc, %" #i ", %8\ns_nop 1\nv_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define CMP_I(i) "v_cmp_lt_u32 vcc, %" #i ", %8\n"
#define ADDC_I(i) "v_addc_co_u32 %" #i ", vcc, %" #i ", %8, vcc\n"
#define SUBB_SEL_I(i) "v_subrev_co_u32 %" #i ", vcc, %8, %" #i "\nv_subbrev_co_u32 %9, vcc, 0, %9, vcc\n"
template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t t = threadIdx.x + seed, a0 = t, a1 = t * 3, a2 = t * 5, a3 = t * 7, a4 = t * 11, a5 = t * 13, a6 = t * 17, a7 = t * 19, b = seed | 1, m = t * 0x9e3779b9u;
    uint64_t sm = 0x5555aaaa5555aaaaULL;
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) asm volatile(BLOCK(XOR_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (KIND == 1) asm volatile(BLOCK(CND_VCC_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
        if (KIND == 2) asm volatile(BLOCK(CND_SGPR_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(sm));
        if (KIND == 3) asm volatile(BLOCK(BFI_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(m));
        if (KIND == 4) asm volatile(BLOCK(CMP_CND_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
        if (KIND == 5) asm volatile(BLOCK(CMP_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
        if (KIND == 6) asm volatile(BLOCK(ADDC_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ m;
}
template <int KIND>
static void run(const char* name, uint32_t* buf, int per) {
    const int iters = 2000, blocks = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, buf, iters, 7u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%-44s %8.3f ms  %6.2f cycles per instruction per SIMD @2.4GHz (8 waves/SIMD)\n", name, best, best * 1e-3 * 2.4e9 / (8.0 * iters * 64 * per));
}
int main() {
    uint32_t* buf;
    if (hipMalloc(&buf, (size_t)256 * 8 * 256 * 4) != hipSuccess) return 1;
    run<0>("v_xor_b32", buf, 1);
    run<1>("v_cndmask_b32 .., vcc (vcc constant)", buf, 1);
    run<2>("v_cndmask_b32_e64 .., s[n:n+1]", buf, 1);
    run<3>("v_bfi_b32 (select on a VGPR mask)", buf, 1);
    run<5>("v_cmp_lt_u32 vcc", buf, 1);
    run<4>("v_cmp_lt_u32 vcc; s_nop 1; v_cndmask (x2 instr)", buf, 2);
    run<6>("v_addc_co_u32 (vcc in and out)", buf, 1);
    return 0;
}
