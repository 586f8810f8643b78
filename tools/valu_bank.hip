// valu_bank.hip — do VGPR bank conflicts (register index mod 4) cost VALU issue cycles on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CLOB "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51"
// same-bank: dst, src0, src1 all == 0 mod 4 ; spread: different banks
#define XS(d,a,b) "v_xor_b32 v" #d ", v" #a ", v" #b "\n"
#define AS(d,a,b,c) "v_alignbit_b32 v" #d ", v" #a ", v" #b ", v" #c "\n"
#define L64(d,a,b) "v_lshl_add_u64 v[" #d ":" #d "+1], v[" #a ":" #a "+1], 0, v[" #b ":" #b "+1]\n"
#define REP8(x) x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters) {
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) asm volatile(REP8(XS(20,24,28) XS(32,36,40) XS(24,28,20) XS(36,40,32) XS(28,20,24) XS(40,32,36) XS(44,48,20) XS(48,44,24)) ::: CLOB);   // same bank
        if (KIND == 1) asm volatile(REP8(XS(20,25,30) XS(32,37,42) XS(25,30,20) XS(37,42,32) XS(30,20,25) XS(42,32,37) XS(44,49,22) XS(49,44,27)) ::: CLOB);   // spread
        if (KIND == 2) asm volatile(REP8(AS(20,24,28,32) AS(36,40,44,48) AS(24,28,32,20) AS(40,44,48,36) AS(28,32,20,24) AS(44,48,36,40) AS(32,20,24,28) AS(48,36,40,44)) ::: CLOB);
        if (KIND == 3) asm volatile(REP8(AS(20,25,30,35) AS(36,41,46,51) AS(25,30,35,20) AS(41,46,51,36) AS(30,35,20,25) AS(46,51,36,41) AS(35,20,25,30) AS(51,36,41,46)) ::: CLOB);
        if (KIND == 4) asm volatile(REP8(L64(20,24,28) L64(32,36,40) L64(24,28,20) L64(36,40,32) L64(28,20,24) L64(40,32,36) L64(44,48,20) L64(48,44,24)) ::: CLOB);    // pairs start at same bank
        if (KIND == 5) asm volatile(REP8(L64(20,22,26) L64(32,34,38) L64(22,26,20) L64(34,38,32) L64(26,20,22) L64(38,32,34) L64(44,46,50) L64(46,50,44)) ::: CLOB);    // pairs alternate banks {0,1},{2,3}
    }
    uint32_t r;
    asm volatile("v_xor_b32 %0, v20, v32" : "=v"(r) :: CLOB);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int KIND>
static void run(const char* name, uint32_t* buf, int wps) {
    const int iters = 2000, blocks = 256 * wps;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), 0, 0, buf, iters);
    (void)hipDeviceSynchronize(); (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), 0, 0, buf, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s waves/SIMD=%d %7.3f ms  %5.2f cycles/instr/SIMD @2.4GHz\n", name, wps, ms, ms * 1e-3 * 2.4e9 / ((double)wps * iters * 64));
}
int main() {
    uint32_t* buf; (void)hipMalloc(&buf, (size_t)2048 * 256 * 4);
    for (int wps : {4, 8}) {
        run<0>("v_xor same bank", buf, wps); run<1>("v_xor spread banks", buf, wps);
        run<2>("v_alignbit same bank", buf, wps); run<3>("v_alignbit spread banks", buf, wps);
        run<4>("v_lshl_add_u64 pairs same bank", buf, wps); run<5>("v_lshl_add_u64 pairs alternating", buf, wps);
    }
    return 0;
}
