// valu_peak.hip — micro-benchmark of the gfx950 integer VALU issue rates that bound the Blake2b and
// Montgomery kernels ("measure, don't assume" the quarter-rate multiply and the 64-bit add).
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_peak.hip -o tools/valu_peak ; run on the GPU box.
// Prints lane-ops/s per instruction kind with every SIMD holding `waves` waves.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// 8 independent chains x 8 instructions per asm block
#define REP8(INS) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)
#define BLOCK(INS) REP8(INS) REP8(INS) REP8(INS) REP8(INS) REP8(INS) REP8(INS) REP8(INS) REP8(INS)

#define XOR_I(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define ADD_I(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define ALIGN_I(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 7\n"
#define PERM_I(i) "v_alignbyte_b32 %" #i ", %" #i ", %8, 3\n"
#define MULLO_I(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define MULHI_I(i) "v_mul_hi_u32 %" #i ", %" #i ", %8\n"
#define ADD3_I(i) "v_add3_u32 %" #i ", %" #i ", %8, %8\n"
#define ADDCOP_I(i) "v_add_co_u32 %" #i ", vcc, %" #i ", %8\nv_addc_co_u32 %" #i ", vcc, %" #i ", %8, vcc\n"
#define XAD_I(i) "v_xad_u32 %" #i ", %" #i ", %8, %8\n"
// round 2: separate ENCODING size (VOP2 = 4 bytes, VOP3 = 8 bytes) from OP CLASS: the same two-operand ops forced into the
// 8-byte encoding, a three-operand float op the guide prices at full rate, and the other byte/bit permutes
#define XOR64E_I(i) "v_xor_b32_e64 %" #i ", %" #i ", %8\n"
#define ADD64E_I(i) "v_add_u32_e64 %" #i ", %" #i ", %8\n"
#define FMA_I(i) "v_fma_f32 %" #i ", %" #i ", %8, %8\n"
#define VPERM_I(i) "v_perm_b32 %" #i ", %" #i ", %8, %8\n"
#define BFI_I(i) "v_bfi_b32 %" #i ", %" #i ", %8, %8\n"
#define XOR3_I(i) "v_or3_b32 %" #i ", %" #i ", %8, %8\n"
#define LSHLOR_I(i) "v_lshl_or_b32 %" #i ", %" #i ", 3, %8\n"
#define PACK_I(i) "v_pack_b32_f16 %" #i ", %" #i ", %8 op_sel:[1,0,0]\n"
#define CNDMASK_I(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define FMAC_I(i) "v_fmac_f32 %" #i ", %8, %8\n"
#define LSHR_I(i) "v_lshrrev_b32 %" #i ", 7, %" #i "\n"
// sub-dword addressing: a xor that reads word 1 of both sources and writes word 0 of the destination, the rest of it kept (the
// xor and the rotate by 16 of Blake2b's G in one instruction per 16-bit word) or zeroed
// 24-bit multiplies: would a field multiply on 24-bit limbs issue faster than 20 v_mad_u64_u32?
#define MUL24_I(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define MULHI24_I(i) "v_mul_hi_u32_u24 %" #i ", %" #i ", %8\n"
#define MAD24_I(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %8\n"
#define SDWAP_I(i) "v_xor_b32_sdwa %" #i ", %" #i ", %8 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1\n"
#define SDWAZ_I(i) "v_xor_b32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n"
#define SDWAB_I(i) "v_xor_b32_sdwa %" #i ", %" #i ", %8 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_0\n"

template <int KIND>
__global__ __launch_bounds__(256) void k32(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint32_t b = seed | 1;
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) asm volatile(BLOCK(XOR_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (KIND == 1) asm volatile(BLOCK(ADD_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (KIND == 2) asm volatile(BLOCK(ALIGN_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (KIND == 3) asm volatile(BLOCK(PERM_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (KIND == 4) asm volatile(BLOCK(MULLO_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (KIND == 5) asm volatile(BLOCK(MULHI_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (KIND == 6) asm volatile(BLOCK(ADD3_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (KIND == 7) asm volatile(BLOCK(XAD_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (KIND == 9) asm volatile(BLOCK(XOR64E_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (KIND == 10) asm volatile(BLOCK(ADD64E_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (KIND == 11) asm volatile(BLOCK(FMA_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (KIND == 12) asm volatile(BLOCK(VPERM_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (KIND == 13) asm volatile(BLOCK(BFI_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (KIND == 14) asm volatile(BLOCK(XOR3_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (KIND == 15) asm volatile(BLOCK(LSHLOR_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (KIND == 16) asm volatile(BLOCK(FMAC_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (KIND == 17) asm volatile(BLOCK(LSHR_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (KIND == 18) asm volatile(BLOCK(PACK_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (KIND == 19) asm volatile(BLOCK(CNDMASK_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
        if (KIND == 20) asm volatile(BLOCK(SDWAP_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (KIND == 21) asm volatile(BLOCK(SDWAZ_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (KIND == 22) asm volatile(BLOCK(SDWAB_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (KIND == 23) asm volatile(BLOCK(MUL24_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (KIND == 24) asm volatile(BLOCK(MULHI24_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (KIND == 25) asm volatile(BLOCK(MAD24_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (KIND == 8) asm volatile(BLOCK(ADDCOP_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

#define LSHLADD_I(i) "v_lshl_add_u64 %" #i ", %" #i ", 0, %8\n"
#define MAD64_I(i) "v_mad_u64_u32 %" #i ", vcc, %9, %10, %" #i "\n"
#define SHR64_I(i) "v_lshrrev_b64 %" #i ", 24, %" #i "\n"
#define SHL64_I(i) "v_lshlrev_b64 %" #i ", 1, %" #i "\n"
#define MOV64_I(i) "v_mov_b64 %" #i ", %8\n"
#define ADDCO_I(i) "v_add_co_u32 %" #i ", vcc, %" #i ", %8\nv_addc_co_u32 %" #i ", vcc, %" #i ", %8, vcc\n"

template <int KIND>
__global__ __launch_bounds__(256) void k64(uint64_t* out, int iters, uint32_t seed) {
    uint64_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint64_t b = ((uint64_t)seed << 32) | 12345;
    uint32_t x = seed | 1, y = seed * 7 + 3;
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) asm volatile(BLOCK(LSHLADD_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (KIND == 2) asm volatile(BLOCK(SHR64_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (KIND == 3) asm volatile(BLOCK(SHL64_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (KIND == 4) asm volatile(BLOCK(MOV64_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (KIND == 1) asm volatile(BLOCK(MAD64_I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(x), "v"(y) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

template <class F>
static double time_ms(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

__global__ void k_pack_check(const uint32_t* in, uint32_t* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t lo = in[2 * i], hi = in[2 * i + 1], r0, r1;
    asm volatile("v_pack_b32_f16 %0, %2, %3 op_sel:[1,0,0]\nv_pack_b32_f16 %1, %3, %2 op_sel:[1,0,0]\n" : "=&v"(r0), "=&v"(r1) : "v"(lo), "v"(hi));
    out[2 * i] = r0; out[2 * i + 1] = r1;                  // expected: the 64-bit value hi:lo rotated right by 16
}
static int pack_check() {
    const int n = 1 << 16;
    std::vector<uint32_t> h(2 * n), o(2 * n);
    uint64_t s = 12345;
    for (int i = 0; i < 2 * n; ++i) { s = s * 6364136223846793005ULL + 1442695040888963407ULL; h[i] = (uint32_t)(s >> 32); }
    const uint32_t nasty[] = {0x7c017c01u, 0xfc01fc01u, 0x00010001u, 0x80018001u, 0x7e007e00u, 0x7fffffffu, 0x0001fc01u, 0xffffffffu};   // sNaN, denormal, qNaN halves
    for (int i = 0; i < 64; ++i) h[i] = nasty[i % 8];
    uint32_t *di, *dout;
    CHK(hipMalloc(&di, 8 * n)); CHK(hipMalloc(&dout, 8 * n));
    CHK(hipMemcpy(di, h.data(), 8 * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_pack_check, dim3(n / 256), dim3(256), 0, 0, di, dout, n);
    CHK(hipMemcpy(o.data(), dout, 8 * n, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        const uint64_t v = ((uint64_t)h[2 * i + 1] << 32) | h[2 * i], r = (v >> 16) | (v << 48);
        if (o[2 * i] != (uint32_t)r || o[2 * i + 1] != (uint32_t)(r >> 32)) ++bad;
    }
    printf("v_pack_b32_f16 op_sel:[1,0,0] pair as a 64-bit rotate right by 16: %d of %d differ (sNaN / denormal halves included)\n", bad, n);
    return bad;
}
int main() {
    pack_check();
    const int iters = 4000;
    void* buf;
    CHK(hipMalloc(&buf, (size_t)256 * 8 * 4 * 256 * 8));
    const char* n32[] = {"v_xor_b32", "v_add_u32", "v_alignbit_b32", "v_alignbyte_b32", "v_mul_lo_u32", "v_mul_hi_u32", "v_add3_u32", "v_xad_u32", "add_co+addc (x2 instr)",
                         "v_xor_b32_e64 (VOP3 enc)", "v_add_u32_e64 (VOP3 enc)", "v_fma_f32 (VOP3)", "v_perm_b32", "v_bfi_b32", "v_or3_b32", "v_lshl_or_b32", "v_fmac_f32 (VOP2)", "v_lshrrev_b32 (VOP2)", "v_pack_b32_f16 op_sel", "v_cndmask_b32 (VOP2+vcc)",
                         "v_xor_b32_sdwa word->word keep", "v_xor_b32_sdwa word->dword", "v_xor_b32_sdwa byte->byte keep",
                         "v_mul_u32_u24", "v_mul_hi_u32_u24", "v_mad_u32_u24"};
    const char* n64[] = {"v_lshl_add_u64", "v_mad_u64_u32", "v_lshrrev_b64", "v_lshlrev_b64", "v_mov_b64"};
    for (int wps = 4; wps <= 8; wps *= 2) {                // waves per SIMD
        const int blocks = 256 * wps;                      // 256 CUs x (wps*4 waves = wps blocks of 256 threads)
        const double lane_ops = (double)blocks * 256 * iters * 64;
        printf("--- %d wave(s) per SIMD (%d blocks of 256) ---\n", wps, blocks);
        double ms;
#define RUN32(K) ms = time_ms([&] { hipLaunchKernelGGL(k32<K>, dim3(blocks), dim3(256), 0, 0, (uint32_t*)buf, iters, 7u); }); \
        printf("%-26s %8.3f ms  %8.2f T lane-ops/s  %5.2f cycles/instr/SIMD @2.4GHz\n", n32[K], ms, lane_ops / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)wps * iters * 64));
        RUN32(0) RUN32(1) RUN32(2) RUN32(3) RUN32(4) RUN32(5) RUN32(6) RUN32(7) RUN32(8)
        RUN32(9) RUN32(10) RUN32(11) RUN32(12) RUN32(13) RUN32(14) RUN32(15) RUN32(16) RUN32(17) RUN32(18) RUN32(19) RUN32(20) RUN32(21) RUN32(22) RUN32(23) RUN32(24) RUN32(25)
#define RUN64(K) ms = time_ms([&] { hipLaunchKernelGGL(k64<K>, dim3(blocks), dim3(256), 0, 0, (uint64_t*)buf, iters, 7u); }); \
        printf("%-26s %8.3f ms  %8.2f T lane-ops/s  %5.2f cycles/instr/SIMD @2.4GHz\n", n64[K], ms, lane_ops / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)wps * iters * 64));
        RUN64(0) RUN64(1) RUN64(2) RUN64(3) RUN64(4)
    }
    CHK(hipDeviceSynchronize());
    return 0;
}
