// quad_lat.hip — latency of ONE dependent chain on gfx950, in shader cycles (s_memtime) and in wall time
// (s_memrealtime, 100 MHz), for the instruction kinds of a four-lane Blake2b compression and for the compression
// itself (csrc/quad_b2b.cuh).  A Merkle level near the root is one such compression deep, so this is the floor of the
// serial part of a FRI commit.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I stark_anatomy_amd/csrc tools/quad_lat.hip -o tools/quad_lat
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "quad_b2b.cuh"

#define R8(x) x x x x x x x x
#define R64(x) R8(R8(x))

// KIND >= 10: the same instruction on four independent registers (issue cost rather than dependent latency)
// KIND: 0 v_xor_b32 dependent, 1 v_lshl_add_u64 dependent, 2 v_mov_b32_dpp quad_perm dependent, 3 v_alignbit dependent,
//       4 v_xor_b32 x2 independent halves (the shape of a 64-bit xor), 5 v_add_co/v_addc pair dependent
template <int KIND>
__global__ void k_chain(uint64_t* out, int iters) {
    uint32_t a = threadIdx.x * 2654435761u + 1, b = a ^ 0x9e3779b9u, c = 0x85ebca6bu;
    uint32_t e = a * 3, f = a * 5, ldsaddr = (threadIdx.x & 63) * 16;
    uint64_t w = ((uint64_t)a << 32) | b, z = 0x0123456789abcdefULL, w2 = w * 3;
    __shared__ uint64_t lds_dummy[256];
    lds_dummy[threadIdx.x & 255] = w;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) asm volatile(R64("v_xor_b32 %0, %0, %1\n") : "+v"(a) : "v"(c));
        if (KIND == 1) asm volatile(R64("v_lshl_add_u64 %0, %0, 0, %1\n") : "+v"(w) : "v"(z));
        if (KIND == 2) asm volatile(R64("v_mov_b32_dpp %0, %0 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\ns_nop 1\n") : "+v"(a));
        if (KIND == 3) asm volatile(R64("v_alignbit_b32 %0, %0, %1, 7\n") : "+v"(a) : "v"(c));
        if (KIND == 4) asm volatile(R64("v_xor_b32 %0, %0, %2\nv_xor_b32 %1, %1, %2\n") : "+v"(a), "+v"(b) : "v"(c));
        if (KIND == 5) asm volatile(R64("v_add_co_u32 %0, vcc, %0, %2\nv_addc_co_u32 %1, vcc, %1, %2, vcc\n") : "+v"(a), "+v"(b) : "v"(c) : "vcc");
        if (KIND == 6) asm volatile(R64("v_xor_b32_dpp %0, %0, %1 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n") : "+v"(a) : "v"(c));
        if (KIND == 7) asm volatile(R64("v_perm_b32 %0, %0, %1, %2\n") : "+v"(a) : "v"(c), "v"(b));
        if (KIND == 8) asm volatile(R64("v_alignbyte_b32 %0, %0, %1, 3\n") : "+v"(a) : "v"(c));
        if (KIND == 9) asm volatile(R64("v_lshlrev_b64 %0, 1, %0\n") : "+v"(w));
        if (KIND == 10) asm volatile(R8(R8("v_xor_b32 %0, %0, %4\nv_xor_b32 %1, %1, %4\nv_xor_b32 %2, %2, %4\nv_xor_b32 %3, %3, %4\n")) : "+v"(a), "+v"(b), "+v"(e), "+v"(f) : "v"(c));
        if (KIND == 11) asm volatile(R8(R8("v_alignbit_b32 %0, %0, %4, 7\nv_alignbit_b32 %1, %1, %4, 7\nv_alignbit_b32 %2, %2, %4, 7\nv_alignbit_b32 %3, %3, %4, 7\n")) : "+v"(a), "+v"(b), "+v"(e), "+v"(f) : "v"(c));
        if (KIND == 12) asm volatile(R8(R8("v_mov_b32_dpp %0, %0 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\nv_mov_b32_dpp %1, %1 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\nv_mov_b32_dpp %2, %2 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\nv_mov_b32_dpp %3, %3 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n")) : "+v"(a), "+v"(b), "+v"(e), "+v"(f));
        if (KIND == 13) asm volatile(R8(R8("v_lshl_add_u64 %0, %0, 0, %2\nv_lshl_add_u64 %1, %1, 0, %2\n")) : "+v"(w), "+v"(w2) : "v"(z));
        if (KIND == 14) asm volatile(R8(R8("v_xor_b32_dpp %0, %0, %4 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\nv_xor_b32_dpp %1, %1, %4 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\nv_xor_b32_dpp %2, %2, %4 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\nv_xor_b32_dpp %3, %3, %4 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n")) : "+v"(a), "+v"(b), "+v"(e), "+v"(f) : "v"(c));
        if (KIND == 15) asm volatile(R8(R8("ds_read_b64 %0, %2\nds_read_b64 %1, %2 offset:8\n")) "s_waitcnt lgkmcnt(0)\n" : "=v"(w), "=v"(w2) : "v"(ldsaddr));
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }
    if (a == 0x12345 && b == 77 && w == 5 && e == 1 && f == 2 && w2 == 9) out[1000] = a + lds_dummy[3];
}

// Does a wave64 whose upper lanes are switched off issue faster?  The dependent v_xor chain with only lanes < `lanes` active.
__global__ void k_chain_masked(uint64_t* out, int iters, uint32_t lanes) {
    uint32_t a = threadIdx.x * 2654435761u + 1, c = 0x85ebca6bu;
    uint64_t t0 = 0, r0 = 0, t1 = 0, r1 = 0;
    if ((threadIdx.x & 63) < lanes) {
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) asm volatile(R64("v_xor_b32 %0, %0, %1\n") : "+v"(a) : "v"(c));
        t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
    }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }
    if (a == 0x12345) out[1000] = a;
}

// `chain` dependent compressions per quad: the digest goes back into the quad's message slot and is hashed again.
template <int ILP>
__global__ __launch_bounds__(512) void k_quad_chain(uint64_t* out, int chain) {
    __shared__ __align__(16) uint64_t sm[2 * 256 * QR_PAIR_U64];
    const uint32_t tid = threadIdx.x, q = tid & 3, quad = tid >> 2;
    const QuadLane L = quad_lane_init(q);
    for (uint32_t i = tid; i < 2 * 256 * QR_PAIR_U64; i += blockDim.x) sm[i] = i * 0x9e3779b97f4a7c15ULL;
    __syncthreads();
    unsigned char* m0 = reinterpret_cast<unsigned char*>(sm + quad * QR_PAIR_U64);
    unsigned char* m1 = reinterpret_cast<unsigned char*>(sm + (256 + quad) * QR_PAIR_U64);
    const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < chain; ++it) {
        uint64_t lo, hi, lo1, hi1;
        quad_compress(m0, L, L.d0_node, lo, hi);
        if (ILP == 2) quad_compress(m1, L, L.d0_node, lo1, hi1);
        reinterpret_cast<uint64_t*>(m0)[q] = lo;
        reinterpret_cast<uint64_t*>(m0)[4 + q] = hi;
        if (ILP == 2) { reinterpret_cast<uint64_t*>(m1)[q] = lo1; reinterpret_cast<uint64_t*>(m1)[4 + q] = hi1; }
        __builtin_amdgcn_s_waitcnt(0);
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }
    if (tid < 8) out[1024 + tid] = sm[tid];
}

static uint64_t* g_buf;
static void report(const char* name, int blocks, int threads, double units, const char* unit) {
    uint64_t h[2];
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, g_buf, sizeof h, hipMemcpyDeviceToHost);
    const double cyc = (double)h[0], ns = (double)h[1] * 10.0;
    printf("%-34s blocks=%4d threads=%4d  %8.1f cycles/%s  %7.1f ns/%s  clock %.2f GHz\n", name, blocks, threads, cyc / units, unit, ns / units, unit, cyc / ns);
}
template <int KIND>
static void chain(const char* name, int blocks, int threads, int per_iter) {
    const int iters = 400;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k_chain<KIND>), dim3(blocks), dim3(threads), 0, 0, g_buf, iters);
    report(name, blocks, threads, (double)iters * 64 * per_iter, "instr");
}
int main() {
    (void)hipMalloc(&g_buf, 1 << 20);
    for (int cfg = 0; cfg < 2; ++cfg) {
        const int blocks = cfg == 0 ? 1 : 256, threads = cfg == 2 ? 512 : (cfg == 1 ? 256 : 64);
        chain<0>("v_xor_b32 dependent", blocks, threads, 1);
        chain<4>("v_xor_b32 x2 (two halves)", blocks, threads, 2);
        chain<1>("v_lshl_add_u64 dependent", blocks, threads, 1);
        chain<5>("v_add_co + v_addc dependent", blocks, threads, 2);
        chain<2>("v_mov_b32_dpp + s_nop 1", blocks, threads, 1);
        chain<3>("v_alignbit_b32 dependent", blocks, threads, 1);
        chain<6>("v_xor_b32_dpp dependent", blocks, threads, 1);
        chain<7>("v_perm_b32 dependent", blocks, threads, 1);
        chain<8>("v_alignbyte_b32 dependent", blocks, threads, 1);
        chain<9>("v_lshlrev_b64 dependent", blocks, threads, 1);
        chain<10>("v_xor_b32 x4 independent", blocks, threads, 1);
        chain<11>("v_alignbit_b32 x4 independent", blocks, threads, 1);
        chain<12>("v_mov_b32_dpp x4 independent", blocks, threads, 1);
        chain<13>("v_lshl_add_u64 x2 independent", blocks, threads, 1);
        chain<14>("v_xor_b32_dpp x4 independent", blocks, threads, 1);
        chain<15>("ds_read_b64 x2 (+wait per 128)", blocks, threads, 2);
    }
    for (uint32_t lanes : {64u, 32u, 16u, 4u}) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_chain_masked, dim3(1), dim3(64), 0, 0, g_buf, 400, lanes);
        char nm[64]; snprintf(nm, sizeof nm, "v_xor_b32 dependent, %u lanes on", lanes);
        report(nm, 1, 64, 400.0 * 64, "instr");
    }
    const int n = 64;
    for (int threads : {64, 128, 256, 512}) {
        for (int blocks : {1, 256}) {
            for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k_quad_chain<1>), dim3(blocks), dim3(threads), 0, 0, g_buf, n);
            report("quad compression, 1 per quad", blocks, threads, n, "compr");
            for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k_quad_chain<2>), dim3(blocks), dim3(threads), 0, 0, g_buf, n);
            report("quad compression, 2 interleaved", blocks, threads, n, "pair");
        }
    }
    return 0;
}
