// dist_kernels.hip — data movement of the multi-GPU path: the pack and unpack on either side of an
// all-to-all, as HIP kernels on the context's stream (the four-step NTT's exchanges moved whole slices through
// torch permute().contiguous() copies before), and the leaf order of a sharded round's top tree.
#include "kernels.h"

// out[c][r] = in[r][c] for a rows x cols matrix of 16-byte elements.  32 x 32 tiles through LDS (rows padded to 33):
// both the read and the write are 512-byte runs.  With cols = P·b this IS the pack of an exchange: out viewed as
// [dest s][col c][row r] is contiguous per destination.
__global__ __launch_bounds__(256) void k_transpose16(const uint4* __restrict__ in, uint4* __restrict__ out, uint32_t rows, uint32_t cols) {
    __shared__ uint4 tile[32][33];
    const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const uint32_t c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
    for (uint32_t k = 0; k < 32; k += 8) {
        const uint32_t r = r0 + ty + k, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + k][tx] = in[(size_t)r * cols + c];
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < 32; k += 8) {
        const uint32_t c = c0 + ty + k, r = r0 + tx;
        if (r < rows && c < cols) out[(size_t)c * rows + r] = tile[tx][ty + k];
    }
}
// The same transpose with the four-step twiddle folded into the read: out[c][r] = in[r][c]·w_n^((row0 + r)·c) — the pack of the
// exchange between the column and the row transforms, which saves the separate pass over the slice that k_twiddle_scale was.
__global__ __launch_bounds__(256) void k_transpose16_tw(const fp* __restrict__ in, fp* __restrict__ out, uint32_t rows, uint32_t cols, const fp* __restrict__ tw,
                                                        uint32_t log_n, uint64_t row0) {
    __shared__ uint4 tile[32][33];
    const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const uint32_t c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const uint64_t nmask = ((uint64_t)1 << log_n) - 1;
#pragma unroll
    for (uint32_t k = 0; k < 32; k += 8) {
        const uint32_t r = r0 + ty + k, c = c0 + tx;
        if (r < rows && c < cols) {
            const uint64_t e = (((row0 + r) & nmask) * c) & nmask;
            fp t = tw[e & ((1u << NTT_TW_LO_BITS) - 1)];
            if (log_n > NTT_TW_LO_BITS) t = mont_mul(t, tw[(1u << NTT_TW_LO_BITS) + (e >> NTT_TW_LO_BITS)]);
            const fp v = mont_mul(in[(size_t)r * cols + c], t);
            tile[ty + k][tx] = make_uint4(v.w[0], v.w[1], v.w[2], v.w[3]);
        }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < 32; k += 8) {
        const uint32_t c = c0 + ty + k, r = r0 + tx;
        if (r < rows && c < cols) reinterpret_cast<uint4*>(out)[(size_t)c * rows + r] = tile[tx][ty + k];
    }
}
// out[c][s][r] = in[s][c][r]: the unpack after an exchange — runs of `a` elements keep their order, the two outer
// indices (P sources, b columns) swap.
__global__ __launch_bounds__(256) void k_interleave_blocks(const uint4* __restrict__ in, uint4* __restrict__ out, uint32_t P, uint32_t b, uint32_t a) {
    const size_t total = (size_t)P * b * a, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const uint32_t r = (uint32_t)(i % a);
        const size_t sc = i / a;
        const uint32_t s = (uint32_t)(sc % P), c = (uint32_t)(sc / P);          // i runs over the OUTPUT [c][s][r]: writes are coalesced
        out[i] = in[((size_t)s * b + c) * a + r];
    }
}
// leaves of a sharded round's top tree from the all-gathered (P, 2) sub-roots: leaf i < P = pairs[i][0] (the root over
// half-slice i), leaf P + i = pairs[i][1] (half-slice P + i)
__global__ __launch_bounds__(256) void k_pair_leaves(const digest64* __restrict__ pairs, digest64* __restrict__ leaves, uint32_t P) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * P * 8) return;
    const uint32_t leaf = i >> 3, w = i & 7;
    leaves[leaf].h[w] = pairs[leaf < P ? 2 * leaf : 2 * (leaf - P) + 1].h[w];
}

hipError_t launch_transpose16(const fp* in, fp* out, size_t rows, size_t cols, hipStream_t s) {
    if (!rows || !cols) return hipSuccess;
    if (rows > 0xffffffffu || cols > 0xffffffffu || (rows + 31) / 32 > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_transpose16, dim3((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32)), dim3(256), 0, s,
                       reinterpret_cast<const uint4*>(in), reinterpret_cast<uint4*>(out), (uint32_t)rows, (uint32_t)cols);
    return hipGetLastError();
}
hipError_t launch_transpose16_tw(const fp* in, fp* out, size_t rows, size_t cols, const fp* tw, uint32_t log_n, uint64_t row0, hipStream_t s) {
    if (!rows || !cols) return hipSuccess;
    if (rows > 0xffffffffu || cols > 0xffffffffu || (rows + 31) / 32 > 65535 || log_n > 32 || (row0 + rows) > ((uint64_t)1 << 32)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_transpose16_tw, dim3((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32)), dim3(256), 0, s, in, out, (uint32_t)rows, (uint32_t)cols, tw, log_n, row0);
    return hipGetLastError();
}
hipError_t launch_interleave_blocks(const fp* in, fp* out, size_t P, size_t b, size_t a, hipStream_t s) {
    if (!P || !b || !a) return hipSuccess;
    if (P > 0xffffffffu || b > 0xffffffffu || a > 0xffffffffu) return hipErrorInvalidValue;
    size_t blocks = (P * b * a + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(k_interleave_blocks, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const uint4*>(in), reinterpret_cast<uint4*>(out), (uint32_t)P, (uint32_t)b, (uint32_t)a);
    return hipGetLastError();
}
hipError_t launch_pair_leaves(const digest64* pairs, digest64* leaves, size_t P, hipStream_t s) {
    if (!P || P > (1u << 20)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_pair_leaves, dim3((unsigned)((2 * P * 8 + 255) / 256)), dim3(256), 0, s, pairs, leaves, (uint32_t)P);
    return hipGetLastError();
}
