#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    unsigned lane = threadIdx.x;
    unsigned a = 100 + lane, b = 200 + lane;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[lane] = r[0]; out[64 + lane] = r[1];
    auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[128 + lane] = q[0]; out[192 + lane] = q[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); unsigned h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    const char* n[4] = {"p32 vdst", "p32 src ", "p16 vdst", "p16 src "};
    for (int i = 0; i < 4; ++i) { printf("%s:", n[i]); for (int l = 0; l < 64; ++l) printf(" %u", h[i * 64 + l]); printf("\n"); }
}
