// bar_probe.hip — can the host store straight into device memory (fine-grained / uncached allocation seen through the PCIe BAR),
// and how long does a host -> GPU -> host ping take when the GPU polls (a) a pinned host word, (b) such a device word?
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/bar_probe tools/bar_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstdint>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }
// waits until *flag == want (system-scope loads), then answers by storing `want` to *ack (pinned host memory); n rounds
__global__ void k_pong(const uint32_t* flag, uint32_t* ack, uint32_t rounds, uint64_t timeout_ticks) {
    for (uint32_t r = 1; r <= rounds; ++r) {
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != r) {
            if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) { __hip_atomic_store(ack, 0xdeadu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return; }
            __builtin_amdgcn_s_sleep(2);
        }
        __hip_atomic_store(ack, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static int ping(const char* name, volatile uint32_t* h_flag_view, const uint32_t* d_flag_view, volatile uint32_t* h_ack, uint32_t* d_ack) {
    const uint32_t rounds = 2000;
    *h_flag_view = 0; *h_ack = 0;
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    hipLaunchKernelGGL(k_pong, dim3(1), dim3(1), 0, 0, d_flag_view, d_ack, rounds, (uint64_t)200000000);   // 2 s
    CK(hipGetLastError());
    double sum = 0, worst = 0;
    for (uint32_t r = 1; r <= rounds; ++r) {
        const double t0 = now_us();
        *h_flag_view = r;
        __atomic_thread_fence(__ATOMIC_SEQ_CST);
        while (*h_ack != r) { if (*h_ack == 0xdeadu || now_us() - t0 > 3e6) { printf("%s: no answer in round %u (ack %u)\n", name, r, *h_ack); (void)hipDeviceSynchronize(); return 1; } }
        const double dt = now_us() - t0;
        if (r > 100) { sum += dt; if (dt > worst) worst = dt; }
    }
    CK(hipDeviceSynchronize());
    printf("%-44s ping (host store -> GPU sees it -> GPU store -> host sees it): mean %.2f us, worst %.2f us\n", name, sum / (rounds - 100), worst);
    return 0;
}
int main() {
    CK(hipSetDevice(0));
    uint32_t *h_pin = nullptr, *d_pin = nullptr;
    CK(hipHostMalloc((void**)&h_pin, 4096, hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void**)&d_pin, h_pin, 0));
    memset(h_pin, 0, 4096);
    if (ping("pinned host word (product's mailbox)", h_pin, d_pin, h_pin + 64, d_pin + 64)) return 1;
    struct { const char* name; unsigned flag; } kinds[] = {{"device memory, hipDeviceMallocFinegrained", hipDeviceMallocFinegrained}, {"device memory, hipDeviceMallocUncached", hipDeviceMallocUncached},
                                                           {"device memory, hipDeviceMallocDefault", hipDeviceMallocDefault}};
    for (auto& k : kinds) {
        uint32_t* d = nullptr;
        hipError_t e = hipExtMallocWithFlags((void**)&d, 4096, k.flag);
        if (e != hipSuccess) { printf("%s: allocation failed: %s\n", k.name, hipGetErrorString(e)); continue; }
        CK(hipMemset(d, 0, 4096));
        CK(hipDeviceSynchronize());
        signal(SIGSEGV, on_segv); signal(SIGBUS, on_segv);
        if (sigsetjmp(jb, 1)) { printf("%s: host store FAULTS (no host access to this memory)\n", k.name); signal(SIGSEGV, SIG_DFL); signal(SIGBUS, SIG_DFL); continue; }
        volatile uint32_t* hv = d;
        *hv = 0;                                   // faults here if the host cannot reach it
        const uint32_t back = *hv;
        signal(SIGSEGV, SIG_DFL); signal(SIGBUS, SIG_DFL);
        printf("%s: host store and load work (read back %u)\n", k.name, back);
        if (ping(k.name, hv, d, h_pin + 64, d_pin + 64)) return 1;
        (void)hipFree(d);
    }
    return 0;
}
