// api_noalloc.hip — the one entry point of the library that allocates nothing: sh_dev_merkle_commit_elems_noalloc.
//
// For callers that cannot call hipMalloc (tight serving loops, harnesses that forbid allocation).  It allocates nothing but it
// does wait for the root on the host (mailbox poll, or a copy + stream synchronise), so it is NOT usable under stream capture.
// It is sh_dev_merkle_commit_elems (api_merkle.hip) with the digest buffers — which that entry point takes from the
// context's grow-only hipMalloc'ed scratch — placed in a file-scope __device__ array instead.  Same kernels, same root.
//
// `scratch` below is a host-side stand-in for the device array, not a device address: host code never passes it to a
// kernel or a memset.  noalloc_init_device() resolves the array's address on each GPU once (hipGetSymbolAddress with that
// device current; every GPU holds its own copy at its own address) and everything afterwards uses that pointer.
//
// Every call on one GPU shares that GPU's array.  Nothing in it has to start at zero (each call overwrites every digest
// it later reads), but two calls that overlap on one GPU — two contexts / streams on the same device — race on it.
#include "api_internal.h"

#define NOALLOC_A_DIGESTS ((size_t)1 << 17)                       // subtree digests: L >> K <= 2^17
#define NOALLOC_B_DIGESTS ((size_t)1 << QUAD_MAX_LOG_CHUNK)       // output of the first quad reduction
#define NOALLOC_BYTES ((NOALLOC_A_DIGESTS + NOALLOC_B_DIGESTS + 4) * sizeof(digest64))   // 8 MiB + 32 KiB + 256 B

static __device__ __align__(256) unsigned char scratch[NOALLOC_BYTES];

#define NOALLOC_MAX_DEVICES 64
static std::atomic<unsigned char*> g_scratch_ptr[NOALLOC_MAX_DEVICES] = {};   // published once per device; two threads creating contexts may both resolve it (same address)

// Called by sh_ctx_create with `device` current.  Idempotent per device.
int32_t noalloc_init_device(int device) {
    if (device < 0 || device >= NOALLOC_MAX_DEVICES) return SH_OK;      // no slot: the entry point reports it
    if (g_scratch_ptr[device].load(std::memory_order_acquire)) return SH_OK;
    void* p = nullptr;
    HIPCHK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) return sh_fail(SH_EHIP, "hipGetSymbolAddress returned a null address for the no-alloc scratch array");
    g_scratch_ptr[device].store((unsigned char*)p, std::memory_order_release);
    return SH_OK;
}

extern "C" int32_t sh_dev_merkle_commit_elems_noalloc(sh_ctx* c, const uint64_t* d_elems, size_t L, uint8_t root[64]) {
    CTXCHK(c);
    if (!d_elems || !root) return sh_fail(SH_EINVAL, "sh_dev_merkle_commit_elems_noalloc: null pointer");
    if (!is_pow2(L)) return sh_fail(SH_EINVAL, "sh_dev_merkle_commit_elems_noalloc: element count must be a power of two (merkle.rs:36)");
    const uint32_t lg = ilog2(L);
    if (lg > SH_NOALLOC_MAX_LOG_L)
        return sh_failf(SH_EINVAL, "sh_dev_merkle_commit_elems_noalloc: L = 2^%u exceeds the fixed scratch (max 2^%d elements); use sh_dev_merkle_commit_elems", lg, SH_NOALLOC_MAX_LOG_L);
    unsigned char* base = (c->device >= 0 && c->device < NOALLOC_MAX_DEVICES) ? g_scratch_ptr[c->device].load(std::memory_order_acquire) : nullptr;
    if (!base) return sh_fail(SH_EHIP, "sh_dev_merkle_commit_elems_noalloc: scratch array not resolved on this device");
    digest64* bufA = (digest64*)base;
    digest64* bufB = bufA + NOALLOC_A_DIGESTS;
    digest64* d_root = bufB + NOALLOC_B_DIGESTS;
    c->mail_pending = false;
    bool mailed = false;
    if (L == 1) HIPCHK(launch_hash_leaves((const fp*)d_elems, d_root, 1, c->stream));
    else {
        const uint32_t K = lg > 17 ? lg - 17 : 0;                  // <= 3: at most 2^17 subtree digests
        size_t count = L >> K;
        uint32_t lc = lg - K;
        HIPCHK(launch_leaf_subtree((const fp*)d_elems, bufA, count, K, c->stream));
        const digest64* cur = bufA;
        if (lc == 0) HIPCHK(hipMemcpyAsync(d_root, cur, sizeof(digest64), hipMemcpyDeviceToDevice, c->stream));
        else {
            if (lc > QUAD_MAX_LOG_CHUNK) {
                const uint32_t first = (lc + 1) / 2;               // <= 9, leaves <= 2^8 digests
                HIPCHK(launch_quad_reduce(cur, bufB, count, first, nullptr, 0, c->stream));
                cur = bufB; count >>= first; lc -= first;
            }
            if (++c->mail_seq == 0) c->mail_seq = 1;
            HIPCHK(launch_quad_reduce(cur, d_root, count, lc, c->d_mail, c->mail_seq, c->stream));
            mailed = true;
        }
    }
    if (mailed) {
        bool got = false;
        SHCHK(wait_mailbox(c, &got));
        if (got) { memcpy(root, c->h_mail, 64); return SH_OK; }
        ++c->mail_fallbacks;
    }
    HIPCHK(hipMemcpyAsync(c->h_pinned, d_root, 64, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy(root, c->h_pinned, 64);
    return SH_OK;
}
