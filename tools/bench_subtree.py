#!/usr/bin/env python3
"""Time sh_dev_merkle_subtrees (k_leaf_subtree) for K = 0..4 at L = 2^log_l; optional .so path to A/B builds."""
import ctypes as C, sys, time, os
import numpy as np
so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "stark_anatomy_amd", "libstark_hip.so")
log_l = int(sys.argv[2]) if len(sys.argv) > 2 else 24
lib = C.CDLL(os.path.abspath(so))
ctx = C.c_void_p()
assert lib.sh_ctx_create(0, None, C.byref(ctx)) == 0
L = 1 << log_l
d_in, d_out = C.c_void_p(), C.c_void_p()
lib.sh_dev_alloc(ctx, C.c_size_t(L * 16), C.byref(d_in)); lib.sh_dev_alloc(ctx, C.c_size_t(L * 64), C.byref(d_out))
x = np.frombuffer(np.random.default_rng(1).bytes(L * 16), dtype=np.uint64).copy()
x[1::2] &= np.uint64(0x7FFFFFFFFFFFFFFF)
lib.sh_dev_upload(ctx, d_in, C.c_void_p(x.ctypes.data), C.c_size_t(L * 16))
for K in range(0, 5):
    n_out = L >> K
    for rep in range(2):
        lib.sh_dev_merkle_subtrees(ctx, d_in, C.c_size_t(n_out), C.c_uint32(K), d_out)
    lib.sh_ctx_sync(ctx)
    t0 = time.perf_counter(); reps = 5
    for rep in range(reps):
        lib.sh_dev_merkle_subtrees(ctx, d_in, C.c_size_t(n_out), C.c_uint32(K), d_out)
    lib.sh_ctx_sync(ctx)
    ms = (time.perf_counter() - t0) / reps * 1e3
    comp = L + (L - n_out)
    print(f"K={K} {ms:8.3f} ms  {comp/ms/1e6:7.2f} G compress/s  ({os.path.basename(so)})", flush=True)
