#!/usr/bin/env python3
"""Time sh_dev_coset_lde (scale + pad + NTT) at n = 2^(log_N-2) -> N = 2^log_N; optional .so path."""
import ctypes as C, sys, time, os
import numpy as np
so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "stark_anatomy_amd", "libstark_hip.so")
logs = [int(a) for a in sys.argv[2:]] or [24, 20]
lib = C.CDLL(os.path.abspath(so))
ctx = C.c_void_p()
assert lib.sh_ctx_create(0, None, C.byref(ctx)) == 0
G = 85408008396924667383611388730472331217
off = (C.c_uint64 * 2)(G & (2**64 - 1), G >> 64)
lib.sh_dev_coset_lde.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
for log_N in logs:
    N = 1 << log_N; n = N >> 2
    d_c, d_x = C.c_void_p(), C.c_void_p()
    lib.sh_dev_alloc(ctx, C.c_size_t(n * 16), C.byref(d_c)); lib.sh_dev_alloc(ctx, C.c_size_t(N * 16), C.byref(d_x))
    x = np.frombuffer(np.random.default_rng(1).bytes(n * 16), dtype=np.uint64).copy(); x[1::2] &= np.uint64(0x7FFFFFFFFFFFFFFF)
    lib.sh_dev_upload(ctx, d_c, C.c_void_p(x.ctypes.data), C.c_size_t(n * 16))
    for _ in range(2): assert lib.sh_dev_coset_lde(ctx, d_c, log_N - 2, 2, off, d_x) == 0
    lib.sh_ctx_sync(ctx)
    t0 = time.perf_counter(); reps = 10
    for _ in range(reps): lib.sh_dev_coset_lde(ctx, d_c, log_N - 2, 2, off, d_x)
    lib.sh_ctx_sync(ctx)
    y = np.empty(4, dtype=np.uint64); lib.sh_dev_download(ctx, C.c_void_p(y.ctypes.data), d_x, C.c_size_t(32))
    print(f"LDE -> 2^{log_N}: {(time.perf_counter() - t0) / reps * 1e3:8.3f} ms  first={y[0]:016x}", flush=True)
    lib.sh_dev_free(ctx, d_c); lib.sh_dev_free(ctx, d_x)
