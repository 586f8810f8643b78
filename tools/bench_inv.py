#!/usr/bin/env python3
"""Time sh_dev_fp_inv at 2^log_n (resident data).  STARK_HIP_BINV_FLAT=1 / STARK_HIP_BINV_SPLIT_LOG=k choose the form."""
import ctypes as C, os, sys, time
import numpy as np
so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stark_anatomy_amd", "libstark_hip.so")
lib = C.CDLL(os.path.abspath(so))
ctx = C.c_void_p()
assert lib.sh_ctx_create(0, None, C.byref(ctx)) == 0
for log_n in [int(a) for a in sys.argv[1:]] or [24, 22, 20]:
    n = 1 << log_n
    d, o = C.c_void_p(), C.c_void_p()
    lib.sh_dev_alloc(ctx, C.c_size_t(n * 16), C.byref(d)); lib.sh_dev_alloc(ctx, C.c_size_t(n * 16), C.byref(o))
    x = np.frombuffer(np.random.default_rng(log_n).bytes(n * 16), dtype=np.uint64).copy()
    x[1::2] &= np.uint64(0x7FFFFFFFFFFFFFFF); x[0::2] |= np.uint64(1)
    lib.sh_dev_upload(ctx, d, C.c_void_p(x.ctypes.data), C.c_size_t(n * 16))
    lib.sh_dev_fp_inv.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    for _ in range(2): assert lib.sh_dev_fp_inv(ctx, d, o, n) == 0
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps): lib.sh_dev_fp_inv(ctx, d, o, n)
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"2^{log_n}: {ms:7.3f} ms  {n / ms / 1e6:6.1f} G inv/s", flush=True)
    lib.sh_dev_free(ctx, d); lib.sh_dev_free(ctx, o)
