#!/usr/bin/env python3
"""Time sh_dev_ntt at 2^log_n (default 24, 20, 16); optional .so path to A/B builds.  Checks forward∘inverse == id."""
import ctypes as C, sys, time, os
import numpy as np
so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "stark_anatomy_amd", "libstark_hip.so")
logs = [int(a) for a in sys.argv[2:]] or [24, 20, 16]
lib = C.CDLL(os.path.abspath(so))
ctx = C.c_void_p()
assert lib.sh_ctx_create(0, None, C.byref(ctx)) == 0
for log_n in logs:
    n = 1 << log_n
    d = C.c_void_p()
    lib.sh_dev_alloc(ctx, C.c_size_t(n * 16), C.byref(d))
    x = np.frombuffer(np.random.default_rng(1).bytes(n * 16), dtype=np.uint64).copy()
    x[1::2] &= np.uint64(0x7FFFFFFFFFFFFFFF)
    lib.sh_dev_upload(ctx, d, C.c_void_p(x.ctypes.data), C.c_size_t(n * 16))
    assert lib.sh_dev_ntt(ctx, d, C.c_uint32(log_n), 0) == 0 and lib.sh_dev_ntt(ctx, d, C.c_uint32(log_n), 1) == 0
    y = np.empty_like(x)
    lib.sh_dev_download(ctx, C.c_void_p(y.ctypes.data), d, C.c_size_t(n * 16))
    ok = bool((x == y).all())
    lib.sh_ctx_sync(ctx)
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        lib.sh_dev_ntt(ctx, d, C.c_uint32(log_n), 0)
    lib.sh_ctx_sync(ctx)
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"2^{log_n}: {ms:8.3f} ms  {32.0 * n / ms / 1e6:7.1f} GB/s  roundtrip_ok={ok}  ({os.path.basename(so)})", flush=True)
    lib.sh_dev_free(ctx, d)
